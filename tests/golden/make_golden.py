#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the UNMODIFIED
reference (oracle/_ref/ref_harness = /root/reference sources compiled in
place by oracle/Makefile).  Runs only in the build container (the reference
checkout is not present on the GPU box); the fixtures it writes are committed.

    python tests/golden/make_golden.py [--only NAME ...]
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from rl_markets_amd import abi, engine  # noqa: E402  (host-side helpers only: stream generator)
from tests import oracle_lib as ol  # noqa: E402

HARNESS = ol.REF_HARNESS

# Fields of the per-step record kept in the trajectory fixtures.
TRAJ_CASES = [
    # name, algo, n_events, book id, extra harness args, param overrides for the oracle
    ("sarsa_b0", "sarsa", 700, 0, {}, {}),
    ("qlearn_b3", "q_learn", 700, 3, {}, {}),
    ("sarsa_mm_linear_b11", "sarsa", 500, 11, {"reward": "mm_linear", "pos_weight": "0.05"},
     {"reward_measure": abi.REWARD_MM_LINEAR, "pos_weight": 0.05}),
    ("qlearn_pnl_tpmid_b5", "q_learn", 500, 5, {"reward": "pnl", "tp": "microprice", "lb_target": 3},
     {"reward_measure": abi.REWARD_PNL, "target_price": abi.TP_MIDPRICE, "lb_target": 3}),
    ("sarsa_tight_bounds_b7", "sarsa", 600, 7, {"pos_ub": 20, "pos_lb": -20, "order_size": 15, "eps": "0.3"},
     {"pos_ub": 20, "pos_lb": -20, "order_size": 15, "epsilon": 0.3}),
    ("sarsa_book_quotes_b9", "sarsa", 500, 9, {"tp": "book"}, {"quote_mode": abi.QUOTE_BOOK, "target_price": abi.TP_MIDPRICE}),
    # every state variable of Intraday::getVariable (13), rsi / vwap windows switched on
    ("qlearn_all_vars_b13", "q_learn", 420, 13,
     {"vars": '"pos", "spd", "mpm", "imb", "svl", "vol", "rsi", "vwap", "a_dist", "a_queue", "b_dist", "b_queue", "last_action"',
      "lb_rsi": 10, "lb_vwap": 20},
     {"n_vars": 13, "vars": [abi.VAR_POS, abi.VAR_SPD, abi.VAR_MPM, abi.VAR_IMB, abi.VAR_SVL, abi.VAR_VOL, abi.VAR_RSI, abi.VAR_VWAP,
                             abi.VAR_A_DIST, abi.VAR_A_QUEUE, abi.VAR_B_DIST, abi.VAR_B_QUEUE, abi.VAR_LAST_ACTION],
      "lb_rsi": 10, "lb_vwap": 20}),
    # the remaining reward measures of Base::getReward
    ("sarsa_normed_b14", "sarsa", 420, 14, {"reward": "normed", "lb_pnl": 20}, {"reward_measure": abi.REWARD_NORMED, "lb_pnl": 20}),
    ("sarsa_spread_b15", "sarsa", 400, 15, {"reward": "spread"}, {"reward_measure": abi.REWARD_SPREAD}),
    ("qlearn_mm_div_b16", "q_learn", 400, 16, {"reward": "mm_div"}, {"reward_measure": abi.REWARD_MM_DIV}),
    ("sarsa_lovol_b18", "sarsa", 400, 18, {"reward": "lovol"}, {"reward_measure": abi.REWARD_LOVOL}),
    # mm_exp: -(1 - exp(pos_weight * |position|))^2 with the float std::exp of the reference's libm
    # rl::Boltzmann behaviour policy (SARSA also draws its bootstrap action from it, quirk Q9)
    ("sarsa_boltzmann_b23", "sarsa", 500, 23, {"policy": "boltzmann", "tau": "25.0"}, {"policy": abi.POLICY_BOLTZMANN, "tau": 25.0}),
    ("sarsa_mm_exp_b20", "sarsa", 500, 20, {"reward": "mm_exp", "pos_weight": "0.05", "eps": "0.5"},
     {"reward_measure": abi.REWARD_MM_EXP, "pos_weight": 0.05, "epsilon": 0.5}),
    # average-reward agents: rl::RLearn / rl::OnlineRLearn (src/rl/agent.cpp:357-412), rho updated after updateQ
    ("rlearn_b24", "r_learn", 500, 24, {"beta": "0.02", "eps": "0.4"}, {"beta": 0.02, "epsilon": 0.4}),
    ("online_rlearn_b25", "online_r_learn", 500, 25, {"beta": "0.01"}, {"beta": 0.01}),
    # a NaN state variable: vwap over a window without trades is 0/0, ulb() passes the NaN on, the
    # tile coder turns it into INT_MIN coordinates (x86 conversion) -- inside group 0, i.e. in the traces
    ("qlearn_vwap_nan_b19", "q_learn", 420, 19,
     {"vars": '"pos", "spd", "vwap", "imb", "svl", "vol", "a_dist", "b_dist"', "lb_vwap": 3},
     {"vars": [abi.VAR_POS, abi.VAR_SPD, abi.VAR_VWAP, abi.VAR_IMB, abi.VAR_SVL, abi.VAR_VOL, abi.VAR_A_DIST, abi.VAR_B_DIST],
      "lb_vwap": 3, "_gen": {"trade_prob_q16": 3277}}),
]


def gen_for(n_events, over):
    """Stream generator parameters of a trajectory case (`_gen` in its override dict)."""
    g = engine.default_gen_params()
    g.n_events = n_events
    for k, v in over.get("_gen", {}).items():
        setattr(g, k, v)
    return g


def nonfinite_vectors():
    """State vectors with NaN / +-inf / huge entries for the tile-coder known answers."""
    rng = np.random.default_rng(77)
    v = rng.uniform(-5, 5, size=(48, 8)).astype(np.float32)
    for i in range(16):
        v[i, rng.integers(0, 8)] = np.nan
    v[16:20, 2] = np.nan
    v[20:24, 0] = np.inf
    v[24:28, 5] = -np.inf
    v[28:32, 1] = 1e20
    v[32:36, 4] = -3e30
    v[36:40, 7] = np.nan
    v[40:44, 3] = 6.7e7      # floor(x * 32) just inside int range
    v[44:48, 6] = -6.8e7     # ... and just outside
    return v


# name, algo, n_events, book id, harness args (episodes, per-episode step cap)
MULTI_CASES = [
    ("early_stop_x3", "q_learn", 420, 21, {"episodes": 3, "steps": 110}),
    ("exhausted_x2", "sarsa", 260, 22, {"episodes": 2}),
]


def run(cmd):
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("%s\n%s\n%s" % (" ".join(cmd), res.stdout, res.stderr))
    return res.stdout


def main(out_dir=None):
    """Writes the fixtures next to this file, or into `out_dir` (tests/test_oracle_golden.py regenerates them
    there and compares with the committed ones)."""
    assert ol.have_ref(), "build oracle/_ref first: make -C oracle ref"
    OUT = out_dir or HERE
    rng = np.random.default_rng(20260925)
    with tempfile.TemporaryDirectory() as td:
        # ---- hash table ----
        p = os.path.join(td, "rnd.bin")
        run([HARNESS, "rndseq", "--out", p])
        rnd = np.fromfile(p, dtype=np.uint32)
        assert rnd.shape == (2048,)

        # ---- tiles known answers: rl::State::newState(vector<float>&) ----
        v = np.concatenate([
            rng.uniform(-12, 12, size=(200, 8)),
            rng.integers(-100, 20, size=(60, 8)).astype(np.float64),
            np.array([[1, -100, 2, 0, 3, 1.7, -2.5, 0.4], [0] * 8, [-0.0, -1e-9, 1e-9, 31.999999, -32, 5, 5, 5]]),
        ]).astype(np.float32)
        tiles = {}
        for mem in (20000000, 1 << 20, 999983):
            vin, vout = os.path.join(td, "v.f32"), os.path.join(td, "t.i32")
            v.tofile(vin)
            run([HARNESS, "tiles", "--mem", str(mem), "--nvars", "8", "--in", vin, "--out", vout])
            tiles[mem] = np.fromfile(vout, dtype=np.int32).reshape(v.shape[0], 9, 96)
        v5 = rng.uniform(-5, 5, size=(40, 5)).astype(np.float32)
        vin, vout = os.path.join(td, "v5.f32"), os.path.join(td, "t5.i32")
        v5.tofile(vin)
        run([HARNESS, "tiles", "--mem", "20000000", "--nvars", "5", "--in", vin, "--out", vout])
        tiles5 = np.fromfile(vout, dtype=np.int32).reshape(40, 9, 96)

        vn = nonfinite_vectors()
        vin, vout = os.path.join(td, "vn.f32"), os.path.join(td, "tn.i32")
        vn.tofile(vin)
        tiles_n = {}
        for mem in (20000000, 4099):
            run([HARNESS, "tiles", "--mem", str(mem), "--nvars", "8", "--in", vin, "--out", vout])
            tiles_n[mem] = np.fromfile(vout, dtype=np.int32).reshape(vn.shape[0], 9, 96)
        np.savez_compressed(os.path.join(OUT, "kat_nonfinite.npz"), vars=vn, **{"tiles_%d" % m: t for m, t in tiles_n.items()})
        if "--only-nonfinite" in sys.argv:
            return

        # ---- tick conversion known answers: Market::ToTicks / ToPrice / tick_size ----
        ticks = {}
        for ticker, lo, hi in (("HSBA.L", 0.5, 12000.0), ("BAES.L", 0.3, 12000.0), ("AIRF.PA", 0.01, 400.0),
                               ("CRDI.MI", 0.01, 80.0), ("NOKIA.HE", 0.01, 120000.0), ("NESN.VX", 0.2, 15000.0),
                               ("OMV.VI", 0.01, 300.0)):
            prices = np.concatenate([np.exp(rng.uniform(np.log(lo), np.log(hi), size=400)),
                                     np.float32(rng.uniform(lo, min(hi, 1000.0), size=200)).astype(np.float64),
                                     np.array([702.1, 702.5, 2750.0, 9.5, 12.6, 1.96885, 46021.0, 46025.0])])
            prices = prices[(prices >= lo)]
            pin, pout = os.path.join(td, "p.f64"), os.path.join(td, "p.out")
            prices.tofile(pin)
            run([HARNESS, "ticks", "--ticker", ticker, "--in", pin, "--out", pout])
            out = np.fromfile(pout, dtype=[("ticks", np.int32), ("pad", np.int32), ("back", np.float64), ("tick", np.float64)])
            ticks[ticker] = (prices, out["ticks"].copy(), out["back"].copy(), out["tick"].copy())

        np.savez_compressed(os.path.join(OUT, "kat_reference.npz"), rndseq=rnd, tiles_vars=v,
                            **{"tiles_%d" % m: t for m, t in tiles.items()}, tiles5_vars=v5, tiles5=tiles5,
                            **{"ticks_%s_%s" % (k, n): a for k, (pr, tk, bk, ts) in ticks.items()
                               for n, a in (("price", pr), ("ticks", tk), ("back", bk), ("tick", ts))})

        # ---- full trajectories through Intraday + Agent ----
        g = engine.default_gen_params()
        for name, algo, n_events, book, extra, _over in TRAJ_CASES:
            if "--only" in sys.argv and name not in sys.argv:
                continue
            rec = engine.gen_stream_host(gen_for(n_events, _over), 5, 2, book, 1)
            traj, info, theta = ol.run_ref_episode(rec[0], algo=algo, mem=1 << 20, rng_stream=book, extra=extra)
            np.savez_compressed(os.path.join(OUT, "traj_%s.npz" % name), traj=traj, theta_idx=theta[0],
                                theta_val=theta[1], steps=info["steps"], end=info["end"], rng_ctr=info["rng_ctr"])
            print(name, info)
        # ---- DoubleQLearn (config/example.yaml's default algorithm): theta_b + the agent's own mt19937_64 coin ----
        # ... and rl::DoubleRLearn, the same two vectors with the average-reward TD error (agent.cpp:416-467)
        for name, n_events, book, ralgo, rextra in (("double_q_b4", 520, 4, "double_q_learn", {}), ("double_q_b17", 400, 17, "double_q_learn", {}),
                                                    ("double_r_b26", 450, 26, "double_r_learn", {"beta": "0.02"})):
            if "--only" in sys.argv and name not in sys.argv:
                continue
            g.n_events = n_events
            rec = engine.gen_stream_host(g, 5, 2, book, 1)
            tb = os.path.join(td, "theta_b.bin")
            traj, info, theta = ol.run_ref_episode(rec[0], algo=ralgo, mem=1 << 20, rng_stream=book,
                                                   extra=dict({"agent_seed": 1994 + book, "theta_b_out": tb}, **rextra))
            raw = np.fromfile(tb, dtype=np.uint8)
            nn = int(np.frombuffer(raw[:8].tobytes(), dtype=np.int64)[0])
            pairs = np.frombuffer(raw[8:8 + 16 * nn].tobytes(), dtype=[("i", np.int64), ("v", np.float64)])
            np.savez_compressed(os.path.join(OUT, "traj_%s.npz" % name), traj=traj, theta_idx=theta[0], theta_val=theta[1],
                                theta_b_idx=pairs["i"].copy(), theta_b_val=pairs["v"].copy(), steps=info["steps"],
                                end=info["end"], rng_ctr=info["rng_ctr"])
            print(name, info, "theta_b nonzeros", nn)
        # ---- multi-episode runs: Runner::RunEpisode x N on one agent (quirks Q7, Q19) ----
        for name, algo, n_events, book, extra in MULTI_CASES:
            g.n_events = n_events
            rec = engine.gen_stream_host(g, 5, 2, book, 1)
            traj, info, theta = ol.run_ref_episode(rec[0], algo=algo, mem=1 << 20, rng_stream=book, extra=extra)
            np.savez_compressed(os.path.join(OUT, "multi_%s.npz" % name), traj=traj, theta_idx=theta[0],
                                theta_val=theta[1], steps=info["steps"], ends=np.array(info["ends"]), rng_ctr=info["rng_ctr"])
            print(name, info)
        # ---- recorded-data path: the reference reading CSV files that contain same-timestamp
        # depth rows, a crossed (invalid) book, multi-row trade hand-over (quirk Q14) ----
        from tests.csv_io import write_reference_csvs
        g.n_events = 420
        rec = engine.gen_stream_host(g, 5, 2, 31, 1)[0].copy()
        for r in (150, 151, 230, 305):          # rows sharing the previous row's timestamp
            rec[r, 0] = rec[r - 1, 0]
        for r in (180, 260):                     # crossed book: asks below bids -> IsValidState false
            rec[r, 2:7], rec[r, 12:17] = rec[r, 12:17][::-1].copy(), rec[r, 2:7][::-1].copy()
        md, tas = os.path.join(OUT, "q14_md.csv"), os.path.join(OUT, "q14_tas.csv")
        write_reference_csvs(rec, 5, 2, md, tas)
        out = os.path.join(td, "q14.traj")
        th = os.path.join(td, "q14.theta")
        res = run([HARNESS, "episode", "--md", md, "--tas", tas, "--algo", "sarsa", "--mem", str(1 << 20), "--seed", "1994",
                   "--rng_stream", "31", "--eps", "0.8", "--out", out, "--theta_out", th, "--tmp", os.path.join(td, "q14h")])
        import json
        info = json.loads(res.strip().splitlines()[-1])
        traj = np.fromfile(out, dtype=ol.STEP_DTYPE)
        raw = np.fromfile(th, dtype=np.uint8)
        nn = int(np.frombuffer(raw[:8].tobytes(), dtype=np.int64)[0])
        pairs = np.frombuffer(raw[8:8 + 16 * nn].tobytes(), dtype=[("i", np.int64), ("v", np.float64)])
        np.savez_compressed(os.path.join(OUT, "csv_q14.npz"), traj=traj, theta_idx=pairs["i"].copy(), theta_val=pairs["v"].copy(),
                            steps=info["steps"], end=info["end"])
        print("csv_q14", info)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
