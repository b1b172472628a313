"""Randomised differential run of the CSV ingestion: the UNMODIFIED reference reading a depth /
time-and-sales CSV pair through its own loaders (MarketDepth / TimeAndSales, src/data/basic.cpp) against the
oracle on the records `lob_convert_csv` makes of the same two files.  The files are crafted to hit what real
vendor files do: depth rows sharing a timestamp, crossed books, levels out of order, rows with a missing
column (the reference's reader never recovers: the day ends there), rows with a non-positive price (dropped),
a time-and-sales file that runs dry before the depth file does, several trades at one price inside one interval,
trades stamped exactly on a depth row, zero-size / zero-price trades, trade rows out of time order (the
reference consumes them in FILE order).  tests/golden/csv_q14.npz pins one hand-made day; this pins the
space around it.  Runs where oracle/_ref exists (the build container); LOB_REF_SWEEP=n widens it."""
import ctypes as C
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from rl_markets_amd import abi, engine
from tests import oracle_lib as ol
from tests.csv_io import ms_to_str
from tests.test_oracle_golden import _params_for, compare_traj

pytestmark = pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref/ref_harness not built (needs the reference checkout)")

T_SLOTS = abi.LOB_MAX_TRADES


def craft_csvs(seed, md_path, tas_path, ulps=True):
    r = np.random.default_rng(seed)
    g = engine.default_gen_params()
    g.seed = int(r.integers(0, 1 << 40))
    g.n_events = int(r.choice([120, 260, 400]))
    g.move_prob_q16 = int(r.uniform(0.1, 1.0) * 65536)
    g.trade_prob_q16 = 0   # trades are written below, independently of the generator's slots
    rec = engine.gen_stream_host(g, 5, 1, 0, 1)[0]
    f32 = rec.view(np.float32)
    n = rec.shape[0]
    times = rec[:, 0].astype(np.int64).copy()
    for i in r.choice(np.arange(2, n), size=int(r.integers(0, 8)), replace=False):
        times[i] = times[i - 1]                       # same-timestamp rows (quirk Q14)
    times = np.maximum.accumulate(times)
    crossed = set(int(i) for i in r.choice(np.arange(5, n), size=int(r.integers(0, 4)), replace=False))
    shuffled = set(int(i) for i in r.choice(np.arange(5, n), size=int(r.integers(0, 6)), replace=False))
    # a row with a missing column ends the reference's day (its reader glues lines until it has 22 columns): mostly late
    # in the file, sometimes before the look-back windows have filled (Initialise fails)
    lo_bad = 5 if r.integers(0, 5) == 0 else (2 * n) // 3
    # (not the last line: 21 columns + the empty read at the end of a newline-terminated file make a 22-column row whose
    #  last field is "" -- the reference dies of an uncaught std::invalid_argument from stol when it gets there)
    short = set(int(i) for i in r.choice(np.arange(lo_bad, n - 1), size=int(r.integers(0, 3)), replace=False))
    short = set(i for i in short if i - 1 not in short)   # two short lines in a row could add up to one 22-column row of garbage
    zeroed = set(int(i) for i in r.choice(np.arange(5, n), size=int(r.integers(0, 4)), replace=False))
    date = 20200102
    with open(md_path, "w") as md:
        md.write("date,time,ap1,ap2,ap3,ap4,ap5,av1,av2,av3,av4,av5,bp1,bp2,bp3,bp4,bp5,bv1,bv2,bv3,bv4,bv5\n")
        for i in range(n):
            ap, av = [float(x) for x in f32[i, 2:7]], [int(x) for x in rec[i, 7:12]]
            bp, bv = [float(x) for x in f32[i, 12:17]], [int(x) for x in rec[i, 17:22]]
            if i in crossed:
                ap, bp = bp[::-1], ap[::-1]
            if i in shuffled:                                   # the reference sorts the levels itself (book.cpp:86)
                o = r.permutation(5)
                ap, av = [ap[k] for k in o], [av[k] for k in o]
                o = r.permutation(5)
                bp, bv = [bp[k] for k in o], [bv[k] for k in o]
            if ulps and r.integers(0, 10) == 0:                 # level prices one float off the grid (same key, other bits)
                k = int(r.integers(0, 5))
                ap[k] = float(np.nextafter(np.float32(ap[k]), np.float32(1e9 if r.integers(0, 2) else 0.0)))
                k = int(r.integers(0, 5))
                bp[k] = float(np.nextafter(np.float32(bp[k]), np.float32(1e9 if r.integers(0, 2) else 0.0)))
            if i in zeroed:
                (ap if r.integers(0, 2) else bp)[int(r.integers(0, 5))] = 0.0   # row dropped (basic.cpp:54-58)
            cols = [str(date), ms_to_str(int(times[i]))] + ["%.9g" % x for x in ap] + ["%d" % x for x in av] + \
                   ["%.9g" % x for x in bp] + ["%d" % x for x in bv]
            if i in short:
                cols = cols[:int(r.integers(3, 22))]            # skipped by _LoadRow (basic.cpp:31-43)
            md.write(",".join(cols) + "\n")
    # trades: per depth interval up to 5 rows on at most 3 distinct prices around the touch
    rows = []
    for i in range(1, n):
        if r.uniform() > 0.45:
            continue
        lo, hi = int(times[i - 1]), int(times[i])
        px = [float(f32[i - 1, 2]), float(f32[i - 1, 12]), float(f32[i - 1, 3]), float(f32[i - 1, 13])]
        px = [px[k] for k in r.permutation(4)[:int(r.integers(1, 4))]]
        for _ in range(int(r.integers(1, 6))):
            t = hi if (hi == lo or r.integers(0, 4) == 0) else int(r.integers(lo + 1, hi + 1))   # (lo, hi], often == hi
            size = int(r.choice([0, 1, 7, 150, 2500], p=[0.05, 0.2, 0.25, 0.3, 0.2]))
            price = 0.0 if r.integers(0, 40) == 0 else px[int(r.integers(0, len(px)))]
            if ulps and price > 0.0 and r.integers(0, 6) == 0:
                # a neighbouring float: mostly the same 1e-4 price key (merged under the first-seen price by the reference's
                # map<double, long, FloatComparator>), now and then the next key
                price = float(np.nextafter(np.float32(price), np.float32(1e9 if r.integers(0, 2) else 0.0)))
            rows.append((t, price, size))
    # file order = generation order: within an interval the times are NOT sorted
    last_t = int(times[-1])
    with open(tas_path, "w") as ts:
        ts.write("date,time,price,size\n")
        bad_from = (len(rows) // 8) if r.integers(0, 5) == 0 else (2 * len(rows)) // 3
        for k, (t, price, size) in enumerate(rows):
            if k >= bad_from and r.integers(0, 120) == 0:
                ts.write("%d,%s,%.9g\n" % (date, ms_to_str(t), price))     # a 3-column row: the streamer's last (basic.cpp:138-150)
            ts.write("%d,%s,%.9g,%d\n" % (date, ms_to_str(t), price, size))
        # two sentinel groups after the last depth row keep the T&S streamer from running dry first; without them
        # the reference's day ends where the last group but one is due
        for k in range(int(r.choice([2, 2, 2, 1, 0]))):
            ts.write("%d,%s,1.0,1\n" % (date, ms_to_str(last_t + 3600000 + k)))
    return r


@pytest.mark.parametrize("seed", range(max(1, int(os.environ.get("LOB_REF_SWEEP", "200")) // 4)))
def test_random_csv_day_against_the_reference(seed):
    with tempfile.TemporaryDirectory() as td:
        md, tas = os.path.join(td, "md.csv"), os.path.join(td, "tas.csv")
        r = craft_csvs(91000 + seed, md, tas)
        algo = str(r.choice(["sarsa", "q_learn"]))
        stream = int(r.integers(0, 1000))
        out, th = os.path.join(td, "t.traj"), os.path.join(td, "theta.bin")
        res = subprocess.run([ol.REF_HARNESS, "episode", "--md", md, "--tas", tas, "--algo", algo, "--mem", str(1 << 18), "--seed", "1994",
                              "--rng_stream", str(stream), "--eps", "0.8", "--out", out, "--theta_out", th, "--tmp", os.path.join(td, "h"), "--clear_inventory", "1"],
                             capture_output=True, text=True)
        if res.returncode == 3 and "Initialise failed" in res.stderr:
            # the day ends before the look-back windows are full: no step is ever taken here either
            try:
                rec = engine.convert_csv(md, tas, T_SLOTS)
            except engine.LobError:
                return
            p = _params_for({}, algo, stream)
            p.max_trades = T_SLOTS
            o = ol.Oracle(p, rec)
            o.reset()
            o.td_step(3)
            assert o.counters()[0] == 0, "csv seed %d: the reference's Initialise fails on this pair" % seed
            o.close()
            return
        assert res.returncode == 0, res.stderr
        info = json.loads(res.stdout.strip().splitlines()[-1])
        traj = np.fromfile(out, dtype=ol.STEP_DTYPE)
        raw = np.fromfile(th, dtype=np.uint8)
        nn = int(np.frombuffer(raw[:8].tobytes(), dtype=np.int64)[0])
        pairs = np.frombuffer(raw[8:8 + 16 * nn].tobytes(), dtype=[("i", np.int64), ("v", np.float64)])
        rec = engine.convert_csv(md, tas, T_SLOTS)
    p = _params_for({}, algo, stream)
    p.memory_size = 1 << 18
    p.max_trades = T_SLOTS
    o = ol.Oracle(p, rec)
    o.reset()
    tag = "csv seed %d" % seed
    assert traj[-1]["action"] == -2      # the record after Runner::RunEpisode's ClearInventory (serial.cpp:31)
    compare_traj(lambda: o.td_step(1), lambda: o.rec(0), traj[:-1], tag)
    o.td_step(1)                         # the step that finds the day over (the close, or out of data: either file)
    assert o.counters()[0] == int(info["steps"]), tag
    o.clear_inventory()                  # ... and what it leaves behind: the books ClearInventory walks, position, PnL
    got = o.rec(0)["book"]
    for name in got.dtype.names:
        if name not in ("cursor", "n_traces", "terminal"):
            assert np.array_equal(got[name], traj[-1]["book"][name]), "%s after the episode: book.%s %r != %r" % (tag, name, got[name], traj[-1]["book"][name])
    thv = o.theta(0)
    nz = np.nonzero(thv)[0]
    np.testing.assert_array_equal(nz, pairs["i"], err_msg=tag)
    np.testing.assert_array_equal(thv[nz], pairs["v"], err_msg=tag)
    o.close()


@pytest.mark.parametrize("seed", range(max(1, int(os.environ.get("LOB_REF_SWEEP", "200")) // 8)))
def test_random_configuration_on_csv_days_multi_episode(seed):
    """The two sweeps crossed: a random configuration (tests/test_oracle_ref_sweep.py: agent, reward, variables,
    look-backs up to 100 events, bounds, policy, venue ...) run for 2-3 episodes over one crafted CSV day -- what the
    window sums carry from a day that ended at a short line, at a dry trade file or at the last row into the next
    Initialise (quirk Q7 x Q21 / Q22), step caps, ClearInventory and HandleTerminal in between."""
    from tests.test_oracle_golden import replay_multi
    from tests.test_oracle_ref_sweep import random_case, sparse, check_sparse
    with tempfile.TemporaryDirectory() as td:
        md, tas = os.path.join(td, "md.csv"), os.path.join(td, "tas.csv")
        r = craft_csvs(123000 + seed, md, tas)
        p, _g, algo, x = random_case(124000 + seed)
        x.pop("ticker")                      # the crafted prices sit on HSBA.L's grid and in its session
        assert abi.load().lob_market_preset(b"HSBA.L", C.byref(p.market)) == 0
        x["episodes"] = int(r.integers(2, 4))
        if r.integers(0, 2):
            x["steps"] = int(r.choice([5, 30, 90]))
        out, th, tb = os.path.join(td, "t.traj"), os.path.join(td, "theta.bin"), os.path.join(td, "theta_b.bin")
        cmd = [ol.REF_HARNESS, "episode", "--md", md, "--tas", tas, "--algo", algo, "--mem", str(p.memory_size), "--seed", str(p.seed),
               "--rng_stream", str(p.book_id_offset), "--eps", repr(p.epsilon), "--out", out, "--theta_out", th, "--tmp", os.path.join(td, "h")]
        if "double" in algo:
            cmd += ["--theta_b_out", tb]
        for k, v in x.items():
            cmd += ["--" + k, str(v)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        tag = "csv multi seed %d (%s, %s, %d episodes)" % (seed, algo, x["reward"], x["episodes"])
        p.max_trades = T_SLOTS
        if res.returncode == 3 and "Initialise failed" in res.stderr:
            try:
                rec = engine.convert_csv(md, tas, T_SLOTS)
            except engine.LobError:
                return
            o = ol.Oracle(p, rec)
            o.reset()
            o.td_step(3)
            assert o.counters()[0] == 0, tag + ": the reference's Initialise fails on this pair"
            o.close()
            return
        assert res.returncode == 0, res.stderr
        info = json.loads(res.stdout.strip().splitlines()[-1])
        traj = np.fromfile(out, dtype=ol.STEP_DTYPE)
        theta = sparse(th)
        theta_b = sparse(tb) if "double" in algo else None
        rec = engine.convert_csv(md, tas, T_SLOTS)
    o = ol.Oracle(p, rec)
    replay_multi({"traj": traj, "ends": np.array(info["ends"])}, o.reset, lambda: o.td_step(1), o.clear_inventory,
                 lambda: ol.load().oracle_handle_terminal(o.h), lambda: o.rec(0), tag)
    check_sparse(o.theta(0), theta[0], theta[1], tag)
    if theta_b is not None:
        check_sparse(o.theta_b(0), theta_b[0], theta_b[1], tag + " theta_b")
    o.close()
