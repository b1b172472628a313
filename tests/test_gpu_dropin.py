"""The drop-in, demonstrated: the reference's UNMODIFIED experiment::serial::Learner::RunEpisode and
rl::SARSA / rl::QLearn (compiled from /root/reference into oracle/_ref/ref_dropin, see oracle/Makefile)
drive environment::GpuIntraday -- a subclass of the reference's own environment::Base
(rl_markets_amd/host/ref_binding/gpu_intraday.h) whose book lives in the HIP engine -- through the very
call sites they use for environment::Intraday<>, and reproduce, step for step and bit for bit, the
trajectories the all-CPU reference produced (tests/golden/traj_*.npz): actions, rewards (the
reference's non-virtual Base::getReward() evaluated on the engine's mirrored members), TD errors, state
variables, RNG draw counts, the full book / order / position state, and the learned weights."""
import json
import os
import subprocess

import numpy as np
import pytest

from rl_markets_amd import abi, engine
from tests import oracle_lib as ol
from tests.golden.make_golden import TRAJ_CASES, gen_for
from tests.test_oracle_golden import GOLD

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "oracle", "_ref", "ref_dropin")

# the fixtures whose reward Base::getReward() can compute from mirrored members, default state variables
CASES = [c for c in TRAJ_CASES if c[0] in ("sarsa_b0", "qlearn_b3", "sarsa_mm_linear_b11", "sarsa_tight_bounds_b7", "sarsa_book_quotes_b9",
                                          "qlearn_mm_div_b16", "sarsa_lovol_b18", "sarsa_mm_exp_b20", "sarsa_boltzmann_b23", "sarsa_spread_b15")]


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/ref_dropin is built where the reference checkout is (make -C oracle dropin)")
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_reference_learner_drives_the_gpu_environment(tmp_path, case):
    name, algo, n_events, book, extra, over = case
    fx = np.load(os.path.join(GOLD, "traj_%s.npz" % name))
    want = fx["traj"][1:]  # (the fixture's first row is the state right after the reset)
    rec = engine.gen_stream_host(gen_for(n_events, over), 5, 2, book, 1)
    sp, out, th = str(tmp_path / "s.bin"), str(tmp_path / "t.traj"), str(tmp_path / "theta.bin")
    rec[0].tofile(sp)
    cmd = [DROPIN, "dropin", "--stream", sp, "--events", str(n_events), "--book", "0", "--depth", "5", "--trades", "2",
           "--algo", algo, "--mem", str(1 << 20), "--seed", "1994", "--rng_stream", str(book), "--eps", "0.8", "--out", out,
           "--theta_out", th, "--tmp", str(tmp_path / "h")]
    for k, v in extra.items():
        cmd += ["--" + k, str(v)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["sizeof_steprec"] == ol.STEP_DTYPE.itemsize and info["ok"] == 1
    got = np.fromfile(out, dtype=ol.STEP_DTYPE)
    assert len(got) == len(want) == int(fx["steps"]) == info["steps"]
    for f in ("action", "reward", "td", "rng_ctr", "n_vars"):
        np.testing.assert_array_equal(got[f], want[f], err_msg="%s: %s" % (name, f))
    np.testing.assert_array_equal(got["vars"], want["vars"], err_msg="%s: state variables" % name)
    for f in want["book"].dtype.names:
        if f == "cursor":
            continue
        assert np.array_equal(got["book"][f], want["book"][f]), "%s: book.%s first differs at step %d" % (
            name, f, int(np.argwhere(got["book"][f] != want["book"][f])[0][0]))
    # the weights the reference's agent learned from the GPU environment's rewards and states
    raw = np.fromfile(th, dtype=np.uint8)
    n = int(np.frombuffer(raw[:8].tobytes(), dtype=np.int64)[0])
    pairs = np.frombuffer(raw[8:8 + 16 * n].tobytes(), dtype=[("i", np.int64), ("v", np.float64)])
    np.testing.assert_array_equal(pairs["i"], fx["theta_idx"])
    np.testing.assert_array_equal(pairs["v"], fx["theta_val"])
    # Base::getEpisodeReward() / getEpisodePnL(): the reference's getters over the mirrored members
    # (RunEpisode's epilogue, ClearInventory through a Base&, is the reference's no-op on Base's own empty books)
    assert info["episode_reward"] == want["book"]["episode_reward"][-1]


def test_binding_rejects_window_rewards(tmp_path):
    """`normed` reads the mean AND deviation of Base's two PnL windows inside the NON-virtual Base::getReward(): the binding
    refuses it instead of returning a reward computed on empty windows (`spread` needs one mean, which mirror() supplies:
    the sarsa_spread_b15 case above)."""
    if not os.path.exists(DROPIN):
        pytest.skip("no ref_dropin")
    rec = engine.gen_stream_host(gen_for(200, {}), 5, 2, 0, 1)
    sp = str(tmp_path / "s.bin")
    rec[0].tofile(sp)
    res = subprocess.run([DROPIN, "dropin", "--stream", sp, "--events", "200", "--book", "0", "--algo", "sarsa", "--mem", "4096",
                          "--reward", "normed", "--lb_pnl", "10", "--tmp", str(tmp_path / "h")], capture_output=True, text=True)
    assert res.returncode != 0 and "getReward" in res.stderr


# ---- the learner half of the binding: experiment::serial::GpuLearner + rl::GpuAgent (gpu_learner.h) ------------------------
def _sparse_theta(path):
    raw = np.fromfile(path, dtype=np.uint8)
    n = int(np.frombuffer(raw[:8].tobytes(), dtype=np.int64)[0])
    return np.frombuffer(raw[8:8 + 16 * n].tobytes(), dtype=[("i", np.int64), ("v", np.float64)])


def _run_learner(tmp_path, name, books, episodes=1, extra=()):
    case = [c for c in TRAJ_CASES if c[0] == name][0]
    _, algo, n_events, book, _extra, over = case
    rec = engine.gen_stream_host(gen_for(n_events, over), 5, 2, book, 1)
    sp, th, st = str(tmp_path / "s.bin"), str(tmp_path / "theta.bin"), str(tmp_path / "stats.csv")
    rec[0].tofile(sp)
    cmd = [DROPIN, "dropin_learner", "--stream", sp, "--events", str(n_events), "--book", "0", "--depth", "5", "--trades", "2",
           "--algo", algo, "--mem", str(1 << 20), "--seed", "1994", "--rng_stream", str(book), "--eps", "0.8", "--books", str(books),
           "--episodes", str(episodes), "--theta_out", th, "--stats_out", st, "--tmp", str(tmp_path / "h")] + list(extra)
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    lines = [json.loads(l) for l in res.stdout.strip().splitlines() if l.startswith("{")]
    eps = [d for d in lines if "episode" in d]
    # the rows GpuLearner handed to the reference's "training_log" logger (Learner::RunEpisode, serial.cpp:81-88): episode, reward,
    # pnl, n_steps, epsilon -- one per episode, the numbers Base's getters report
    tlog = [d["training_log"] for d in lines if "training_log" in d]
    assert len(tlog) == len(eps) == episodes
    for e, row in zip(eps, tlog):
        assert row == [e["episode"], e["reward"], e["pnl"], e["steps"], e["descr"]]
    return case, rec, eps, _sparse_theta(th), open(st).read()


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/ref_dropin is built where the reference checkout is (make -C oracle dropin)")
def test_gpu_learner_writes_the_model_log_rows(tmp_path):
    """The `model_log` logger (Agent::HandleTransition, src/rl/agent.cpp:93-100: the mean |delta| of every 1000 updates): GpuLearner
    hands the engine's rows (lob_model_log_read) to the logger the reference's own Agent constructor registered.  One book over an
    episode of more than 3000 updates: the rows the all-CPU reference writes (the oracle's, pinned on them by
    tests/test_oracle_golden.py), bit for bit."""
    from tests import oracle_lib as ol
    n_events, book = 8000, 5
    g = engine.default_gen_params()
    g.n_events = n_events
    rec = engine.gen_stream_host(g, 5, 2, book, 1)
    sp = str(tmp_path / "s.bin")
    rec[0].tofile(sp)
    res = subprocess.run([DROPIN, "dropin_learner", "--stream", sp, "--events", str(n_events), "--book", "0", "--depth", "5", "--trades", "2",
                          "--algo", "sarsa", "--mem", str(1 << 16), "--seed", "1994", "--rng_stream", str(book), "--eps", "0.8", "--books", "1",
                          "--episodes", "1", "--tmp", str(tmp_path / "h")], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    lines = [json.loads(l) for l in res.stdout.strip().splitlines() if l.startswith("{")]
    rows = [d["model_log"] for d in lines if "model_log" in d]
    assert len(rows) == 1
    p = engine.default_params()
    p.memory_size, p.algo, p.book_id_offset = 1 << 16, abi.ALGO_SARSA, book
    o = ol.Oracle(p, rec)
    o.reset()
    o.td_step(6000)
    want = o.model_log()
    assert len(want) >= 3
    np.testing.assert_array_equal(np.array(rows[0]), want)
    o.close()


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/ref_dropin is built where the reference checkout is (make -C oracle dropin)")
def test_train_call_sequence_over_the_gpu_learner_one_book(tmp_path):
    """src/main.cpp's train() sequence -- runner.RunEpisode(agent) through a Runner& and an rl::Agent*, then Base's episode
    getters -- over GpuLearner / GpuAgent / GpuIntraday with ONE book: the all-CPU reference's trajectory fixture must come
    out -- the learned weights bit for bit (write_theta's host copy), episode reward / PnL, the step count, and the
    statistics behind getTotalTransactions() / writeStats() (Q17: only the trade statistics survive in the file)."""
    case, rec, eps, theta, stats = _run_learner(tmp_path, "qlearn_b3", books=1)
    fx = np.load(os.path.join(GOLD, "traj_qlearn_b3.npz"))
    last = fx["traj"][-1]["book"]
    np.testing.assert_array_equal(theta["i"], fx["theta_idx"])
    np.testing.assert_array_equal(theta["v"], fx["theta_val"])
    e = eps[0]
    assert e["reward"] == last["episode_reward"] and e["steps"] >= int(fx["steps"])
    # the fixture's last record is the state BEFORE RunEpisode's closing ClearInventory: one more market order may follow
    # (TradeStatistics::ask/bid_transactions: the books' n_transacted as of the last decision, base.cpp:415-416)
    ntr = int(last["ask_transactions"] + last["bid_transactions"] + last["market_buys"] + last["market_sells"])
    assert e["nTr"] in (ntr, ntr + 1)
    want = ["asks_placed,0", "bids_placed,0", "asks_cancelled,0", "bids_cancelled,0", "ask_transactions,%d" % last["ask_transactions"],
            "bid_transactions,%d" % last["bid_transactions"]]
    assert stats.splitlines()[:6] == want and stats.splitlines()[6].startswith("market_sells,") and len(stats.splitlines()) == 8


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/ref_dropin is built where the reference checkout is (make -C oracle dropin)")
@pytest.mark.parametrize("name", ["qlearn_b3", "sarsa_b0"])
def test_train_call_sequence_over_the_gpu_learner_batched(tmp_path, name):
    """The same call sequence with 96 books in the engine (every book replays the loaded day from its own policy stream),
    two episodes: against the oracle running the same batch -- shared weights to 1e-9 (f64 atomic order), book 0's episode
    totals as Base's getters report them."""
    B = 96
    case, rec, eps, theta, _ = _run_learner(tmp_path, name, books=B, episodes=2)
    _, algo, n_events, book, _extra, over = case
    from rl_markets_amd import abi
    p = engine.default_params()
    p.memory_size = 1 << 20
    p.algo = {"sarsa": abi.ALGO_SARSA, "q_learn": abi.ALGO_QLAMBDA}[algo]
    p.theta_mode = abi.THETA_SHARED
    p.book_id_offset = book
    for k, v in over.items():
        setattr(p, k, v)
    orc = ol.Oracle(p, np.repeat(rec, B, axis=0))
    for ep in range(2):
        orc.reset()
        for _ in range(4000):
            orc.td_step(8)
            if orc.counters()[2] == 0:
                break
        want_book0 = orc.rec(0)["book"]
        orc.clear_inventory()
        orc.handle_terminal()
        assert eps[ep]["reward"] == want_book0["episode_reward"], "episode %d" % ep
    oth = orc.theta(0)
    idx = np.flatnonzero(oth)
    np.testing.assert_array_equal(theta["i"], idx)
    np.testing.assert_allclose(theta["v"], oth[idx], rtol=1e-9, atol=1e-15)
    orc.close()
