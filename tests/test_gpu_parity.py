"""GPU parity tests: the HIP engine (through the C ABI) against the oracle
(oracle/liblob_oracle.so, itself pinned to the unmodified reference by
tests/test_oracle_vs_reference.py and tests/golden/).  Bit-exact for book /
order / state / reward / RNG; learned weights bit-exact for private theta and
within 1e-9 relative for shared theta (f64 atomic ordering)."""
import ctypes as C

import numpy as np
import pytest

from rl_markets_amd import abi, engine
from tests import oracle_lib as ol
from tests.parity import compare_env, compare_learner_step, dumps_to_np

pytestmark = pytest.mark.gpu


def make(depth=5, trades=2, n_events=500, B=4, algo=abi.ALGO_SARSA, theta_mode=abi.THETA_PRIVATE, mem=1 << 18,
         first_book=0, seed=1994, **over):
    p = engine.default_params()
    p.depth, p.max_trades = depth, trades
    p.algo, p.theta_mode, p.memory_size = algo, theta_mode, mem
    p.book_id_offset = first_book
    p.seed = seed
    for k, v in over.items():
        setattr(p, k, v)
    g = engine.default_gen_params()
    g.n_events = n_events
    rec = engine.gen_stream_host(g, depth, trades, first_book, B)
    eng = engine.Engine(p, B)
    eng.load_events(rec)
    orc = ol.Oracle(p, rec)
    return p, g, rec, eng, orc


def experiments_build():
    """-DLOB_EXPERIMENTS (tools/exp_variants.sh; LOB_ENGINE_LIB points the tests at that library): the kernel variants measured and
    lost exist -- a product build ignores their switches, so their parity tests would only repeat the product kernels'."""
    fn = abi.load().lob_experiments_enabled
    fn.restype, fn.argtypes = C.c_int, []
    return fn() == 1


def test_no_cpu_fallback_symbols():
    lib = abi.load()
    assert lib.lob_abi_version() == 6


def test_features_match_oracle():
    p = engine.default_params()  # M = 20 000 000
    eng = engine.Engine(p, 1)
    rng = np.random.default_rng(7)
    v = np.concatenate([
        rng.uniform(-12, 12, size=(300, 8)),
        rng.integers(-100, 20, size=(100, 8)).astype(np.float64),
        np.array([[1, -100, 2, 0, 3, 1.7, -2.5, 0.4], [0] * 8, [-0.0, -1e-9, 1e-9, 31.999999, -32, 5, 5, 5]]),
    ]).astype(np.float32)
    got = eng.features(v)
    want = np.zeros_like(got)
    ol.load().oracle_tiles(p.memory_size, ol.ptr(v), 8, v.shape[0], ol.ptr(want))
    np.testing.assert_array_equal(got, want)
    # SURVEY.md §8c known answer from the unmodified reference
    i = 400
    assert list(got[i, 0, :4]) == [7277022, 11975445, 7036608, 19940000]
    assert list(got[i, 8, 64:68]) == [6750499, 5357726, 13423606, 5907867]


def test_q_values_bitwise():
    p = engine.default_params()
    p.memory_size = 1 << 16
    eng = engine.Engine(p, 1)
    rng = np.random.default_rng(3)
    th = rng.standard_normal(p.memory_size)
    eng.set_theta(th)
    v = rng.uniform(-10, 10, size=(64, 8)).astype(np.float32)
    q = eng.q_values(v)
    f = eng.features(v)
    w = list(p.group_weights)
    for i in range(v.shape[0]):
        for a in range(9):
            Q = 0.0
            for k in range(32):
                Q += w[0] * th[f[i, a, k]]
            for k in range(32, 64):
                Q += w[1] * th[f[i, a, k]]
            for k in range(32, 96):
                Q += w[2] * th[f[i, a, k]]  # quirk Q3
            assert q[i, a] == Q


@pytest.mark.parametrize("depth,trades", [(5, 2), (10, 2), (3, 1)])
def test_env_random_actions(depth, trades):
    B = 64
    p, g, rec, eng, orc = make(depth=depth, trades=trades, n_events=700, B=B)
    eng.reset()
    orc.reset()
    compare_env(eng, orc, "reset")
    np.testing.assert_array_equal(eng.get_state(), orc.recs()["vars"][:, :8])
    rng = np.random.default_rng(depth)
    for step in range(150):
        a = rng.integers(0, 9, size=B).astype(np.int32)
        eng.step(a)
        orc.env_step(a)
        compare_env(eng, orc, "D=%d step %d" % (depth, step))
    live = eng.get_terminal() == 0
    np.testing.assert_array_equal(eng.get_state()[live], orc.recs()["vars"][live][:, :8])
    np.testing.assert_array_equal(eng.get_reward()[live], orc.recs()["reward"][live])
    c = eng.counters()
    oc = orc.counters()
    assert c[0] == oc[0] and c[1] == oc[1]


@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA, abi.ALGO_DOUBLE_Q])
def test_td_private_theta_bit_exact(algo):
    B = 16
    p, g, rec, eng, orc = make(n_events=600, B=B, algo=algo, theta_mode=abi.THETA_PRIVATE, first_book=100)
    eng.reset()
    orc.reset()
    for step in range(250):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "algo %d step %d" % (algo, step))
    for b in range(B):
        np.testing.assert_array_equal(eng.theta(b), orc.theta(b))
        if algo == abi.ALGO_DOUBLE_Q:
            np.testing.assert_array_equal(eng.theta(B + b), orc.theta_b(b))
        ei, ee = eng.traces(b)
        oi, oe = orc.traces(b)
        assert dict(zip(ei.tolist(), ee.tolist())) == dict(zip(oi.tolist(), oe.tolist()))


def test_td_runs_to_stream_end():
    B = 8
    p, g, rec, eng, orc = make(n_events=300, B=B, theta_mode=abi.THETA_PRIVATE)
    eng.reset()
    orc.reset()
    for step in range(260):
        eng.td_step(1)
        orc.td_step(1)
    compare_env(eng, orc, "end")
    assert (eng.get_terminal() == 2).all()
    for b in range(B):
        np.testing.assert_array_equal(eng.theta(b), orc.theta(b))
    assert eng.counters()[0] == orc.counters()[0]


@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA, abi.ALGO_DOUBLE_Q])
def test_td_shared_theta(algo):
    B = 32
    p, g, rec, eng, orc = make(depth=10, n_events=400, B=B, algo=algo, theta_mode=abi.THETA_SHARED, mem=1 << 20)
    eng.reset()
    orc.reset()
    for step in range(120):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "shared step %d" % step, exact=False, rtol=1e-9)
    if algo == abi.ALGO_DOUBLE_Q:
        np.testing.assert_allclose(eng.theta(1), orc.theta_b(), rtol=1e-9, atol=1e-12)
    th, oth = eng.theta(), orc.theta()
    assert np.array_equal(th != 0, oth != 0)
    np.testing.assert_allclose(th, oth, rtol=1e-9, atol=1e-12)  # north-star tolerance: 1e-5 relative


@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA, abi.ALGO_DOUBLE_Q, abi.ALGO_DOUBLE_R_LEARN])
@pytest.mark.parametrize("theta_mode,first_book", [(abi.THETA_PRIVATE, 100), (abi.THETA_SHARED, 0), (abi.THETA_SHARED, 7)])
def test_random_init_weights(algo, theta_mode, first_book):
    """learning.random_init: true (src/rl/agent.cpp:37-39; DoubleAgent, agent.cpp:190-192): theta -- then theta_b -- filled with
    2u - 1 from the agent's own std::mt19937_64, which goes on to toss the double agents' coin from where the fill left it.  The
    oracle does this with libstdc++'s generator and distribution exactly as the reference does (pinned against the reference
    itself by tests/test_oracle_ref_sweep.py); the engine draws the same numbers with its own restatement of the generator
    (lob_engine.hip theta_random_init).  Private theta: every book's agent its own vectors, bit for bit, coin included.  Shared
    theta: the vector of global book 0's agent on every shard (first_book 7: the same weights as first_book 0, and no book of the
    shard has spent draws on them)."""
    B = 6 if theta_mode == abi.THETA_PRIVATE else 48
    p, g, rec, eng, orc = make(n_events=300, B=B, algo=algo, theta_mode=theta_mode, mem=1 << 14, first_book=first_book, seed=(1 << 33) + 77,
                               random_init=1, beta=0.01)
    nt = B if theta_mode == abi.THETA_PRIVATE else 1
    double = algo in (abi.ALGO_DOUBLE_Q, abi.ALGO_DOUBLE_R_LEARN)
    for t in range(nt):
        th = eng.theta(t)
        np.testing.assert_array_equal(th, orc.theta(t))
        assert th.min() >= -1.0 and th.max() < 1.0 and abs(th.mean()) < 0.05 and np.count_nonzero(th) >= th.size - 2
        if double:
            np.testing.assert_array_equal(eng.theta(nt + t), orc.theta_b(t))
            assert not np.array_equal(eng.theta(nt + t), th)
    if theta_mode == abi.THETA_SHARED and first_book:
        p0 = engine.default_params()
        for f, _ in abi.Params._fields_:
            setattr(p0, f, getattr(p, f))
        p0.book_id_offset = 0
        e0 = engine.Engine(p0, 2)
        np.testing.assert_array_equal(e0.theta(), eng.theta())       # every shard starts from the same vector
        e0.close()
    eng.reset()
    orc.reset()
    exact = theta_mode == abi.THETA_PRIVATE
    for step in range(40):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "random_init algo %d step %d" % (algo, step), exact=exact, rtol=1e-9)
    for t in range(nt):
        if exact:
            np.testing.assert_array_equal(eng.theta(t), orc.theta(t))
        else:
            np.testing.assert_allclose(eng.theta(t), orc.theta(t), rtol=1e-9, atol=1e-12)
        if double:
            np.testing.assert_allclose(eng.theta(nt + t), orc.theta_b(t), rtol=0 if exact else 1e-9, atol=0 if exact else 1e-12)
    eng.close()
    orc.close()


@pytest.mark.parametrize("B,theta_mode,algo", [(1, abi.THETA_PRIVATE, abi.ALGO_SARSA), (3, abi.THETA_PRIVATE, abi.ALGO_DOUBLE_Q),
                                               (1500, abi.THETA_SHARED, abi.ALGO_QLAMBDA), (40000, abi.THETA_SHARED, abi.ALGO_SARSA)])
def test_model_log_rows(B, theta_mode, algo):
    """lob_model_log_enable / lob_model_log_read: the rows of the reference's `model_log` logger (Agent::HandleTransition,
    src/rl/agent.cpp:93-100 -- mean |delta| per 1000 updates; the oracle's rows are pinned on the reference's own by
    tests/test_oracle_golden.py).  One book: the reference's rows bit for bit; a few books: a row every time the running count
    reaches 1000; a batch of 1000 books or more: a row per learner step (the mean |delta| over the batch; the big batch also
    goes through the lane-per-book learner kernels and learn_q_rest_kernel)."""
    n_steps = 2300 if B <= 3 else (40 if B < 5000 else 12)
    p, g, rec, eng, orc = make(depth=5 if B <= 3 else 10, n_events=4200 if B <= 3 else 300, B=B, algo=algo, theta_mode=theta_mode,
                               mem=1 << 16 if B <= 3 else 1 << 20)
    eng.model_log_enable()
    eng.reset(); orc.reset()
    eng.td_step(n_steps); orc.td_step(n_steps)
    rows, lost = eng.model_log_read()
    want = orc.model_log()
    assert lost == 0 and len(rows) == len(want) == (n_steps * B // 1000 if B <= 3 else n_steps)
    if theta_mode == abi.THETA_PRIVATE and B == 1:
        np.testing.assert_array_equal(rows, want)
    else:
        np.testing.assert_allclose(rows, want, rtol=1e-9)
    assert np.all(rows > 0)
    eng.td_step(3); orc.td_step(3)
    rows2, _ = eng.model_log_read()          # (handed over once: only what was written since)
    np.testing.assert_allclose(rows2, orc.model_log()[len(want):], rtol=1e-9)
    eng.close()
    orc.close()


def test_eval_step_greedy():
    B = 8
    p, g, rec, eng, orc = make(n_events=400, B=B, theta_mode=abi.THETA_SHARED)
    rng = np.random.default_rng(5)
    th = rng.standard_normal(p.memory_size) * 1e-3
    eng.set_theta(th)
    orc.theta()[:] = th
    eng.reset()
    orc.reset()
    for step in range(60):
        eng.eval_step(1)
        orc.eval_step(1)
        compare_env(eng, orc, "eval step %d" % step)
        np.testing.assert_array_equal(eng.last_actions(), orc.recs()["action"])
    np.testing.assert_array_equal(eng.theta(), th)


def test_device_generator_matches_host():
    p = engine.default_params()
    p.depth = 10
    g = engine.default_gen_params()
    g.n_events = 200
    B = 16
    p.book_id_offset = 5
    rec = engine.gen_stream_host(g, 10, 2, 5, B)
    e1 = engine.Engine(p, B)
    e1.load_events(rec)
    e1.reset()
    e2 = engine.Engine(p, B)
    e2.gen_events(g)
    e2.reset()
    e1.td_step(50)
    e2.td_step(50)
    from tests.parity import dumps_to_np, assert_books_equal
    assert_books_equal(dumps_to_np(e1.get_books()), dumps_to_np(e2.get_books()), "host vs device generator")


def test_bad_stream_rejected():
    p, g, rec, eng, orc = make(n_events=50, B=2)
    bad = rec.copy()
    bad[1, 10, 2 + 5] = 0  # ask volume of level 0 -> 0 : reference throws (src/market/book.cpp:76)
    with pytest.raises(engine.LobError) as ei:
        eng.load_events(bad)
    assert ei.value.code == abi.LOB_EDATA


def test_step_before_reset_is_an_error():
    p = engine.default_params()
    eng = engine.Engine(p, 2)
    with pytest.raises(engine.LobError) as ei:
        eng.td_step(1)
    assert ei.value.code == abi.LOB_ESTATE


@pytest.mark.parametrize("stop", ["early", "exhausted"])
def test_two_episodes_window_sums_persist(stop):
    """Runner::RunEpisode twice on the same engine: window sums survive ClearWindows()
    (quirk Q7), traces are decayed by HandleTerminal, theta keeps learning, and the
    second episode's first step acts on the previous episode's leftover State (quirk Q19)."""
    B = 8
    p, g, rec, eng, orc = make(n_events=330, B=B, algo=abi.ALGO_QLAMBDA, theta_mode=abi.THETA_PRIVATE, first_book=40)
    n1 = 90 if stop == "early" else 300
    for ep in range(2):
        eng.reset()
        orc.reset()
        compare_env(eng, orc, "episode %d reset" % ep)
        for step in range(n1 if ep == 0 else 60):
            eng.td_step(1)
            orc.td_step(1)
            if step % 10 == 0 or step < 3:
                compare_learner_step(eng, orc, "episode %d step %d" % (ep, step))
        eng.clear_inventory()
        orc.clear_inventory()
        compare_env(eng, orc, "episode %d after ClearInventory" % ep)
        eng.handle_terminal()
        ol.load().oracle_handle_terminal(orc.h)
    for b in range(B):
        np.testing.assert_array_equal(eng.theta(b), orc.theta(b))


def _repack_trades(rec, depth, t_old, t_new):
    """records with t_old trade slots -> layout with t_new slots (extra slots empty)."""
    B, N, _ = rec.shape
    w_new = engine.record_words(depth, t_new)
    out = np.zeros((B, N, w_new), np.uint32)
    out[..., :2 + 4 * depth] = rec[..., :2 + 4 * depth]
    out[..., 2 + 4 * depth:2 + 4 * depth + t_old] = rec[..., 2 + 4 * depth:2 + 4 * depth + t_old]
    out[..., 2 + 4 * depth + t_new:2 + 4 * depth + t_new + t_old] = rec[..., 2 + 4 * depth + t_old:2 + 4 * depth + 2 * t_old]
    return out


@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA])
def test_same_timestamp_rows_and_crossed_books_batched(algo):
    """Quirk Q14 in the batched engine at depth 10: rows sharing a timestamp are applied inside one
    event, crossed (invalid) snapshots are skipped without re-stash, and the trades of the swallowed
    rows are handed to the next event (up to 4 price levels here)."""
    B, D, N = 24, 10, 420
    p = engine.default_params()
    p.depth, p.max_trades, p.algo, p.theta_mode, p.memory_size = D, 4, algo, abi.THETA_PRIVATE, 1 << 18
    g = engine.default_gen_params()
    g.n_events = N
    rec = _repack_trades(engine.gen_stream_host(g, D, 2, 0, B), D, 2, 4)
    rng = np.random.default_rng(11)
    o_ap, o_bp = 2, 2 + 2 * D
    for b in range(B):
        for r in rng.choice(np.arange(70, N - 5), size=10, replace=False):
            if rng.random() < 0.5:
                rec[b, r, 0] = rec[b, r - 1, 0]               # same timestamp as the previous row
            else:                                             # crossed book: asks below bids
                a = rec[b, r, o_ap:o_ap + D].copy()
                rec[b, r, o_ap:o_ap + D] = rec[b, r, o_bp:o_bp + D][::-1]
                rec[b, r, o_bp:o_bp + D] = a[::-1]
    # keep time monotone after the edits
    for b in range(B):
        t = rec[b, :, 0].astype(np.int64)
        assert (np.diff(t) >= 0).all()
    eng = engine.Engine(p, B)
    eng.load_events(rec)
    orc = ol.Oracle(p, rec)
    eng.reset()
    orc.reset()
    compare_env(eng, orc, "reset")
    for step in range(200):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "q14 step %d" % step)
    for b in range(B):
        np.testing.assert_array_equal(eng.theta(b), orc.theta(b))
    assert eng.counters()[1] == orc.counters()[1]


@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA])
def test_verdict_carry_over_is_invalidated(algo):
    """act_kernel reuses learn_kernel's 'weight was never written' verdicts for the same State
    (verdict carry-over).  A tiny weight table makes group-1/2 tiles collide with
    written weights all the time, and the sequence walks through everything that must void the
    saved verdicts: an externally driven step, lob_theta_set, a new episode."""
    B = 16
    p, g, rec, eng, orc = make(depth=5, n_events=500, B=B, algo=algo, theta_mode=abi.THETA_SHARED, mem=1 << 13)
    eng.reset()
    orc.reset()
    rng = np.random.default_rng(11)
    n = 0

    def td(k):
        nonlocal n
        for _ in range(k):
            eng.td_step(1)
            orc.td_step(1)
            compare_learner_step(eng, orc, "carry-over step %d" % n, exact=False, rtol=1e-9)
            n += 1

    td(25)
    acts = rng.integers(0, 9, size=B).astype(np.int32)
    eng.step(acts)
    orc.env_step(acts)
    compare_env(eng, orc, "after external step")
    td(10)
    th = eng.theta()
    th[rng.integers(0, th.size, size=2000)] += 1e-3  # writes weights the bitmap has never seen
    eng.set_theta(th)
    orc.theta()[:] = th
    td(10)
    eng.clear_inventory(); orc.clear_inventory()
    eng.handle_terminal(); orc.handle_terminal()
    eng.reset(); orc.reset()
    td(25)
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("algo", [abi.ALGO_QLAMBDA, abi.ALGO_DOUBLE_Q])
def test_verdict_carry_over_on_off_identical(monkeypatch, algo):
    """Same run with the carry-over switched off (LOB_NO_CARRY=1, read by lob_create): identical
    books, actions and weights, also across evaluation steps in the middle of training (a
    sequence the reference never produces, so there is no oracle for it)."""
    B = 16
    out = []
    for off in ("1", "0"):
        monkeypatch.setenv("LOB_NO_CARRY", off)
        p, g, rec, eng, orc = make(depth=5, n_events=500, B=B, algo=algo, theta_mode=abi.THETA_SHARED, mem=1 << 13)
        orc.close()
        eng.reset()
        trail = []
        for phase in range(4):
            for _ in range(15):
                eng.td_step(1)
                trail.append((eng.last_actions().copy(), bytes(eng.get_books())))
            eng.eval_step(3)
            trail.append((eng.last_actions().copy(), bytes(eng.get_books())))
        out.append((trail, eng.theta()))
        eng.close()
    for k, ((a0, b0), (a1, b1)) in enumerate(zip(out[0][0], out[1][0])):
        np.testing.assert_array_equal(a0, a1, err_msg="actions, record %d" % k)
        assert b0 == b1, "books differ at record %d" % k
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-9, atol=1e-12)


def test_config2_full_size_against_oracle():
    """BASELINE config 2 at its full size: 4 096 parallel 10-level books, SARSA(lambda), shared
    20M-weight table -- small enough for the oracle to follow step by step for a while."""
    B = 4096
    p, g, rec, eng, orc = make(depth=10, n_events=160, B=B, algo=abi.ALGO_SARSA, theta_mode=abi.THETA_SHARED, mem=20000000)
    eng.reset()
    orc.reset()
    for step in range(12):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "C2 step %d" % step, exact=False, rtol=1e-9)
    th, oth = eng.theta(), orc.theta()
    assert np.array_equal(th != 0, oth != 0)
    np.testing.assert_allclose(th, oth, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA])
def test_long_traces_use_the_64_generation_ring(algo):
    """gamma*lambda = 0.92: eligibilities stay above the 0.01 tolerance for 56 steps, beyond the
    32-generation ring the default configuration (25 generations) gets."""
    B = 4
    p, g, rec, eng, orc = make(depth=5, n_events=500, B=B, algo=algo, theta_mode=abi.THETA_PRIVATE, mem=1 << 16,
                               gamma=1.0, lambda_=0.92, epsilon=0.05)
    eng.reset()
    orc.reset()
    most = 0
    for step in range(150):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "long traces step %d" % step, exact=True)
        most = max(most, int(dumps_to_np(eng.get_books())["n_traces"].max()))
    assert most > 32 * 32  # more live traces than 32 generations could hold
    for b in range(B):
        ei, ee = eng.traces(b)
        oi, oe = orc.traces(b)
        assert dict(zip(ei.tolist(), ee.tolist())) == dict(zip(oi.tolist(), oe.tolist()))
        np.testing.assert_array_equal(eng.theta(b), orc.theta(b))
    with pytest.raises(engine.LobError):
        make(B=1, gamma=1.0, lambda_=0.95)   # 90 generations: more than the ring holds


@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA, abi.ALGO_DOUBLE_Q])
def test_combined_update_on_off_identical(monkeypatch, algo):
    """Shared theta sums the updates per distinct trace generation before touching theta
    (accumulate_kernel + apply_kernel); LOB_NO_COMBINE=1 applies them trace by trace (update_kernel).
    Same books, same actions; weights equal up to the order of the f64 additions.  A tiny weight
    table and many books in lock-step make shared generations the rule."""
    B = 64
    out = []
    for off in ("1", "0"):
        monkeypatch.setenv("LOB_NO_COMBINE", off)
        p, g, rec, eng, orc = make(depth=5, n_events=400, B=B, algo=algo, theta_mode=abi.THETA_SHARED, mem=1 << 13,
                                   epsilon=0.3)
        orc.close()
        eng.reset()
        trail = []
        for _ in range(80):
            eng.td_step(1)
            trail.append((eng.last_actions().copy(), bytes(eng.get_books())))
        out.append((trail, eng.theta(), eng.theta(1) if algo == abi.ALGO_DOUBLE_Q else None))
        eng.close()
    for k, ((a0, b0), (a1, b1)) in enumerate(zip(out[0][0], out[1][0])):
        np.testing.assert_array_equal(a0, a1, err_msg="actions, step %d" % k)
        assert b0 == b1, "books differ at step %d" % k
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-9, atol=1e-12)
    assert np.array_equal(out[0][1] != 0, out[1][1] != 0)
    if algo == abi.ALGO_DOUBLE_Q:
        np.testing.assert_allclose(out[0][2], out[1][2], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("acc_lanes", [0, 8, 16, 64])
def test_combined_update_table_overflow(monkeypatch, acc_lanes):
    """More distinct trace generations than the claim table has slots (300 books x up to 56 live
    generations against 2 048 slots): the generations that find no slot are applied directly.
    accumulate_kernel's lanes per book (LOB_ACC_LANES; 0 = the engine's own choice) walk a book's
    generations in rounds of 8 / 16 / 32 / 64: the shape must not show."""
    if acc_lanes:
        monkeypatch.setenv("LOB_ACC_LANES", str(acc_lanes))
    monkeypatch.setenv("LOB_CB_SLOTS", "2048")
    B = 300
    p, g, rec, eng, orc = make(depth=5, n_events=330, B=B, algo=abi.ALGO_SARSA, theta_mode=abi.THETA_SHARED, mem=1 << 16,
                               gamma=1.0, lambda_=0.92, epsilon=0.6)
    eng.reset()
    orc.reset()
    for step in range(90):
        eng.td_step(1)
        orc.td_step(1)
        if step % 10 == 9 or step < 3:
            compare_learner_step(eng, orc, "overflow step %d" % step, exact=False, rtol=1e-9)
    assert int(dumps_to_np(eng.get_books())["n_traces"].sum()) > 60 * 2048  # far more live generations than slots... times 32 tiles
    th, oth = eng.theta(), orc.theta()
    assert np.array_equal(th != 0, oth != 0)
    np.testing.assert_allclose(th, oth, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("mem", [1, 2, 2147483647])
def test_features_at_extreme_table_sizes(mem):
    """The device keeps the hash sum reduced mod M in 32 bits (s, x < M, s + x < 2^32): exercise the
    largest table lob_create accepts (2^31 - 1 weights, 17 GB) and the degenerate ones."""
    p = engine.default_params()
    p.memory_size = mem
    eng = engine.Engine(p, 1)
    rng = np.random.default_rng(3)
    v = rng.uniform(-20, 20, size=(64, 8)).astype(np.float32)
    got = eng.features(v)
    want = np.zeros_like(got)
    ol.load().oracle_tiles(mem, ol.ptr(v), 8, v.shape[0], ol.ptr(want))
    np.testing.assert_array_equal(got, want)
    assert got.min() >= 0 and got.max() < mem
    eng.close()


# ---- BASELINE's headline configuration against the oracle, at full size -----------------------------
def test_config3_full_size_against_oracle():
    """BASELINE config 3 -- the configuration bench.py times -- at its full size: 65 536 parallel
    10-level books, Q(lambda) with eligibility traces, ONE shared 20M-weight table.  Exactly the
    kernels and launch shapes of the timed run (env_kernel<64>, the shared-theta learner path with
    its claim / accumulate / apply update) followed step by step by the oracle: every book bit-exact
    (all 65 536 dumps, actions, rewards, state variables, RNG counters), TD errors and weights within
    1e-9 relative (f64 atomic ordering; north-star allows 1e-5)."""
    B = 65536
    p, g, rec, eng, orc = make(depth=10, n_events=160, B=B, algo=abi.ALGO_QLAMBDA, theta_mode=abi.THETA_SHARED, mem=20000000)
    eng.reset()
    orc.reset()
    for step in range(12):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "C3 step %d" % step, exact=False, rtol=1e-9)
    assert eng.counters()[0] == 12 * B
    th, oth = eng.theta(), orc.theta()
    assert np.array_equal(th != 0, oth != 0) and np.count_nonzero(th) > 50000
    np.testing.assert_allclose(th, oth, rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("reset_lanes", [16, 32, 64])
@pytest.mark.parametrize("env_lanes", [16, 32, 64, 256])
def test_books_per_wave_and_group_choices(monkeypatch, env_lanes, reset_lanes, groups):
    """Every instantiation of the lane-per-book kernels (env_kernel<16|32|64>, env_compact_kernel = 256,
    reset_kernel<16|32|64>)
    and both step pipelines (one group / two groups on two streams) against the oracle on 10-level
    books: the launch shape must not show in the results.  (lob_create reads the switches.)"""
    if (env_lanes in (32, 256) or reset_lanes in (16, 32) or groups == 2) and not experiments_build():
        pytest.skip("a kernel variant measured and lost: compiled with -DLOB_EXPERIMENTS only (tools/exp_variants.sh)")
    monkeypatch.setenv("LOB_ENV_LANES", str(env_lanes))
    monkeypatch.setenv("LOB_RESET_LANES", str(reset_lanes))
    monkeypatch.setenv("LOB_GROUPS", str(groups))
    B = 1100  # >= 1024: the two-group pipeline only splits batches of that size; not a multiple of any wave shape
    p, g, rec, eng, orc = make(depth=10, n_events=150, B=B, algo=abi.ALGO_QLAMBDA, theta_mode=abi.THETA_SHARED, mem=1 << 20)
    eng.reset()
    orc.reset()
    for step in range(10):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "lanes %d/%d groups %d step %d" % (env_lanes, reset_lanes, groups, step), exact=False, rtol=1e-9)
    # a second episode on the same streams: finalize_kernel + the pre-pass again
    eng.clear_inventory(); orc.clear_inventory()
    eng.handle_terminal(); orc.handle_terminal()
    eng.reset(); orc.reset()
    for step in range(4):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "lanes %d/%d groups %d episode 2 step %d" % (env_lanes, reset_lanes, groups, step), exact=False, rtol=1e-9)
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


def test_env_step_kernel_with_32_books_per_wave(monkeypatch):
    """LOB_ENV_STEP_LANES=32 (experiments build: env_step_kernel<false, false, 32>, two half-full waves per SIMD -- measured slower, 0.158
    against 0.098 ms): the fused action selection + step with the upper half of every wave idle, against the oracle (the
    idle lanes have a slot of their own)."""
    if not experiments_build():
        pytest.skip("a kernel variant measured and lost: compiled with -DLOB_EXPERIMENTS only (tools/exp_variants.sh)")
    monkeypatch.setenv("LOB_ENV_STEP_LANES", "32")
    monkeypatch.setenv("LOB_FUSE_ACT", "1")
    monkeypatch.setenv("LOB_Q_LANES", "1")
    B = 1100
    p, g, rec, eng, orc = make(depth=10, n_events=150, B=B, algo=abi.ALGO_QLAMBDA, theta_mode=abi.THETA_SHARED, mem=1 << 20)
    eng.reset(); orc.reset()
    for step in range(16):
        eng.td_step(1); orc.td_step(1)
        compare_learner_step(eng, orc, "32 books per wave, step %d" % step, exact=False, rtol=1e-9)
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


def test_prepass_on_two_waves_per_64_books(monkeypatch):
    """LOB_PREPASS_ROLES=1: the market pre-pass with the book side and the window side of every event on two waves of a block
    (reset2_kernel / prepass_extend2_kernel, lob_env.h prepass_run2 -- opt-in: measured slower than one wave).  Same track, same
    windows: engine against the oracle through two episodes, and through a stream longer than the track ring (the resumed
    pre-pass)."""
    if not experiments_build():
        pytest.skip("a kernel variant measured and lost: compiled with -DLOB_EXPERIMENTS only (tools/exp_variants.sh)")
    monkeypatch.setenv("LOB_PREPASS_ROLES", "1")
    B = 1100
    p, g, rec, eng, orc = make(depth=10, n_events=150, B=B, algo=abi.ALGO_QLAMBDA, theta_mode=abi.THETA_SHARED, mem=1 << 20)
    for episode in range(2):
        eng.reset(); orc.reset()
        for step in range(8):
            eng.td_step(1); orc.td_step(1)
            compare_learner_step(eng, orc, "two-wave pre-pass episode %d step %d" % (episode, step), exact=False, rtol=1e-9)
        eng.clear_inventory(); orc.clear_inventory()
        eng.handle_terminal(); orc.handle_terminal()
    eng.close(); orc.close()
    monkeypatch.setenv("LOB_TRACK_RING", "256")
    monkeypatch.setenv("LOB_TRACK_REFILL", "16")
    p, g, rec, eng, orc = make(depth=5, n_events=900, B=70, algo=abi.ALGO_SARSA, theta_mode=abi.THETA_SHARED, mem=1 << 18)
    eng.reset(); orc.reset()
    for step in range(260):
        eng.td_step(1); orc.td_step(1)
        if step % 20 == 19:
            compare_learner_step(eng, orc, "two-wave pre-pass, ring, step %d" % step, exact=False, rtol=1e-9)
    eng.close(); orc.close()


@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA])
def test_group0_memo_on_off_identical(monkeypatch, algo):
    """Shared theta evaluates the group-0 part of Q once per distinct (inventory, quote distances)
    triple and continues every book's ordered sum from there (memo_kernel + q_values_memo);
    LOB_NO_MEMO=1 evaluates all 128 terms per book and action (q_values).  Both are the reference's
    sum term by term, so everything must agree bit for bit except theta's f64 atomic ordering --
    also across evaluation steps, external actions, a weight load and a second episode.  A small
    table makes hash collisions between group-0 and group-1/2 tiles the rule, which is what the
    sparse continuation has to get right."""
    B = 96
    out = []
    for off in ("1", "0"):
        monkeypatch.setenv("LOB_NO_MEMO", off)
        p, g, rec, eng, orc = make(depth=5, n_events=600, B=B, algo=algo, theta_mode=abi.THETA_SHARED, mem=1 << 12, epsilon=0.3)
        orc.close()
        eng.reset()
        trail = []
        rng = np.random.default_rng(17)
        for phase in range(3):
            for _ in range(25):
                eng.td_step(1)
                trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books())))
            eng.eval_step(3)
            trail.append((eng.last_actions().copy(), None, bytes(eng.get_books())))
            eng.step(rng.integers(0, 9, size=B).astype(np.int32))
            th = eng.theta()
            th[rng.integers(0, th.size, size=50)] += 1e-3
            eng.set_theta(th)
        eng.clear_inventory()
        eng.handle_terminal()
        eng.reset()
        for _ in range(20):
            eng.td_step(1)
            trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books())))
        out.append((trail, eng.theta()))
        eng.close()
    for k, ((a0, t0, b0), (a1, t1, b1)) in enumerate(zip(out[0][0], out[1][0])):
        np.testing.assert_array_equal(a0, a1, err_msg="actions, record %d" % k)
        assert b0 == b1, "books differ at record %d" % k
        if t0 is not None:
            np.testing.assert_allclose(t0, t1, rtol=1e-9, atol=1e-12, err_msg="td, record %d" % k)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-9, atol=1e-12)


def _light_counts(eng):
    """(books whose action came from act_light_kernel so far, step id of the last update that voided the hit lists)"""
    import ctypes
    out = (ctypes.c_int64 * 2)()
    eng.lib.lob_debug_light.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
    assert eng.lib.lob_debug_light(eng.h, out) == 0
    return int(out[0]), int(out[1])


@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA])
@pytest.mark.parametrize("mem", [1 << 12, 1 << 16, 1 << 22])
def test_hit_list_carry_on_off_identical(monkeypatch, algo, mem):
    """learn_q_fast_kernel leaves, per book, the ordered list of additions beyond the memoised group-0 sum;
    the next step's act_light_kernel replays it under the new weights instead of hashing and filtering the
    576 group-1/2 tiles again (the trace kernel marks a new generation's tiles before the learn kernel
    looks, so the set of tiles on a marked weight cannot change in between).  LOB_NO_LIGHT=1 keeps the full
    act kernel: everything must agree bit for bit except theta's f64 atomic ordering -- across evaluation
    steps, external actions, weight loads, a second episode.  Table sizes: 4 096 and 65 536 weights (most
    lists overflow: general path), 4 M (short lists: the light path serves nearly every book)."""
    B = 96
    out = []
    for off in ("1", "0"):
        monkeypatch.setenv("LOB_NO_LIGHT", off)
        p, g, rec, eng, orc = make(depth=5, n_events=700, B=B, algo=algo, theta_mode=abi.THETA_SHARED, mem=mem, epsilon=0.3)
        orc.close()
        eng.reset()
        trail = []
        rng = np.random.default_rng(23)
        for phase in range(3):
            for n in (1, 1, 3, 1, 7, 12):
                eng.td_step(n)
                trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books())))
            eng.eval_step(2)
            trail.append((eng.last_actions().copy(), None, bytes(eng.get_books())))
            eng.td_step(4)
            trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books())))
            eng.step(rng.integers(0, 9, size=B).astype(np.int32))
            eng.td_step(3)
            th = eng.theta()
            th[rng.integers(0, th.size, size=50)] += 1e-3
            if phase == 1:
                th[:] = 0.0  # every written weight back to +0.0: the maps must not forget the live traces' tiles
            eng.set_theta(th)
            eng.td_step(5)
            trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books())))
        eng.clear_inventory()
        eng.handle_terminal()
        eng.reset()
        for _ in range(20):
            eng.td_step(1)
            trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books())))
        light, dirty = _light_counts(eng)
        out.append((trail, eng.theta(), light, dirty))
        eng.close()
    assert out[0][2] == 0, "LOB_NO_LIGHT=1 must never launch act_light_kernel"
    assert out[1][3] == -1, "an update set a map bit the trace kernel had not: step %d" % out[1][3]
    if mem >= 1 << 22:
        assert out[1][2] > 40 * B, "the light path was hardly used: %d book-steps" % out[1][2]
    for k, ((a0, t0, b0), (a1, t1, b1)) in enumerate(zip(out[0][0], out[1][0])):
        np.testing.assert_array_equal(a0, a1, err_msg="actions, record %d" % k)
        assert b0 == b1, "books differ at record %d" % k
        if t0 is not None:
            np.testing.assert_allclose(t0, t1, rtol=1e-9, atol=1e-12, err_msg="td, record %d" % k)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("pair", ["0", "1"])
@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA])
@pytest.mark.parametrize("mem,n_vars", [(1 << 12, 8), (1 << 18, 8), (1 << 22, 8), (1 << 20, 13), (1 << 20, 4)])
def test_q_lane_kernel_on_off_identical(monkeypatch, algo, mem, n_vars, pair):
    """learn_q_lane_kernel (a lane per book: it walks the book's 64 group-1/2 tilings itself) against
    learn_q_fast_kernel (a wave per book, tilings spread over the lanes): same tile indices, same hit lists, same
    ordered sums -- actions, TD errors and books bit for bit, theta up to its atomics' ordering.  With 13 and with
    4 state variables (group 1 then hashes 10 / 1 of them), and with tables small enough for long lists.
    `pair` 1: learn_q_pair_kernel, two lanes per book (one walks the group-1 tilings, the other the group-2 ones)."""
    B = 160
    out = []
    monkeypatch.setenv("LOB_Q_PAIR", pair)
    for lanes in ("0", "1"):
        monkeypatch.setenv("LOB_Q_LANES", lanes)
        p = engine.default_params()
        p.depth, p.max_trades, p.algo, p.theta_mode, p.memory_size, p.epsilon = 5, 2, algo, abi.THETA_SHARED, mem, 0.3
        if n_vars != 8:
            # the first three stay pos / a_dist / b_dist (the group-0 triple); 13: every variable, 4: one market variable
            order = list(p.vars[:3]) + [v for v in range(13) if v not in list(p.vars[:3])]
            p.n_vars = n_vars
            for i in range(13):
                p.vars[i] = order[i] if i < n_vars else 0
        g = engine.default_gen_params()
        g.n_events = 500
        rec = engine.gen_stream_host(g, 5, 2, 0, B)
        eng = engine.Engine(p, B)
        eng.load_events(rec)
        eng.reset()
        trail = []
        for n in (1, 1, 2, 5, 1, 9, 14, 3, 25):
            eng.td_step(n)
            trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books())))
        th = eng.theta()
        th[::7] += 1e-3
        eng.set_theta(th)
        for n in (1, 6, 11):
            eng.td_step(n)
            trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books())))
        out.append((trail, eng.theta()))
        eng.close()
    for k, ((a0, t0, b0), (a1, t1, b1)) in enumerate(zip(out[0][0], out[1][0])):
        np.testing.assert_array_equal(a0, a1, err_msg="actions, record %d" % k)
        assert b0 == b1, "books differ at record %d" % k
        np.testing.assert_allclose(t0, t1, rtol=1e-9, atol=1e-12, err_msg="td, record %d" % k)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("pair", ["0", "1"])
@pytest.mark.parametrize("algo", [abi.ALGO_QLAMBDA, abi.ALGO_DOUBLE_Q])
@pytest.mark.parametrize("mem,slots", [(1 << 12, 0), (1 << 16, 256), (1 << 20, 0)])
def test_rest_kernel_merged_against_oracle_and_split(monkeypatch, algo, mem, slots, pair):
    """The fused Q(lambda) / double Q flow's rare paths in ONE launch (trace_rest_kernel: a handed-back book's TD error, a handed-on
    book's trace step, their sums) against the oracle step by step, and against the three launches it replaces (LOB_REST_MERGE=0).
    A 4 096-weight table hands most books back AND on; a 256-slot combine table leaves most generations without a slot -- in
    the merged kernel those wait for apply_kernel (other waves read theta meanwhile: the fold of round 5 that applied them on the
    spot was off by 6e-3 in one book's TD error at step 9 of exactly this configuration), counted by lob_debug_deferred."""
    B = 192
    monkeypatch.setenv("LOB_Q_LANES", "1")
    monkeypatch.setenv("LOB_FUSE_ACT", "1")   # (double Q's fast path at fewer than 1 024 books)
    monkeypatch.setenv("LOB_Q_PAIR", pair)
    if slots:
        monkeypatch.setenv("LOB_CB_SLOTS", str(slots))
    out = []
    for merge in ("1", "0"):
        monkeypatch.setenv("LOB_REST_MERGE", merge)
        p, g, rec, eng, orc = make(depth=5, n_events=500, B=B, algo=algo, theta_mode=abi.THETA_SHARED, mem=mem, epsilon=0.3)
        eng.reset()
        orc.reset()
        trail = []
        for step in range(40):
            eng.td_step(1)
            orc.td_step(1)
            compare_learner_step(eng, orc, "merge %s step %d" % (merge, step), exact=False, rtol=1e-9)
            trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books())))
        np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
        ps, fl, dg = eng.path_stats(), eng.flow_stats(), eng.deferred_generations()
        print("merge", merge, "handed on", int(ps[0]), "handed back", int(ps[7]), "deferred generations", dg, fl)
        assert fl["added_in_place"] >= 38, fl
        if merge == "1":
            if mem == 1 << 12:
                assert ps[7] > 4 * B and ps[0] > 0, ps      # books handed back in every step, some handed on as well
            if slots:
                assert dg > B, dg                           # generations without a slot: applied by apply_kernel
        else:
            assert dg == 0
        out.append((trail, eng.theta()))
        eng.close()
        orc.close()
    for k, ((a0, t0, b0), (a1, t1, b1)) in enumerate(zip(out[0][0], out[1][0])):
        np.testing.assert_array_equal(a0, a1, err_msg="actions, step %d" % k)
        assert b0 == b1, "books differ at step %d" % k
        np.testing.assert_allclose(t0, t1, rtol=1e-9, atol=1e-12, err_msg="td, step %d" % k)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("lanes", ["0", "1"])
@pytest.mark.parametrize("mem,eps", [(1 << 12, 0.8), (1 << 22, 0.8), (1 << 22, 0.1)])
def test_trace_light_kernel_on_off_identical(monkeypatch, mem, eps, lanes):
    """Q(lambda): trace_light_kernel (a lane per book) serves the books whose step leaves no older trace generation
    behind -- the new generation copied from the memo slot's tile record -- and hands the others to the wave-per-book
    kernel; LOB_NO_TLIGHT=1 sends every book there.  Actions, TD errors, books and trace lists bit for bit; theta up
    to its atomics' ordering.  A 4 096-weight table makes colliding tiles the rule (no slot is ever "known distinct").
    `lanes` 1: with the lane-per-book learn kernel the light trace step is part of THAT kernel (learn_q_lane_kernel<.., TR>)
    and the wave-per-book trace kernel runs after it on the books it left (trace_fast_kernel<.., 2>)."""
    B = 160
    out = []
    for off in ("1", "0"):
        monkeypatch.setenv("LOB_NO_TLIGHT", off)
        monkeypatch.setenv("LOB_Q_LANES", "0" if off == "1" else lanes)
        p, g, rec, eng, orc = make(depth=5, n_events=600, B=B, algo=abi.ALGO_QLAMBDA, theta_mode=abi.THETA_SHARED, mem=mem, epsilon=eps)
        orc.close()
        eng.reset()
        trail = []
        rng = np.random.default_rng(5)
        for phase in range(2):
            for n in (1, 1, 2, 5, 1, 9, 14, 3):
                eng.td_step(n)
                tr = [tuple(sorted(zip(*[x.tolist() for x in eng.traces(b)]))) for b in (0, 7, B - 1)]
                trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books()), tr))
            eng.step(rng.integers(0, 9, size=B).astype(np.int32))
            th = eng.theta()
            th[::5] += 1e-3
            eng.set_theta(th)
        eng.reset()
        for n in (1, 4, 10):
            eng.td_step(n)
            trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books()), None))
        out.append((trail, eng.theta()))
        eng.close()
    for k, ((a0, t0, b0, r0), (a1, t1, b1, r1)) in enumerate(zip(out[0][0], out[1][0])):
        np.testing.assert_array_equal(a0, a1, err_msg="actions, record %d" % k)
        assert b0 == b1, "books differ at record %d" % k
        assert r0 == r1, "trace lists differ at record %d" % k
        np.testing.assert_allclose(t0, t1, rtol=1e-9, atol=1e-12, err_msg="td, record %d" % k)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA])
@pytest.mark.parametrize("mem,eps", [(1 << 12, 0.8), (1 << 16, 0.3), (1 << 22, 0.8), (1 << 22, 0.1)])
def test_trace_lane_kernel_on_off_identical(monkeypatch, mem, eps, algo):
    """SARSA(lambda), and the books of Q(lambda) whose traces survive the step (the lane-per-book learn kernel forced on
    for this small batch: LOB_Q_LANES=1): trace_lane_kernel (a lane per trace generation: which tiles a generation loses to the new state
    is decided from the quantised coordinates, index coincidences through the tile registry) against the wave-per-book
    kernel that compares the indices themselves (LOB_TRACE_LANES=0).  Actions, TD errors, books and the trace lists of
    every book bit for bit; theta up to its atomics' ordering.  With 4 096 / 65 536 weights most indices are shared by
    several different tiles (every generation takes the index-by-index path, many books go back to the wave kernel);
    with 4 M few are.  Two episodes (the registry and the memo table start afresh, old generations carry the old
    epoch), external steps and a lob_theta_set in between."""
    B = 192
    out = []
    for on in ("0", "1"):
        monkeypatch.setenv("LOB_TRACE_LANES", on)
        monkeypatch.setenv("LOB_Q_LANES", "1")
        p, g, rec, eng, orc = make(depth=5, n_events=600, B=B, algo=algo, theta_mode=abi.THETA_SHARED, mem=mem, epsilon=eps)
        orc.close()
        eng.reset()
        trail = []
        rng = np.random.default_rng(5)
        for phase in range(2):
            for n in (1, 1, 2, 5, 1, 9, 14, 3, 30):
                eng.td_step(n)
                tr = [tuple(sorted(zip(*[x.tolist() for x in eng.traces(b)]))) for b in range(0, B, 7)]
                trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books()), tr))
            eng.step(rng.integers(0, 9, size=B).astype(np.int32))
            th = eng.theta()
            th[::5] += 1e-3
            eng.set_theta(th)
        eng.reset()  # (no HandleTerminal: the generations of the first episode are still there)
        for n in (1, 4, 10, 40):
            eng.td_step(n)
            tr = [tuple(sorted(zip(*[x.tolist() for x in eng.traces(b)]))) for b in range(0, B, 7)]
            trail.append((eng.last_actions().copy(), eng.last_td().copy(), bytes(eng.get_books()), tr))
        out.append((trail, eng.theta()))
        eng.close()
    for k, ((a0, t0, b0, r0), (a1, t1, b1, r1)) in enumerate(zip(out[0][0], out[1][0])):
        np.testing.assert_array_equal(a0, a1, err_msg="actions, record %d" % k)
        assert b0 == b1, "books differ at record %d" % k
        assert r0 == r1, "trace lists differ at record %d" % k
        np.testing.assert_allclose(t0, t1, rtol=1e-9, atol=1e-12, err_msg="td, record %d" % k)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("squeeze", ["LOB_OW_SLOTS=4096", "LOB_AMB_CAP=3", "LOB_CB_SLOTS=256"])
@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA])
def test_trace_lane_kernel_without_room(monkeypatch, algo, squeeze):
    """The lane trace path when its tables have no room, against the oracle: a tile registry of 4 096 entries (a memo
    slot brings 288: most slots stay unregistered and their books go to the wave-per-book kernel), a list of three new
    ambiguous indices per step (it overflows in the first steps of each episode: the lane kernel is off until the next
    reset), a combine table of 256 slots (most generations find none: the direct per-tile update; the slots that exist
    persist from step to step)."""
    k, v = squeeze.split("=")
    monkeypatch.setenv(k, v)
    monkeypatch.setenv("LOB_Q_LANES", "1")
    B = 96
    p, g, rec, eng, orc = make(depth=5, n_events=500, B=B, algo=algo, theta_mode=abi.THETA_SHARED, mem=1 << 15, epsilon=0.3)
    for episode in range(2):
        eng.reset()
        orc.reset()
        for step in range(60):
            eng.td_step(1)
            orc.td_step(1)
            compare_learner_step(eng, orc, "%s episode %d step %d" % (squeeze, episode, step), exact=False, rtol=1e-9)
        eng.handle_terminal(); orc.handle_terminal()
    st = eng.path_stats()
    if k == "LOB_AMB_CAP":
        assert st[4] == 1, "the list of new ambiguous indices must have overflowed"
    if k != "LOB_CB_SLOTS":
        assert st[0] > 0, "books must have been handed on to the wave-per-book kernel"
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


@pytest.mark.parametrize("algo", [abi.ALGO_R_LEARN, abi.ALGO_ONLINE_R_LEARN, abi.ALGO_DOUBLE_R_LEARN])
@pytest.mark.parametrize("theta_mode", [abi.THETA_PRIVATE, abi.THETA_SHARED])
def test_r_learning_against_oracle(algo, theta_mode):
    """rl::RLearn / rl::OnlineRLearn / rl::DoubleRLearn (src/rl/agent.cpp:357-467): TD error without discount against the average reward
    rho, and rho's own update after updateQ -- conditional on maxQ(from_state) under the NEW weights (rho_kernel).  The
    oracle reproduces two reference trajectories of these agents (tests/golden/traj_rlearn_b24, traj_online_rlearn_b25);
    here the engine follows the oracle step by step: private weights bit for bit (rho included), shared weights up to
    the order of the atomic additions."""
    import ctypes
    B = 24
    p, g, rec, eng, orc = make(depth=5, n_events=500, B=B, algo=algo, theta_mode=theta_mode, mem=1 << 16, epsilon=0.4, beta=0.02)
    eng.reset()
    orc.reset()
    n_rho = B if theta_mode == abi.THETA_PRIVATE else 1
    eng.lib.lob_debug_rho.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
    exact = theta_mode == abi.THETA_PRIVATE
    for step in range(150):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, tag="r-learning step %d" % step, exact=exact, rtol=1e-9)
        if step % 10 == 9:
            er, orr = np.zeros(n_rho), np.zeros(n_rho)
            assert eng.lib.lob_debug_rho(eng.h, ol.ptr(er), n_rho) == 0
            assert orc.lib.oracle_get_rho(orc.h, ol.ptr(orr), n_rho) == 0
            assert np.any(orr != 0.0) or step < 20, "rho never moved: the test exercises nothing"
            if exact:
                np.testing.assert_array_equal(er, orr, err_msg="rho, step %d" % step)
            else:
                np.testing.assert_allclose(er, orr, rtol=1e-9, atol=1e-12, err_msg="rho, step %d" % step)
    for b in range(n_rho if exact else 1):
        if exact:
            np.testing.assert_array_equal(eng.theta(b), orc.theta(b))
        else:
            np.testing.assert_allclose(eng.theta(b), orc.theta(b), rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


# ---- long streams: the market track as a ring, refilled while the episode runs ------------------------
@pytest.mark.parametrize("ring,refill", [(256, 8), (512, 40)])
def test_long_streams_use_a_track_ring(monkeypatch, ring, refill):
    """A stream longer than LOB_TRACK_RING keeps only a ring of the latest track entries per book; the
    pre-pass runs on every LOB_TRACK_REFILL steps (prepass_extend_kernel).  Nothing may show in the
    results: every step against the oracle, through the end of the stream, a second episode (the window
    sums replayed to where the first one stopped, quirk Q7) and external actions."""
    monkeypatch.setenv("LOB_TRACK_RING", str(ring))
    monkeypatch.setenv("LOB_TRACK_REFILL", str(refill))
    B = 40
    p, g, rec, eng, orc = make(depth=10, n_events=1400, B=B, algo=abi.ALGO_QLAMBDA, theta_mode=abi.THETA_SHARED, mem=1 << 18)
    eng.reset()
    orc.reset()
    step = 0
    while eng.counters()[2] > 0 and step < 1500:
        n = 1 if step < 200 or step % 7 == 0 else 5
        eng.td_step(n)
        orc.td_step(n)
        step += n
        if n == 1:
            compare_learner_step(eng, orc, "ring %d step %d" % (ring, step), exact=False, rtol=1e-9)
    assert eng.counters()[2] == 0 and step > 400           # every book ran its stream dry, far beyond one ring
    compare_env(eng, orc, "ring: end of episode 1")
    eng.clear_inventory(); orc.clear_inventory()
    eng.handle_terminal(); orc.handle_terminal()
    eng.reset(); orc.reset()
    rng = np.random.default_rng(1)
    for step in range(60):
        if step % 9 == 8:
            a = rng.integers(0, 9, size=B).astype(np.int32)
            eng.step(a)
            orc.env_step(a)
            compare_env(eng, orc, "ring: episode 2 external step %d" % step)
        else:
            eng.td_step(1)
            orc.td_step(1)
            compare_learner_step(eng, orc, "ring: episode 2 step %d" % step, exact=False, rtol=1e-9)
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


def test_track_ring_underrun_is_reported(monkeypatch):
    """A ring that is not refilled in time must not go unnoticed: the run is void and the next
    synchronising call says so."""
    monkeypatch.setenv("LOB_TRACK_RING", "256")
    monkeypatch.setenv("LOB_TRACK_REFILL", "100000")
    p, g, rec, eng, orc = make(depth=5, n_events=1200, B=8)
    orc.close()
    eng.reset()
    with pytest.raises(engine.LobError, match="ring underrun"):
        eng.td_step(400)
        eng.sync()
    eng.close()
