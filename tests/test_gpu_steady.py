"""The kernels bench.py TIMES, against the oracle, in the states a long run actually reaches.

tests/test_gpu_parity.py follows the headline configuration for a dozen steps from a reset; the
instantiations the bench spends its time in (env_kernel<64, 2, 1> with the fused action selection,
learn_q_pair_kernel / learn_q_lane_kernel, trace_fast_kernel<., 2>, the memo pair, the hit lists)
only reach their steady state later: long hit lists, several live trace generations, written-weights
maps filling up, lists voided by a weight exchange, a second episode.  Here they are compared with
the oracle step by step through exactly that (reference: Learner::_step / Runner::RunEpisode,
src/experiment/serial.cpp:18-34,53-70).  The oracle's read phase runs on the host's cores
(ORACLE_THREADS, tests/conftest.py); the engine is the thing checked."""
import ctypes as C
import os

import numpy as np
import pytest

from rl_markets_amd import abi, engine
from tests import oracle_lib as ol
from tests.parity import compare_learner_step
from tests.test_gpu_fuzz import random_case

pytestmark = pytest.mark.gpu


def light_books(eng):
    """Books whose action came from the hit-list replay so far (lob_debug_light, a diagnostic export)."""
    import ctypes
    out = (ctypes.c_int64 * 2)()
    eng.lib.lob_debug_light.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
    assert eng.lib.lob_debug_light(eng.h, out) == 0
    return int(out[0])


def make(B, algo, n_events, mem=20000000, depth=10, **over):
    p = engine.default_params()
    p.depth, p.max_trades = depth, 2
    p.algo, p.theta_mode, p.memory_size = algo, abi.THETA_SHARED, mem
    for k, v in over.items():
        setattr(p, k, v)
    g = engine.default_gen_params()
    g.n_events = n_events
    rec = engine.gen_stream_host(g, depth, 2, 0, B)
    eng = engine.Engine(p, B)
    eng.load_events(rec)
    orc = ol.Oracle(p, rec)
    return p, eng, orc


@pytest.mark.parametrize("algo", [abi.ALGO_QLAMBDA, abi.ALGO_SARSA, abi.ALGO_DOUBLE_Q], ids=["qlambda", "sarsa", "double_q"])
def test_steady_state_at_the_dispatch_threshold_across_an_exchange(algo, tmp_path):
    """32 768 books (the smallest batch that takes the lane-per-book learner kernels and the fused
    env kernel by itself -- no switches), D = 10, M = 20 M, one shared weight vector, 70 steps against
    the oracle; a one-rank lob_theta_allreduce (RCCL on the engine's buffers: theta must not change) INSIDE step 32
    (lob_td_step_begin / _end: where the product places it -- no hit list is live there, nothing is voided) and another
    BETWEEN steps 64 and 65 (every hit list voided, every memo record re-stamped: step 65 runs act_fast_kernel and steps
    66+ are back on the light path); then
    ClearInventory / HandleTerminal / Initialise and 4 steps of a second episode."""
    from rl_markets_amd.comm import RcclComm
    from rl_markets_amd.parallel import EngineBackend
    B = 32768
    p, eng, orc = make(B, algo, n_events=330)
    comm = RcclComm(str(tmp_path / "rdzv"), 0, 1, 0)
    eng.delta_init()
    eng.reset()
    orc.reset()
    light0 = light_books(eng)
    for step in range(70):
        if step == 31:
            # the exchange where ShardedLearner puts it: inside the step, between its two halves -- nothing is voided
            eng.td_step_begin(); orc.td_step_begin()
            before = eng.theta()
            comm.sync_weights(EngineBackend(eng))
            eng.sync()
            np.testing.assert_array_equal(eng.theta(), before)
            eng.td_step_end(); orc.td_step_end()
        else:
            eng.td_step(1)
            orc.td_step(1)
        # every step around the start and the exchange, every 4th in between (a comparison moves 40 MB of dumps)
        if step < 4 or step % 4 == 3 or 60 <= step or 30 <= step <= 34:
            compare_learner_step(eng, orc, "steady %d step %d" % (algo, step), exact=False, rtol=1e-9)
        if step == 63:
            before = eng.theta()
            comm.sync_weights(EngineBackend(eng))
            eng.sync()
            # (theta_sync + (theta - theta_sync): the identity up to rounding once theta_sync is no longer the initial zeros)
            np.testing.assert_allclose(eng.theta(), before, rtol=1e-12, atol=1e-16)
    # the light path really was the one running: from step 3 on every live book's action comes from its hit list
    # (minus the step after the exchange)
    assert light_books(eng) - light0 > 60 * B * 0.9
    th, oth = eng.theta(), orc.theta()
    assert np.array_equal(th != 0, oth != 0) and np.count_nonzero(th) > (25000 if algo == abi.ALGO_DOUBLE_Q else 50000)
    np.testing.assert_allclose(th, oth, rtol=1e-9, atol=1e-12)
    if algo == abi.ALGO_DOUBLE_Q:  # DoubleAgent::theta_b: the coin sends about half of the updates there
        thb, othb = eng.theta(1), orc.theta_b()
        assert np.array_equal(thb != 0, othb != 0) and np.count_nonzero(thb) > 25000
        np.testing.assert_allclose(thb, othb, rtol=1e-9, atol=1e-12)
    eng.clear_inventory(); orc.clear_inventory()
    eng.handle_terminal(); orc.handle_terminal()
    eng.reset(); orc.reset()
    for step in range(4):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "steady %d episode 2 step %d" % (algo, step), exact=False, rtol=1e-9)
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    if algo == abi.ALGO_DOUBLE_Q:
        np.testing.assert_allclose(eng.theta(1), orc.theta_b(), rtol=1e-9, atol=1e-12)
    comm.close()
    eng.close()
    orc.close()


@pytest.mark.parametrize("algo", [abi.ALGO_QLAMBDA, abi.ALGO_SARSA], ids=["qlambda", "sarsa"])
def test_lane_learner_kernel_for_large_tables(algo):
    """learn_q_lane_kernel (one lane per book) is what tables of 2^27 weights and more take -- the pair kernel's
    LDS rows hold tile indices in 27 bits.  M = 2^27 + 1, 32 768 books, against the oracle."""
    B = 32768
    p, eng, orc = make(B, algo, n_events=160, mem=(1 << 27) + 1)
    eng.reset()
    orc.reset()
    for step in range(10):
        eng.td_step(1)
        orc.td_step(1)
        if step < 3 or step >= 7:
            compare_learner_step(eng, orc, "lane %d step %d" % (algo, step), exact=False, rtol=1e-9)
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


@pytest.mark.parametrize("batches", ["1", "4"])
def test_mostly_greedy_q_lambda_block_sums(monkeypatch, batches):
    """Q(lambda) late in the epsilon schedule (0.05: hardly any Watkins cut, books keep their generations like SARSA's): the
    combined update goes through accumulate_block_kernel -- sums per slot in a block's LDS table first, one or several batches
    of 1 024 books per block (LOB_ACC_BATCHES) -- 32 768 books, 30 steps against the oracle, and the flow counter says so."""
    monkeypatch.setenv("LOB_ACC_BATCHES", batches)
    monkeypatch.setenv("LOB_ACC_DENSE", "0")     # (the hash-table kernel; the dense-id kernel that replaced it by default: below)
    B = 32768
    p, eng, orc = make(B, abi.ALGO_QLAMBDA, n_events=200, epsilon=0.05)
    eng.reset()
    orc.reset()
    for step in range(30):
        eng.td_step(1)
        orc.td_step(1)
        if step < 3 or step % 9 == 0 or step >= 27:
            compare_learner_step(eng, orc, "greedy batches %s step %d" % (batches, step), exact=False, rtol=1e-9)
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    flow = eng.flow_stats()
    assert flow["block_sums"] == 30 and flow["added_in_place"] == 0 and flow["every_book"] == 0 and flow["dense_sums"] == 0, flow
    eng.close()
    orc.close()


@pytest.mark.parametrize("ids", ["all", "64", "off"])
@pytest.mark.parametrize("algo,eps", [(abi.ALGO_SARSA, 0.8), (abi.ALGO_QLAMBDA, 0.05)], ids=["sarsa", "greedy_qlambda"])
def test_block_sums_by_dense_slot_ids(monkeypatch, algo, eps, ids):
    """accumulate_dense_kernel: SARSA(lambda)'s update (every book keeps all its generations: 25 terms per book and step onto a
    few thousand combine slots) and mostly-greedy Q(lambda)'s, summed per block in an LDS array indexed by the slots' dense
    ids and added up by apply_kernel -- 16 384 books, 40 steps and the start of a second episode against the oracle.  `64`: only
    64 ids exist (LOB_CBD_IDS), so most slots go without one and take the atomics on their sums, in the same launch; `off`
    (LOB_ACC_DENSE=0): accumulate_block_kernel's hash table, the kernel this one replaced."""
    if ids == "off":
        monkeypatch.setenv("LOB_ACC_DENSE", "0")
    elif ids != "all":
        monkeypatch.setenv("LOB_CBD_IDS", ids)
    B = 16384
    p, eng, orc = make(B, algo, n_events=260, epsilon=eps)
    for episode, n_steps in ((0, 40), (1, 6)):
        eng.reset(); orc.reset()
        for step in range(n_steps):
            eng.td_step(1); orc.td_step(1)
            if step < 3 or step % 6 == 0 or step >= n_steps - 2:
                compare_learner_step(eng, orc, "dense ids %s algo %d episode %d step %d" % (ids, algo, episode, step), exact=False, rtol=1e-9)
        eng.clear_inventory(); orc.clear_inventory()
        eng.handle_terminal(); orc.handle_terminal()
    th, oth = eng.theta(), orc.theta()
    assert np.array_equal(th != 0, oth != 0) and np.count_nonzero(th) > 10000
    np.testing.assert_allclose(th, oth, rtol=1e-9, atol=1e-12)
    flow = eng.flow_stats()
    assert flow["block_sums"] == 46 and flow["dense_sums"] == (0 if ids == "off" else 46), flow
    eng.close()
    orc.close()


def test_dense_sums_come_on_and_go_off_with_epsilon(monkeypatch):
    """The exploration rate decides which shape the combined update takes (lob_engine.hip acc_blocked: Q(lambda) below 0.34 sums per
    block by dense slot ids, above it the updates are added to their slots inside the learn / trace kernels), and
    lob_set_epsilon may move it at any step: slots claimed while the dense sums are off have no id, generations verified
    meanwhile carry no record of one, and every record is void when the mode comes back on.  16 384 books, epsilon 0.8 ->
    0.05 -> 0.8 -> 0.05 inside one episode, every phase against the oracle.  (LOB_Q_LANES=1: the lane learner kernels, which add in
    place, at this batch size.)"""
    monkeypatch.setenv("LOB_Q_LANES", "1")
    B = 16384
    p, eng, orc = make(B, abi.ALGO_QLAMBDA, n_events=300, epsilon=0.8)
    eng.reset(); orc.reset()
    step = 0
    for eps, n in ((0.8, 12), (0.05, 14), (0.8, 10), (0.05, 14)):
        eng.set_epsilon(eps)
        ol.load().oracle_set_epsilon(orc.h, C.c_double(eps))
        for i in range(n):
            eng.td_step(1); orc.td_step(1)
            if i < 2 or i >= n - 2 or i % 4 == 0:
                compare_learner_step(eng, orc, "epsilon %g, step %d" % (eps, step), exact=False, rtol=1e-9)
            step += 1
    flow = eng.flow_stats()
    assert flow["dense_sums"] == 28 and flow["added_in_place"] >= 18, flow
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


@pytest.mark.parametrize("algo", [abi.ALGO_QLAMBDA, abi.ALGO_DOUBLE_Q], ids=["qlambda", "double_q"])
def test_hit_lists_longer_than_the_replay_registers(algo):
    """A hit-list record holds up to LOB_HL_MAX = 35 additions; the env kernel replays 23 from registers and the ones beyond in a
    pass of their own (two round trips for the wave that holds such a book, instead of handing the book to the whole-wave
    evaluations of act_book and learn_q_rest_kernel: in a long run 8-11 books per step are that long).  Here a pre-loaded theta with
    2.5 % of its weights written makes such lists the rule: 32 768 books, 14 steps against the oracle, and the statistics say that
    thousands of live books act from lists of 24-35 entries."""
    B = 32768
    p, eng, orc = make(B, algo, n_events=200)
    rng = np.random.default_rng(23)
    th = np.zeros(p.memory_size)
    idx = rng.choice(p.memory_size, size=500000, replace=False)
    th[idx] = rng.normal(0.0, 0.01, size=idx.size)
    eng.reset(); orc.reset()
    eng.set_theta(th)
    orc.theta()[:] = th
    if algo == abi.ALGO_DOUBLE_Q:
        thb = np.zeros(p.memory_size)
        thb[idx[::2]] = rng.normal(0.0, 0.01, size=idx[::2].size)
        eng.set_theta(thb, 1)
        orc.theta_b()[:] = thb
    light0 = light_books(eng)
    for step in range(14):
        eng.td_step(1); orc.td_step(1)
        if step < 3 or step % 3 == 1:
            compare_learner_step(eng, orc, "long lists, step %d" % step, exact=False, rtol=1e-9)
    st = eng.fastpath_stats()
    long_lists = int(st["hist"][24:36].sum())
    print("long lists: %d of %d live books hold 24-35 entries (mean %s, max %s), %d without a list" %
          (long_lists, st["live_books"], st["list_len_mean"], st["list_len_max"], st["books_without_list"]))
    assert long_lists > 2000 and st["list_len_max"] > 28
    assert light_books(eng) - light0 > 10 * long_lists
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    if algo == abi.ALGO_DOUBLE_Q:
        np.testing.assert_allclose(eng.theta(1), orc.theta_b(), rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


def test_preloaded_theta_with_a_million_written_weights():
    """The state a long training run is in, at scale: 32 768 books acting from a weight vector with 1.2 M non-zero entries
    (6 % of the table: a hit list would need ~50 entries, so most books take the in-kernel full evaluation and the
    wave-per-book learn kernel; the folded written-weights maps are half full).  20 steps against the oracle."""
    B = 32768
    p, eng, orc = make(B, abi.ALGO_QLAMBDA, n_events=200)
    rng = np.random.default_rng(11)
    th = np.zeros(p.memory_size)
    idx = rng.choice(p.memory_size, size=1200000, replace=False)
    th[idx] = rng.normal(0.0, 0.01, size=idx.size)
    eng.reset(); orc.reset()
    eng.set_theta(th)
    orc.theta()[:] = th
    for step in range(20):
        eng.td_step(1); orc.td_step(1)
        if step < 3 or step % 4 == 3:
            compare_learner_step(eng, orc, "pre-loaded theta, step %d" % step, exact=False, rtol=1e-9)
    st = eng.fastpath_stats()
    assert st["written_weights"] >= 1200000
    ps, flow = eng.path_stats(), eng.flow_stats()
    print("pre-loaded theta: %d written weights, %d of %d live books without a hit list, mean list %s; books acted on: %d from their "
          "hit list, %d in full inside the env kernel, %d handed back by the learn kernel; steps: %s" %
          (st["written_weights"], st["books_without_list"], st["live_books"], st["list_len_mean"], ps[1], ps[6], ps[7], flow))
    # which paths the 20 steps compared above went through: most books have no usable list (a list would need ~50 entries) --
    # they are acted on in full and handed back by the lane learn kernel; the rest replay their lists.  The hand-back count
    # reaches the host LOB_HINT_LAG = 16 steps late (lob_engine.hip): steps 2-16 serve the list-less books inside the env
    # kernel, the last steps go through the work list to the wave-per-book act kernel.
    assert ps[7] > 0 and ps[6] > 0 and ps[1] > 0
    assert flow["act_inline_general"] >= 10, flow
    if st["books_without_list"] > B // 8:      # (clearly above the B / 16 at which the host switches paths)
        assert flow["act_work_list_dense"] >= 2 and flow["every_book"] >= 2, flow
    elif st["books_without_list"] < B // 32:
        assert flow["act_work_list_dense"] == 0, flow
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


@pytest.mark.parametrize("forced", [False, True], ids=["by_hint", "forced"])
def test_dense_theta_from_random_init_takes_the_work_list_act_path(monkeypatch, forced):
    """learning.random_init: true (src/rl/agent.cpp:37-39) at scale -- EVERY weight 2u - 1 from the agent's mt19937_64, so every
    tile lies on a written weight, no hit list fits a record and every book takes two full evaluations per step.  32 768 books,
    M = 20 M, against the oracle, and the flow counters ASSERT which act path the compared steps took: `by_hint` -- the first
    steps in full inside the env kernel (act_book), then, once the learn kernels' hand-back count of 16 steps ago says "most
    books" (lob_engine.hip LOB_HINT_LAG), through the work list to the wave-per-book act kernel + env_kernel<64, 2, 2> and the
    accumulate pass over every book; `forced` (LOB_MOSTLY_GENERAL=1) -- that path from the second step on."""
    if forced:
        monkeypatch.setenv("LOB_MOSTLY_GENERAL", "1")
    B, n_steps = 32768, 12 if forced else 22
    p, eng, orc = make(B, abi.ALGO_QLAMBDA, n_events=200, random_init=1)
    th0 = eng.theta()
    np.testing.assert_array_equal(th0, orc.theta())          # the same 20 M draws, bit for bit
    assert np.count_nonzero(th0) >= p.memory_size - 4 and th0.min() >= -1.0 and th0.max() < 1.0
    eng.reset(); orc.reset()
    for step in range(n_steps):
        eng.td_step(1); orc.td_step(1)
        if step < 2 or step % 5 == 4 or step >= n_steps - 3:
            compare_learner_step(eng, orc, "dense theta (%s), step %d" % ("forced" if forced else "by hint", step), exact=False, rtol=1e-9)
    flow, ps = eng.flow_stats(), eng.path_stats()
    print("dense theta:", flow, "hit-list books %d, in-kernel full evaluations %d, handed back %d" % (ps[1], ps[6], ps[7]))
    assert ps[1] == 0 and ps[7] >= (n_steps - 1) * B          # no book ever replays a list; the learn kernel hands every book back
    if forced:
        assert flow["act_work_list_dense"] == n_steps - 1 and flow["act_inline_general"] == 0, flow
    else:
        assert flow["act_inline_general"] >= 14 and flow["act_work_list_dense"] >= 4, flow
        assert flow["every_book"] >= 4, flow
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


def test_config3_headline_size_seventy_steps():
    """BASELINE config 3 at its FULL size -- 65 536 books, D = 10, Q(lambda), one 20 M-weight table -- for 70 steps (round 4
    followed it for 12 from a reset and left the long comparison to 32 768 books): the steady state of the timed kernels
    (long hit lists, several live generations, a combine table that persists) at the batch size bench.py quotes."""
    B = 65536
    p, eng, orc = make(B, abi.ALGO_QLAMBDA, n_events=330)
    eng.reset(); orc.reset()
    light0 = light_books(eng)
    for step in range(70):
        eng.td_step(1); orc.td_step(1)
        if step < 2 or step % 8 == 7 or step >= 67:
            compare_learner_step(eng, orc, "C3 x 70, step %d" % step, exact=False, rtol=1e-9)
    assert light_books(eng) - light0 > 60 * B * 0.9
    flow = eng.flow_stats()
    assert flow["added_in_place"] >= 60, flow
    th, oth = eng.theta(), orc.theta()
    assert np.array_equal(th != 0, oth != 0) and np.count_nonzero(th) > 60000
    np.testing.assert_allclose(th, oth, rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


def episode_length(p, rec, B):
    """Learner steps until no book of the batch is live: a function of the streams alone (a step ends when the midprice has moved,
    an episode when the market closes or the stream runs dry -- Base::performAction, base.cpp:285-305; nothing the agent does moves
    either), so an engine pass with alpha = 0 tells how long the compared run below will be."""
    import copy
    q = copy.copy(p)
    q.alpha = 0.0
    probe = engine.Engine(q, B)
    probe.load_events(rec)
    probe.reset()
    n = 0
    live = []
    while True:
        probe.td_step(1)
        n += 1
        live.append(int(probe.counters()[2]))
        if live[-1] == 0:
            break
        assert n < 100000
    probe.close()
    return n, live


def test_config3_headline_size_to_exhaustion_and_a_second_episode():
    """The regime the sustained figure is credited in (VERDICT r5, weak #1a): BASELINE config 3 at its full 65 536 books run until
    the LAST book has stopped -- through the tail in which the live fraction falls to zero and the lane kernels work on a
    shrinking share of their books -- then Runner::RunEpisode's epilogue (ClearInventory, serial.cpp:31), Agent::HandleTerminal,
    Initialise and 20 steps of a second episode.  Short streams (330 events: ~160 learner steps) so that the threaded oracle
    follows; compared every 8th step, at EVERY step of the last 20 before the final book stops, and through episode 2; the
    flow counters assert that the timed kernels (pair learn kernel + lane trace kernel, updates added in place) served the
    tail as they serve the full batch."""
    B = 65536
    p, eng, orc = make(B, abi.ALGO_QLAMBDA, n_events=330)
    T, live = episode_length(p, orc.records, B)
    print("episode of %d learner steps; live books at steps T-20, T-10, T-2: %d, %d, %d" % (T, live[T - 21], live[T - 11], live[T - 3]))
    assert T > 100 and live[T - 21] < B and live[T - 6] < B // 2     # a real tail: the last steps run on a minority of the books
    eng.reset(); orc.reset()
    light0 = light_books(eng)
    for step in range(T):
        eng.td_step(1); orc.td_step(1)
        if step < 2 or step % 8 == 7 or step >= T - 20 or (live[step] < B // 2 and step % 3 == 0):
            compare_learner_step(eng, orc, "C3 to exhaustion, step %d of %d (%d live)" % (step, T, live[step]), exact=False, rtol=1e-9)
    assert eng.counters()[2] == 0 and orc.counters()[2] == 0
    flow, ps = eng.flow_stats(), eng.path_stats()
    # every step but the first (act_fast_kernel: no hit lists yet) took the fused env kernel and added its updates in place; no
    # step fell back to the pass over every book, and the hit lists served (almost) every live book's action
    assert flow["added_in_place"] >= T - 1 and flow["every_book"] == 0 and flow["act_work_list_dense"] == 0, flow
    assert light_books(eng) - light0 > 0.9 * (sum(live[:-1])), (light_books(eng) - light0, sum(live))
    th, oth = eng.theta(), orc.theta()
    assert np.array_equal(th != 0, oth != 0) and np.count_nonzero(th) > 60000
    np.testing.assert_allclose(th, oth, rtol=1e-9, atol=1e-12)
    eng.clear_inventory(); orc.clear_inventory()
    compare_learner_step(eng, orc, "C3 to exhaustion: after ClearInventory", exact=False, rtol=1e-9)
    eng.handle_terminal(); orc.handle_terminal()
    eng.reset(); orc.reset()
    for step in range(20):
        eng.td_step(1); orc.td_step(1)
        if step < 3 or step % 4 == 3:
            compare_learner_step(eng, orc, "C3 episode 2, step %d" % step, exact=False, rtol=1e-9)
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    print("path stats:", ps, "flow:", flow)
    eng.close()
    orc.close()


def test_config2_seventy_steps_and_a_second_episode_on_its_own_dispatch():
    """BASELINE config 2 -- 4 096 books, D = 10, SARSA(lambda), one 20 M-weight table -- on the kernels ITS batch size selects (no
    switch set: learn_q_fast_kernel, the wave-per-book trace kernels, accumulate over 64 waves; VERDICT r5 weak #1b): 70 steps
    against the oracle at every step -- long hit lists, 25 live generations per book, a combine table that persists -- then the
    end-of-episode calls and 20 steps of a second episode."""
    B = 4096
    p, eng, orc = make(B, abi.ALGO_SARSA, n_events=330)
    for episode, n_steps in ((0, 70), (1, 20)):
        eng.reset(); orc.reset()
        for step in range(n_steps):
            eng.td_step(1); orc.td_step(1)
            compare_learner_step(eng, orc, "C2 episode %d step %d" % (episode, step), exact=False, rtol=1e-9)
        eng.clear_inventory(); orc.clear_inventory()
        eng.handle_terminal(); orc.handle_terminal()
    th, oth = eng.theta(), orc.theta()
    assert np.array_equal(th != 0, oth != 0) and np.count_nonzero(th) > 20000
    np.testing.assert_allclose(th, oth, rtol=1e-9, atol=1e-12)
    st = eng.fastpath_stats()
    print("C2 x 70 + 20:", fp_stats_line(st), eng.flow_stats())
    eng.close()
    orc.close()


def fp_stats_line(st):
    return {k: v for k, v in st.items() if k != "hist"}


@pytest.mark.parametrize("algo", [abi.ALGO_QLAMBDA, abi.ALGO_DOUBLE_Q], ids=["qlambda", "double_q"])
def test_model_log_rows_in_the_merged_rest_flow(algo, monkeypatch):
    """in the fused Q(lambda) / double Q flow a book the learn kernel hands back gets its TD error from
    trace_rest_kernel -- AFTER the point where td_stats_kernel used to run, so its row summed |Q(s, a)| instead of |delta|.  16 384
    books on the lane kernels (LOB_Q_LANES=1), a pre-loaded theta with 6 % of its weights written so that most books ARE handed
    back (asserted): every model_log row against the oracle's."""
    monkeypatch.setenv("LOB_Q_LANES", "1")
    B = 16384
    p, eng, orc = make(B, algo, n_events=200)
    rng = np.random.default_rng(31)
    th = np.zeros(p.memory_size)
    idx = rng.choice(p.memory_size, size=1200000, replace=False)
    th[idx] = rng.normal(0.0, 0.01, size=idx.size)
    eng.model_log_enable()
    eng.reset(); orc.reset()
    eng.set_theta(th)
    orc.theta()[:] = th
    if algo == abi.ALGO_DOUBLE_Q:
        eng.set_theta(th[::-1].copy(), 1)
        orc.theta_b()[:] = th[::-1]
    ps0 = eng.path_stats()
    n_steps = 12
    for step in range(n_steps):
        eng.td_step(1); orc.td_step(1)
    compare_learner_step(eng, orc, "model_log merged flow, step %d" % n_steps, exact=False, rtol=1e-9)
    ps1, flow = eng.path_stats(), eng.flow_stats()
    assert ps1[7] - ps0[7] > n_steps * B // 4, (ps0, ps1)      # books handed back by the learn kernel: a large share of every step
    assert flow["added_in_place"] >= n_steps - 1, flow         # ... in the flow that finishes them in trace_rest_kernel
    rows, lost = eng.model_log_read()
    want = orc.model_log()
    assert lost == 0 and len(rows) == len(want) == n_steps
    np.testing.assert_allclose(rows, want, rtol=1e-9)
    eng.close()
    orc.close()


def test_sarsa_learn_kernels_in_front_of_its_trace_kernels_must_fail(monkeypatch):
    """The order SARSA(lambda)'s kernels may NOT have (rl_markets_amd/csrc/lob_state.h, above `hl_rec`): its trace step marks the
    new generation's 32 tiles in the written-weights maps, and the learn kernel builds the next step's hit list by those maps --
    a list built BEFORE the mark lacks every tile of s' that hashes onto a weight this very step's update is about to write, and
    the next action selection then drops that addition.  Round 5 ran the trace kernels BESIDE the learn kernels, saw TD errors
    off by 2e-4 in three books of 16 384 one step into the second episode, and did not find the reader.  A -DLOB_EXPERIMENTS
    build keeps the forbidden order behind LOB_SARSA_LEARN_FIRST=1 (the learn kernels first, Q(s, a) taken from qs_last):
    this test runs it against the oracle and REQUIRES a mismatch -- if it ever passes, the dependency documented there is gone
    or the comparison has gone blind."""
    from tests.test_gpu_parity import experiments_build
    if not experiments_build():
        pytest.skip("the forbidden order exists in -DLOB_EXPERIMENTS builds only (tools/exp_variants.sh)")
    monkeypatch.setenv("LOB_SARSA_LEARN_FIRST", "1")
    B = 16384
    p, eng, orc = make(B, abi.ALGO_SARSA, n_events=260)
    with pytest.raises(AssertionError):
        for episode, n_steps in ((0, 40), (1, 12)):
            eng.reset(); orc.reset()
            for step in range(n_steps):
                eng.td_step(1); orc.td_step(1)
                compare_learner_step(eng, orc, "learn first, episode %d step %d" % (episode, step), exact=False, rtol=1e-9)
            eng.clear_inventory(); orc.clear_inventory()
            eng.handle_terminal(); orc.handle_terminal()
    eng.close()
    orc.close()


def test_reset_then_weight_load_then_steps(monkeypatch):
    """lob_reset -> lob_theta_set -> lob_td_step: the weight load re-evaluates the memo records of the slots on the
    current list, which after a reset must be EMPTY -- slots of the episode before would be re-stamped as holding
    valid tiles for triples whose hashes the reset has wiped.  Engine against the oracle through that
    sequence, small table so that slots are reused by different triples."""
    B = 300
    p, eng, orc = make(B, abi.ALGO_QLAMBDA, n_events=260, mem=1 << 12, depth=5, epsilon=0.5)
    eng.reset(); orc.reset()
    for step in range(30):
        eng.td_step(1); orc.td_step(1)
    compare_learner_step(eng, orc, "episode 1", exact=False, rtol=1e-9)
    eng.clear_inventory(); orc.clear_inventory()
    eng.handle_terminal(); orc.handle_terminal()
    eng.reset(); orc.reset()
    th = orc.theta().copy()
    rng = np.random.default_rng(5)
    th[rng.integers(0, th.size, size=200)] += 1e-3
    eng.set_theta(th)
    orc.theta()[:] = th
    for step in range(30):
        eng.td_step(1); orc.td_step(1)
        compare_learner_step(eng, orc, "after reset + load, step %d" % step, exact=False, rtol=1e-9)
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


# ---- the randomised configuration sweep with the timed kernels forced on ------------------------------------
FLOWS = []       # (variant, algorithm, Engine.flow_stats()) of every case: which update path it has compared with the oracle
CUT_SHORT = []   # cases the shadow oracle ended early (a chaotic configuration proves nothing beyond that step): reported below


VARIANTS = {
    "dq_lane": {"LOB_Q_LANES": "1", "LOB_FUSE_ACT": "1", "LOB_DQ_PAIR": "0"},   # DoubleQLearn through env_step_kernel<., true> / learn_q_lane_kernel<LOB_ALGO_DOUBLE_Q>
    "dq_pair": {"LOB_Q_LANES": "1", "LOB_FUSE_ACT": "1"},                       # ... / learn_q_pair_kernel<LOB_ALGO_DOUBLE_Q> (two lanes per book: the default)
    "lane": {"LOB_Q_LANES": "1", "LOB_Q_PAIR": "0", "LOB_FUSE_ACT": "1"},
    "pair": {"LOB_Q_LANES": "1", "LOB_Q_PAIR": "1", "LOB_FUSE_ACT": "1"},
    # (batches of up to 4 096 books take env_step16_kernel -- a book's levels across 16 lanes -- by themselves, so every other variant
    # of this sweep runs it; here the lane-per-book env_step_kernel that the large batches use)
    "pair_env64": {"LOB_Q_LANES": "1", "LOB_Q_PAIR": "1", "LOB_FUSE_ACT": "1", "LOB_ENV16_MAX": "0"},
    "pair_nofuse": {"LOB_Q_LANES": "1", "LOB_Q_PAIR": "1", "LOB_NO_FUSE": "1", "LOB_FUSE_ACT": "1"},
    # ("pair": what the lane trace kernel hands on and what the fused accumulation left are served by ONE launch, trace_rest_kernel;
    # here by trace_fast_kernel<., 2> + accumulate_kernel over its list, the two launches it replaced)
    "pair_rest_split": {"LOB_Q_LANES": "1", "LOB_Q_PAIR": "1", "LOB_FUSE_ACT": "1", "LOB_REST_MERGE": "0"},
    # (Q(lambda): "pair" adds the updates to their slots inside the learn / trace kernels; here accumulate_kernel does, over every book)
    "pair_acc_pass": {"LOB_Q_LANES": "1", "LOB_Q_PAIR": "1", "LOB_FUSE_ACT": "1", "LOB_ACC_FUSE": "0"},
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("seed", range(int(os.environ.get("LOB_FUZZ_SEEDS", "32")) // 2))
def test_random_configuration_timed_kernels(monkeypatch, seed, variant):
    """tests/test_gpu_fuzz.py's random configurations (depth, trade slots, state-variable sets, rewards, bounds,
    look-backs, table sizes, learning constants, stream statistics), restricted to what the fast path serves
    (shared theta, SARSA(lambda) / Q(lambda)) and with the kernels of the timed run forced on for batches that
    would not select them by size: the lane-per-book learner kernels (one or two lanes per book, with and without
    the fused trace step) and the action selection inside env_kernel.  Two episodes of 70 steps, every step
    against the oracle."""
    for k, v in VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    p, g, _ = random_case(9000 + seed)
    r = np.random.default_rng(77 + seed)
    p.theta_mode = abi.THETA_SHARED
    p.algo = int(r.choice([abi.ALGO_SARSA, abi.ALGO_QLAMBDA]))
    B = int(r.choice([3, 64, 130, 300]))
    if variant.startswith("dq_"):
        p.algo = abi.ALGO_DOUBLE_Q
        p.max_trades = min(p.max_trades, 2)   # (its fused env kernel is the two-trade-slot one; more slots: the general kernels)
        g.trade2_prob_q16 = g.trade2_prob_q16 if p.max_trades > 1 else 0
    rec = engine.gen_stream_host(g, p.depth, p.max_trades, p.book_id_offset, B)
    eng = engine.Engine(p, B)
    eng.load_events(rec)
    orc = ol.Oracle(p, rec)
    # How far rounding noise travels in THIS configuration: a shadow oracle whose weights are nudged by 1e-13 (relative) after
    # step 3.  Most configurations keep the two oracles within 1e-11 of each other; some (alpha = 0.3 with 300 greedy books on
    # one small table: seed 27 of the wider sweep) blow the nudge up to 1e-9 by step 18 and 1e-4 by step 30 -- there the
    # order of the engine's atomic additions shows just as much, on every path (LOB_NO_COMBINE=1, the trace-by-trace update of
    # round 1, included), and the comparison allows 100 x what the shadow has drifted; once the shadow takes a different
    # action the configuration says nothing any more.
    shadow = ol.Oracle(p, rec)
    noise, chaotic = 0.0, False
    for episode in range(2):
        eng.reset()
        orc.reset()
        shadow.reset()
        for step in range(70):
            eng.td_step(1)
            orc.td_step(1)
            shadow.td_step(1)
            if episode == 0 and step == 3:
                th = shadow.theta()
                th[th != 0] *= 1.0 + 1e-13
            a, b = orc.recs(), shadow.recs()
            if not np.array_equal(a["action"], b["action"]) or not np.array_equal(a["rng_ctr"], b["rng_ctr"]):
                chaotic = True
                CUT_SHORT.append("%s seed %d at episode %d step %d" % (variant, seed, episode, step))
                break
            noise = max(noise, float(np.abs(a["td"] - b["td"]).max()))
            try:
                compare_learner_step(eng, orc, "%s seed %d episode %d step %d" % (variant, seed, episode, step), exact=False, rtol=1e-9,
                                     td_floor=100.0 * noise)
            except AssertionError:
                # An ill-conditioned configuration (greedy ties under alpha = 0.3: seed 29 of the wider sweep, LOB_FUZZ_SEEDS=96 -- one run
                # in three, with four test processes sharing the GPU and so other orders of the f64 atomics) can send the ENGINE
                # across a tie a step before the shadow oracle crosses it.  Only where the shadow has already blown its 1e-13 nudge up
                # a thousandfold is that taken for what it is and the case cut short (counted below); anywhere else it is a failure.
                if noise < 1e-10:
                    raise
                chaotic = True
                CUT_SHORT.append("%s seed %d at episode %d step %d (the engine left first; shadow drift %.1e)" % (variant, seed, episode, step, noise))
                break
        if chaotic:
            break
        eng.clear_inventory(); orc.clear_inventory(); shadow.clear_inventory()
        eng.handle_terminal(); orc.handle_terminal(); shadow.handle_terminal()
    # which update path the comparison above has been through
    flow = eng.flow_stats()
    FLOWS.append((variant, p.algo, flow))
    if variant == "pair_acc_pass":
        assert flow["added_in_place"] == 0, flow
    if not chaotic:
        drift = float(np.abs(orc.theta() - shadow.theta()).max())
        np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=max(1e-12, 100.0 * drift))
        if p.algo == abi.ALGO_DOUBLE_Q:
            drift_b = float(np.abs(orc.theta_b() - shadow.theta_b()).max())
            np.testing.assert_allclose(eng.theta(1), orc.theta_b(), rtol=1e-9, atol=max(1e-12, 100.0 * drift_b))
    eng.close()
    orc.close()
    shadow.close()


def test_zz_report_cases_cut_short():
    """(runs after the sweep: file order)  How many of the sweep's cases stopped comparing early, and where -- printed with -s /
    -rA; a sweep in which most cases end early would be a sweep that checks little."""
    n = len(VARIANTS) * (int(os.environ.get("LOB_FUZZ_SEEDS", "32")) // 2)
    line = "timed-kernel sweep: %d of %d cases cut short by the shadow oracle%s" % (len(CUT_SHORT), n, (": " + "; ".join(CUT_SHORT)) if CUT_SHORT else "")
    print(line)
    # (kept where the GPU call's merged output lands, so that the count is on record beside the test log and not only under -s)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "sweep_cut_short.txt"), "w") as f:
            f.write(line + "\n")
    except OSError:
        pass
    assert len(CUT_SHORT) <= n // 10, line    # 8 of the default 80: a sweep in which more cases end early checks too little


def test_zz_update_paths_the_sweep_went_through():
    """(after the sweep)  The Q(lambda) cases of the "pair" variant must have had their updates added to the slots inside the
    learn / trace kernels (Engine.flow_stats), those of "pair_acc_pass" by accumulate_kernel over every book: both shapes of
    the combined update have then been compared with the oracle step by step."""
    if not FLOWS:
        pytest.skip("the sweep did not run in this session")
    ql = [(v, f) for v, a, f in FLOWS if a == abi.ALGO_QLAMBDA]
    in_place = [f for v, f in ql if v in ("pair", "pair_rest_split", "pair_env64") and f["added_in_place"] > 0 and f["every_book"] == 0]
    whole = [f for v, f in ql if v == "pair_acc_pass" and f["every_book"] > 0]
    print("Q(lambda) cases with updates added in place: %d, with the accumulate pass over every book: %d" % (len(in_place), len(whole)))
    if os.environ.get("PYTEST_XDIST_WORKER"):
        pytest.skip("under pytest-xdist this process has seen only its share of the sweep")
    assert len(in_place) >= 6 and len(whole) >= 3
    assert all(f["added_in_place"] == 0 for v, f in ql if v not in ("pair", "pair_rest_split", "pair_env64", "lane"))
