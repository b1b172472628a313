"""BASELINE config 5: every book replays ONE recorded LOBSTER-format stream from its own phase
(lob_load_events_shared), with the risk-averse rewards.  Parity against the oracle at a size it
finishes in seconds (the oracle gets each book's window as a private copy), then at the full
65 536 books through properties that do not depend on the size."""
import numpy as np
import pytest

from rl_markets_amd import abi, engine
from tests import oracle_lib as ol
from tests.csv_io import write_lobster
from tests.parity import compare_learner_step, dumps_to_np

pytestmark = pytest.mark.gpu


def lobster_day(tmp_path, n_total, depth=10, trades=2):
    """Synthetic stream -> LOBSTER message/orderbook files -> lob_convert_lobster -> records."""
    g = engine.default_gen_params()
    g.n_events = n_total
    rec = engine.gen_stream_host(g, depth, trades, 0, 1)
    ob, msg = str(tmp_path / "ob.csv"), str(tmp_path / "msg.csv")
    write_lobster(rec[0], depth, trades, depth, ob, msg)
    day = engine.convert_lobster(ob, msg, depth, depth, trades)[0]
    assert day.shape[0] == n_total
    return day


def replay_params(reward, algo=abi.ALGO_QLAMBDA, mem=1 << 20):
    p = engine.default_params()
    p.depth, p.max_trades = 10, 2
    p.algo, p.theta_mode, p.memory_size = algo, abi.THETA_SHARED, mem
    p.reward_measure = reward
    if reward == abi.REWARD_MM_LINEAR:
        p.pos_weight = 0.5  # the inventory-penalised variant
    return p


@pytest.mark.parametrize("reward", [abi.REWARD_PNL_DAMPED, abi.REWARD_MM_LINEAR])
def test_replayed_stream_matches_oracle(tmp_path, reward):
    B, n_total, n_events = 24, 2500, 500
    day = lobster_day(tmp_path, n_total)
    rng = np.random.default_rng(3)
    phase = rng.integers(0, n_total - n_events + 1, size=B)
    phase[0], phase[1], phase[2] = 0, n_total - n_events, phase[3]  # both ends, and two books in step
    p = replay_params(reward)
    eng = engine.Engine(p, B)
    eng.load_events_shared(day, phase, n_events)
    windows = np.stack([day[s:s + n_events] for s in phase])
    orc = ol.Oracle(p, windows)
    eng.reset()
    orc.reset()
    for step in range(150):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "replay step %d" % step, exact=False, rtol=1e-9)
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    # a book with its own copy of the window is the same book
    eng2 = engine.Engine(p, B)
    eng2.load_events(windows)
    eng2.reset()
    eng2.td_step(150)
    d1, d2 = dumps_to_np(eng.get_books()), dumps_to_np(eng2.get_books())
    for name in d1.dtype.names:
        assert np.array_equal(d1[name], d2[name]), name


def test_shared_stream_arguments():
    p = replay_params(abi.REWARD_PNL_DAMPED)
    g = engine.default_gen_params()
    g.n_events = 300
    day = engine.gen_stream_host(g, 10, 2, 0, 1)[0]
    eng = engine.Engine(p, 4)
    with pytest.raises(engine.LobError):
        eng.load_events_shared(day, [0, 0, 0, 101], 200)   # runs past the stream
    with pytest.raises(engine.LobError):
        eng.load_events_shared(day, [0, -1, 0, 0], 200)
    eng.load_events_shared(day, [0, 50, 100, 100], 200)
    eng.reset()
    eng.td_step(5)
    eng.load_events(np.stack([day[:200]] * 4))                # back to per-book streams
    eng.reset()
    eng.td_step(5)


def test_config5_full_size(tmp_path):
    """65 536 books on one replayed day.  With alpha = 0 the books do not interact (theta stays 0,
    actions come from each book's own RNG stream), so any book of the big run must equal a
    one-book engine given the same global book id and a private copy of its window."""
    B, n_total, n_events, steps = 65536, 6000, 400, 60
    day = lobster_day(tmp_path, n_total)
    rng = np.random.default_rng(5)
    phase = rng.integers(0, n_total - n_events + 1, size=B)
    p = replay_params(abi.REWARD_PNL_DAMPED, mem=20000000)
    p.alpha = 0.0
    eng = engine.Engine(p, B)
    eng.load_events_shared(day, phase, n_events)
    eng.reset()
    eng.td_step(steps)
    cnt = eng.counters()
    assert cnt[0] == steps * B
    assert not eng.theta().any()
    sample = [0, 1, 4097, 32768, 65535]
    big = dumps_to_np(eng.get_books())
    acts = eng.last_actions()
    rew = eng.last_rewards()
    for b in sample:
        p1 = replay_params(abi.REWARD_PNL_DAMPED, mem=20000000)
        p1.alpha = 0.0
        p1.book_id_offset = b
        one = engine.Engine(p1, 1)
        one.load_events(day[phase[b]:phase[b] + n_events][None])
        one.reset()
        one.td_step(steps)
        d = dumps_to_np(one.get_books())
        for name in d.dtype.names:
            assert np.array_equal(d[name][0], big[name][b]), (b, name)
        assert one.last_actions()[0] == acts[b] and one.last_rewards()[0] == rew[b]
        one.close()
    # and with learning on: the run completes, every book steps, weights get written
    eng.set_alpha(0.001)
    eng.td_step(steps)
    assert eng.counters()[0] == 2 * steps * B
    th = eng.theta()
    assert np.isfinite(th).all() and np.count_nonzero(th) > 1000


def test_config5_full_size_against_oracle(tmp_path):
    """BASELINE config 5 at its full size WITH learning on: 65 536 books replaying one LOBSTER-format
    day from their own phases, risk-averse reward pnl_damped, Q(lambda), one shared 20M-weight table,
    followed step by step by the oracle (which gets every book's window as a private copy)."""
    B, n_total, n_events, steps = 65536, 4000, 150, 10
    day = lobster_day(tmp_path, n_total)
    rng = np.random.default_rng(11)
    phase = rng.integers(0, n_total - n_events + 1, size=B)
    p = replay_params(abi.REWARD_PNL_DAMPED, mem=20000000)
    eng = engine.Engine(p, B)
    eng.load_events_shared(day, phase, n_events)
    windows = day[phase[:, None] + np.arange(n_events)[None, :]]
    orc = ol.Oracle(p, windows)
    eng.reset()
    orc.reset()
    for step in range(steps):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "C5 full size step %d" % step, exact=False, rtol=1e-9)
    th, oth = eng.theta(), orc.theta()
    assert np.array_equal(th != 0, oth != 0) and np.count_nonzero(th) > 10000
    np.testing.assert_allclose(th, oth, rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


def test_config5_learning_across_ring_refills_against_oracle(tmp_path, monkeypatch):
    """Config 5 with learning ON across refills of the market-track ring (VERDICT r5 weak #1c: the full-day test compares the ring
    with a resident track at alpha = 0 -- the engine with itself).  4 096 books replay windows of 1 500 events of one
    LOBSTER-format day through a 256-entry ring refilled every 16 steps (prepass_extend_kernel runs a dozen times), risk-averse
    reward, Q(lambda) on one shared 20 M-weight table, alpha = 0.001: every 5th step -- and every step around the first refills
    -- against the oracle, 220 steps deep."""
    monkeypatch.setenv("LOB_TRACK_RING", "256")
    monkeypatch.setenv("LOB_TRACK_REFILL", "16")
    B, n_total, n_events, steps = 4096, 6000, 1500, 220
    day = lobster_day(tmp_path, n_total)
    rng = np.random.default_rng(17)
    phase = rng.integers(0, n_total - n_events + 1, size=B)
    p = replay_params(abi.REWARD_PNL_DAMPED, mem=20000000)
    eng = engine.Engine(p, B)
    eng.load_events_shared(day, phase, n_events)
    windows = day[phase[:, None] + np.arange(n_events)[None, :]]
    orc = ol.Oracle(p, windows)
    eng.kernel_timing(True)
    eng.reset()
    orc.reset()
    for step in range(steps):
        eng.td_step(1)
        orc.td_step(1)
        if step % 5 == 4 or 14 <= step <= 18 or 30 <= step <= 34:
            compare_learner_step(eng, orc, "C5 ring + learning, step %d" % step, exact=False, rtol=1e-9)
    eng.sync()
    _, refills = eng.kernel_time_ms("prepass_extend_kernel")
    assert refills >= 10, refills                                  # the ring really was refilled while the books learnt
    th, oth = eng.theta(), orc.theta()
    assert np.array_equal(th != 0, oth != 0) and np.count_nonzero(th) > 10000
    np.testing.assert_allclose(th, oth, rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


def test_config5_full_day_at_full_size(tmp_path, monkeypatch):
    """BASELINE config 5 with a whole recorded day per book: 65 536 books x 50 000 events each of one
    61 200-event LOBSTER-format day.  A resident market track would take 65 536 x 50 000 x 96 B = 315 GB;
    the 4 096-entry ring takes 26 GB.  With alpha = 0 the books do not interact, so sampled books of the
    big run must equal one-book engines given the same global book id and window -- whose tracks are
    made RESIDENT (LOB_TRACK_RING above the stream length): ring against resident, 600 steps deep
    (several ring refills)."""
    B, n_total, n_events, steps = 65536, 61200, 50000, 600
    day = lobster_day(tmp_path, n_total)
    rng = np.random.default_rng(9)
    phase = rng.integers(0, n_total - n_events + 1, size=B)
    p = replay_params(abi.REWARD_PNL_DAMPED, mem=20000000)
    p.alpha = 0.0
    eng = engine.Engine(p, B)
    eng.load_events_shared(day, phase, n_events)
    eng.reset()
    eng.td_step(steps)
    cnt = eng.counters()
    assert cnt[0] == steps * B and cnt[1] > 4096 * B / 4     # (events consumed incl. the skip to market open)
    big = dumps_to_np(eng.get_books())
    acts, rew = eng.last_actions(), eng.last_rewards()
    monkeypatch.setenv("LOB_TRACK_RING", str(1 << 16))
    for b in [0, 4097, 65535]:
        p1 = replay_params(abi.REWARD_PNL_DAMPED, mem=20000000)
        p1.alpha = 0.0
        p1.book_id_offset = b
        one = engine.Engine(p1, 1)
        one.load_events(day[phase[b]:phase[b] + n_events][None])
        one.reset()
        one.td_step(steps)
        d = dumps_to_np(one.get_books())
        for name in d.dtype.names:
            assert np.array_equal(d[name][0], big[name][b]), (b, name)
        assert one.last_actions()[0] == acts[b] and one.last_rewards()[0] == rew[b]
        one.close()
    # and with learning on
    eng.set_alpha(0.001)
    eng.td_step(100)
    assert eng.counters()[0] == (steps + 100) * B
    th = eng.theta()
    assert np.isfinite(th).all() and np.count_nonzero(th) > 1000
    eng.close()
