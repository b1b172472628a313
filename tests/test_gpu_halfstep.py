"""The half-step state of the C ABI (include/lob_engine.h lob_td_step_begin / lob_td_step_end): between the two halves only
the weight exchange may run.  Everything that would move the books, the traces or the weights there is refused with
LOB_ESTATE, and lob_reset abandons a half-done step together with its episode -- the case of an exchange that failed between
the halves (ShardedLearner.run, lob::Learner::_step): the next episode must run as if nothing had happened."""
import numpy as np
import pytest

from rl_markets_amd import abi, engine
from tests import oracle_lib as ol
from tests.parity import compare_learner_step

pytestmark = pytest.mark.gpu


def _make(B=96, algo=abi.ALGO_QLAMBDA):
    p = engine.default_params()
    p.memory_size = 1 << 16
    p.theta_mode, p.algo = abi.THETA_SHARED, algo
    g = engine.default_gen_params()
    g.n_events = 300
    rec = engine.gen_stream_host(g, p.depth, p.max_trades, 0, B)
    eng = engine.Engine(p, B)
    eng.load_events(rec)
    return p, rec, eng


def test_mutating_calls_are_refused_between_the_halves():
    p, rec, eng = _make()
    eng.reset()
    eng.td_step(3)
    eng.td_step_begin()
    B = eng.B
    for call in (lambda: eng.td_step(1), lambda: eng.td_step_begin(), lambda: eng.eval_step(1), lambda: eng.step(np.zeros(B, np.int32)),
                 lambda: eng.clear_inventory(), lambda: eng.handle_terminal(), lambda: eng.set_theta(np.zeros(eng.M))):
        with pytest.raises(engine.LobError) as ei:
            call()
        assert ei.value.code == abi.LOB_ESTATE
    eng.td_step_end()          # ... and the step can still be finished
    eng.td_step(1)
    eng.close()


def test_reset_abandons_a_half_done_step():
    """begin, no end (an exchange that threw), reset: the next episode is bit for bit what the oracle runs after the same
    abandoned half step (the books of the abandoned step have performed their action; its learner half never happened)."""
    p, rec, eng = _make()
    orc = ol.Oracle(p, rec)
    eng.reset(); orc.reset()
    for _ in range(5):
        eng.td_step(1); orc.td_step(1)
    eng.td_step_begin(); orc.td_step_begin()
    eng.reset(); orc.reset()                      # no lob_td_step_end in between
    for step in range(12):
        eng.td_step(1)                            # (used to fail with LOB_ESTATE for ever after)
        orc.td_step(1)
        compare_learner_step(eng, orc, "after an abandoned half step, step %d" % step, exact=False, rtol=1e-9)
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


def test_a_stream_of_more_than_two_million_events_is_accepted():
    """The TickStatistics counters are plain 32-bit counters like the reference's ints (lob_state.h tick_ab / tick_pos /
    tick_both; round 4 packed them in 21-bit fields and refused streams of 2^21 events: a 2-6 M-row LOBSTER day):
    a stream of 2^21 + 7 events per book loads (the track is a ring for such a stream), resets and steps, and the counters add up."""
    p, rec, eng = _make(B=2)
    g = engine.default_gen_params()
    g.n_events = (1 << 21) + 7
    eng.gen_events(g)
    eng.reset()
    eng.td_step(40)
    for d in eng.get_books(0, 2):
        assert d.total_ticks == 40
        assert d.ticks_with_position == d.ticks_long + d.ticks_short <= 40
        assert max(d.ticks_with_ask, d.ticks_with_bid) <= 40 and d.ticks_with_both <= min(d.ticks_with_ask, d.ticks_with_bid)
        assert d.ticks_with_ask + d.ticks_with_bid - d.ticks_with_both <= 40
    eng.close()


@pytest.mark.parametrize("pinned", ["1", "0"], ids=["pinned_pieces", "pageable_pieces"])
def test_the_piecewise_upload_paths_hand_over_the_same_stream(monkeypatch, pinned):
    """lob_load_events on the path big streams take: pieces through two pinned staging buffers filled by a
    persistent pool of host threads -- or, when no pinned memory is to be had (LOB_UPLOAD_PINNED=0 forces that branch), the same
    pieces straight from the caller's pageable memory; never a second whole-stream temporary.  Pieces of 1 000 records here
    (LOB_UPLOAD_PIECE_RECS) so that a small stream goes through dozens of them, a ragged last piece included; the books must then
    be the books of the one-copy path, bit for bit, through 30 learner steps."""
    p, rec, eng0 = _make(B=48)
    eng0.reset()
    eng0.td_step(30)
    want = bytes(eng0.get_books())
    th0 = eng0.theta()
    eng0.close()
    monkeypatch.setenv("LOB_UPLOAD_PIECE_RECS", "1000")
    monkeypatch.setenv("LOB_UPLOAD_PINNED", pinned)
    eng = engine.Engine(p, 48)
    eng.load_events(rec)
    eng.reset()
    eng.td_step(30)
    assert bytes(eng.get_books()) == want
    np.testing.assert_allclose(eng.theta(), th0, rtol=1e-9, atol=1e-12)
    eng.close()


def test_the_next_episodes_stream_staged_while_this_one_runs(monkeypatch):
    """lob_stage_events: the reference loads a fresh day before every episode (src/main.cpp:53-55); here the next episode's
    streams are handed over WHILE the current episode steps -- a host thread, a second record buffer, a stream of its own -- and
    the lob_reset that follows adopts them.  Books and weights must be those of an engine that loads the second stream the
    ordinary way between the episodes (pieces of 1 000 records, so that the hand-over is still going on while the steps run)."""
    monkeypatch.setenv("LOB_UPLOAD_PIECE_RECS", "1000")
    p, rec_a, eng = _make(B=48)
    g = engine.default_gen_params()
    g.n_events = 300
    g.seed = 777
    rec_b = engine.gen_stream_host(g, p.depth, p.max_trades, 1000, 48)
    ref = engine.Engine(p, 48)
    ref.load_events(rec_a)
    for e in (eng, ref):
        e.reset()
    eng.stage_events(rec_b)            # ... and the steps of the first episode go on
    for e in (eng, ref):
        e.td_step(40)
        e.clear_inventory()
        e.handle_terminal()
    ref.load_events(rec_b)
    for e in (eng, ref):
        e.reset()                      # (eng: adopts the staged stream)
        e.td_step(40)
    assert bytes(eng.get_books()) == bytes(ref.get_books())
    np.testing.assert_allclose(eng.theta(), ref.theta(), rtol=1e-9, atol=1e-12)
    # a stream of another length, or a second hand-over before the first was adopted, is refused
    eng.stage_events(rec_a)
    with pytest.raises(engine.LobError):
        eng.stage_events(rec_a)
    eng.stage_wait()
    eng.reset()
    with pytest.raises(engine.LobError):
        eng.stage_events(rec_a[:, :200])
    # ... and a stream that fails validation reports it where the hand-over is waited for
    bad = rec_b.copy()
    bad[3, 10, 4] = 0                   # (a zero best-ask price)
    eng.stage_events(bad)
    with pytest.raises(engine.LobError):
        eng.reset()
    eng.close()
    ref.close()
