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


def test_an_episode_longer_than_the_tick_counters_is_refused():
    """The TickStatistics counters are 21-bit fields (lob_state.h tick_ab / tick_pos): a stream of 2^21 events or more is
    refused when it is loaded -- before anything is allocated -- and the stream in place stays usable."""
    p, rec, eng = _make(B=8)
    g = engine.default_gen_params()
    g.n_events = 1 << 21
    with pytest.raises(engine.LobError) as ei:
        eng.gen_events(g)
    assert ei.value.code == abi.LOB_EINVAL and "2^21" in str(ei.value)
    eng.reset()
    eng.td_step(5)             # (the 300-event stream loaded before is still there)
    assert eng.get_books(0, 1)[0].total_ticks == 5
    eng.close()
