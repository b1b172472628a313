"""Randomised configuration sweep: the HIP engine against the oracle over the whole parameter
surface the C ABI accepts (depth, trade slots, state-variable sets and order, rewards, quoting and
target-price modes, bounds, look-backs, weights, learning constants, algorithm, theta mode, table
size, stream statistics).  Each case is fully determined by its seed, printed on failure."""
import os

import numpy as np
import pytest

from rl_markets_amd import abi, engine
from tests import oracle_lib as ol
from tests.parity import compare_learner_step

pytestmark = pytest.mark.gpu

REWARDS = [abi.REWARD_PNL, abi.REWARD_PNL_DAMPED, abi.REWARD_SPREAD, abi.REWARD_NORMED, abi.REWARD_LOVOL,
           abi.REWARD_MM_LINEAR, abi.REWARD_MM_DIV]


def random_case(seed):
    r = np.random.default_rng(seed)
    p = engine.default_params()
    p.depth = int(r.choice([3, 5, 10]))
    p.max_trades = int(r.choice([1, 2, 4]))
    nv = int(r.integers(4, abi.LOB_MAX_VARS + 1))
    order = r.permutation(abi.LOB_MAX_VARS)[:nv]
    p.n_vars = nv
    for i in range(abi.LOB_MAX_VARS):
        p.vars[i] = int(order[i]) if i < nv else 0
    p.order_size = int(r.choice([1, 10, 25, 100]))
    p.reward_measure = int(r.choice(REWARDS))
    bound = int(r.choice([1, 3, 10, 50])) * p.order_size
    p.pos_lb, p.pos_ub = -bound, int(r.choice([bound, 2 * bound]))
    p.damping_factor = float(r.uniform(0.0, 1.0))
    p.pos_weight, p.trd_weight, p.pnl_weight = (float(x) for x in r.uniform(0.0, 2.0, size=3))
    for name in ("lb_mpm", "lb_vlt", "lb_svl", "lb_vwap", "lb_rsi", "lb_spread", "lb_pnl", "lb_target"):
        setattr(p, name, int(r.choice([1, 2, 7, 15, 45, 60, 100])))
    p.target_price = int(r.integers(0, 2))
    p.quote_mode = int(r.integers(0, 2))
    p.memory_size = int(r.choice([4099, 1 << 14, 100003, 1 << 20, 3000017]))
    w = r.uniform(0.05, 1.0, size=3)
    for i in range(3):
        p.group_weights[i] = float(w[i] / w.sum())
    p.gamma = float(r.uniform(0.8, 1.0))
    # gamma*lambda <= 0.93: the trace ring holds at most LOB_TRACE_GENS = 64 generations (lob_create rejects more)
    p.lambda_ = float(r.uniform(0.0, min(0.95, 0.93 / p.gamma)))
    p.alpha = float(r.choice([0.0, 1e-4, 1e-2, 0.3]))
    p.epsilon = float(r.choice([0.0, 0.1, 0.8, 1.0]))
    p.algo = int(r.integers(0, 6))  # SARSA, Q(lambda), double Q, and their average-reward variants (R-learning)
    p.theta_mode = int(r.integers(0, 2))
    p.seed = int(r.integers(0, 1 << 40))
    p.book_id_offset = int(r.choice([0, 7, 1 << 20]))
    g = engine.default_gen_params()
    g.seed = int(r.integers(0, 1 << 40))
    g.n_events = int(r.choice([150, 260, 400]))
    g.move_prob_q16 = int(r.uniform(0.05, 1.0) * 65536)
    g.spread2_prob_q16 = int(r.uniform(0.0, 0.9) * 65536)
    g.trade_prob_q16 = int(r.uniform(0.0, 1.0) * 65536)
    g.trade2_prob_q16 = int(r.uniform(0.0, 1.0) * 65536) if p.max_trades > 1 else 0
    g.touch_prob_q16 = int(r.uniform(0.3, 1.0) * 65536)
    g.vol_min, g.vol_max = 1, int(r.choice([50, 5000]))
    g.trade_min, g.trade_max = 1, int(r.choice([20, 3000]))
    B = int(r.choice([1, 3, 8]))
    return p, g, B


# LOB_FUZZ_SEEDS=n widens the sweep (a few hundred cases take well under a minute)
@pytest.mark.parametrize("seed", range(int(os.environ.get("LOB_FUZZ_SEEDS", "32"))))
def test_random_configuration(seed):
    p, g, B = random_case(1000 + seed)
    rec = engine.gen_stream_host(g, p.depth, p.max_trades, p.book_id_offset, B)
    eng = engine.Engine(p, B)
    eng.load_events(rec)
    orc = ol.Oracle(p, rec)
    exact = p.theta_mode == abi.THETA_PRIVATE or B == 1
    for episode in range(2):
        eng.reset()
        orc.reset()
        for step in range(70):
            eng.td_step(1)
            orc.td_step(1)
            compare_learner_step(eng, orc, "seed %d episode %d step %d" % (seed, episode, step), exact=exact, rtol=1e-9)
        eng.clear_inventory(); orc.clear_inventory()
        eng.handle_terminal(); orc.handle_terminal()
    n = B if p.theta_mode == abi.THETA_PRIVATE else 1
    for which in range(n):
        a, b = eng.theta(which), orc.theta(which)
        if exact:
            np.testing.assert_array_equal(a, b)
        else:
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("LOB_FUZZ_SEEDS", "32")) // 2))
def test_random_configuration_env_only(seed):
    """The environment interface alone (lob_step with caller-chosen actions, then greedy
    evaluation steps) on random configurations, run until every stream is exhausted."""
    from tests.parity import compare_env
    p, g, B = random_case(5000 + seed)
    g.n_events = 260
    rec = engine.gen_stream_host(g, p.depth, p.max_trades, p.book_id_offset, B)
    eng = engine.Engine(p, B)
    eng.load_events(rec)
    orc = ol.Oracle(p, rec)
    eng.reset()
    orc.reset()
    compare_env(eng, orc, "seed %d reset" % seed)
    rng = np.random.default_rng(seed)
    for step in range(400):
        if step % 7 == 6:
            eng.eval_step(1)
            orc.eval_step(1)
        else:
            acts = rng.integers(0, 9, size=B).astype(np.int32)
            eng.step(acts)
            orc.env_step(acts)
        compare_env(eng, orc, "seed %d step %d" % (seed, step))
        live = eng.get_terminal() == 0  # books that have just stepped (the others keep their last reward)
        np.testing.assert_array_equal(eng.get_reward()[live], orc.recs()["reward"][live], err_msg="seed %d step %d reward" % (seed, step))
        if (eng.get_terminal() != 0).all():
            break
    assert (eng.get_terminal() != 0).all(), "streams of 260 events end within 400 steps"
    eng.close()
    orc.close()
