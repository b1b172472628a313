"""The HIP engine directly against fixtures produced by the UNMODIFIED
reference (tests/golden/, see make_golden.py): every step of six reference
episodes (actions, rewards, TD errors, state variables, full book / order /
position state, RNG draw counts) and the final weights, bit-exact.  One book,
so this is the reference's own single-book semantics with no batching
caveats."""
import os

import numpy as np
import pytest

from rl_markets_amd import abi, engine
from tests.golden.make_golden import TRAJ_CASES, gen_for
from tests.parity import dumps_to_np
from tests.test_oracle_golden import GOLD, KAT, _params_for

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", TRAJ_CASES, ids=[c[0] for c in TRAJ_CASES])
def test_engine_reproduces_reference_trajectory(case):
    name, algo, n_events, book, _extra, over = case
    fx = np.load(os.path.join(GOLD, "traj_%s.npz" % name))
    traj = fx["traj"]
    rec = engine.gen_stream_host(gen_for(n_events, over), 5, 2, book, 1)
    p = _params_for(over, algo, book)
    eng = engine.Engine(p, 1)
    eng.load_events(rec)
    eng.reset()

    def check_book(i):
        got = dumps_to_np(eng.get_books())[0]
        want = traj[i]["book"]
        for n in want.dtype.names:
            if n in ("cursor",):
                continue
            assert np.array_equal(got[n], want[n]), "%s step %d: book.%s engine=%r reference=%r" % (name, i, n, got[n], want[n])

    check_book(0)
    np.testing.assert_array_equal(eng.get_state()[0], traj[0]["vars"][:eng.V])
    for i in range(1, len(traj)):
        eng.td_step(1)
        assert eng.stepped()[0] == 1
        assert eng.last_actions()[0] == traj[i]["action"], "%s step %d action" % (name, i)
        assert eng.last_rewards()[0] == traj[i]["reward"], "%s step %d reward" % (name, i)
        assert eng.last_td()[0] == traj[i]["td"], "%s step %d td" % (name, i)
        assert eng.rng_counters()[0] == traj[i]["rng_ctr"]
        np.testing.assert_array_equal(eng.learner_state()[0], traj[i]["vars"][:eng.V])
        check_book(i)
    eng.td_step(1)  # the reference episode ended on out-of-data
    assert eng.stepped()[0] == 0 and eng.get_terminal()[0] == 2
    assert eng.counters()[0] == int(fx["steps"])
    th = eng.theta(0)
    nz = np.nonzero(th)[0]
    np.testing.assert_array_equal(nz, fx["theta_idx"])
    np.testing.assert_array_equal(th[nz], fx["theta_val"])


@pytest.mark.parametrize("mem", [20000000, 1 << 20, 999983])
def test_engine_tiles_match_reference(mem):
    p = engine.default_params()
    p.memory_size = mem
    eng = engine.Engine(p, 1)
    np.testing.assert_array_equal(eng.features(KAT["tiles_vars"]), KAT["tiles_%d" % mem])


@pytest.mark.parametrize("mem", [20000000, 4099])
def test_engine_tiles_nonfinite_inputs(mem):
    fx = np.load(os.path.join(GOLD, "kat_nonfinite.npz"))
    p = engine.default_params()
    p.memory_size = mem
    eng = engine.Engine(p, 1)
    np.testing.assert_array_equal(eng.features(np.ascontiguousarray(fx["vars"])), fx["tiles_%d" % mem])


def test_engine_tiles_five_vars():
    p = engine.default_params()
    p.n_vars = 5
    eng = engine.Engine(p, 1)
    np.testing.assert_array_equal(eng.features(KAT["tiles5_vars"]), KAT["tiles5"])


def test_full_size_properties():
    """BASELINE full size (65 536 books, 10 levels): size-independent properties.
    * every book's integer state is self-consistent (positions within bounds +- one order,
      cumulative volumes monotone, cursors advance, RNG counters advance);
    * a replica started from the same seed reproduces the run bit-for-bit up to f64
      atomic-add ordering in theta (book state identical, theta within 1e-9)."""
    B = 65536
    p = engine.default_params()
    p.depth, p.algo, p.theta_mode = 10, abi.ALGO_QLAMBDA, abi.THETA_SHARED
    g = engine.default_gen_params()
    g.n_events = 200
    runs = []
    for _rep in range(2):
        eng = engine.Engine(p, B)
        eng.gen_events(g)
        eng.reset()
        b0 = dumps_to_np(eng.get_books(0, 4096)).copy()
        eng.td_step(40)
        eng.sync()
        b1 = dumps_to_np(eng.get_books(0, 4096)).copy()
        runs.append((b0, b1, eng.theta(), eng.counters(), eng.rng_counters()))
        eng.close()
    b0, b1, th, cnt, rc = runs[0]
    assert cnt[0] == 40 * B and cnt[2] == B
    assert (b1["cursor"] >= b0["cursor"] + 40).all()
    assert (b1["ask_total_volume"] > b0["ask_total_volume"]).all()
    assert (np.abs(b1["position"]) <= 50 + 10).all()
    assert (b1["ask_px"][:, 0] > b1["bid_px"][:, 0]).all()
    assert (rc >= 2 * 40).all()
    # sampled books against a second, independent run
    for name in b1.dtype.names:
        assert np.array_equal(b1[name], runs[1][1][name]), name
    np.testing.assert_array_equal(rc, runs[1][4])
    np.testing.assert_allclose(th, runs[1][2], rtol=1e-9, atol=1e-15)


from tests.golden.make_golden import MULTI_CASES  # noqa: E402
from tests.test_oracle_golden import replay_multi  # noqa: E402
from tests import oracle_lib as ol  # noqa: E402


@pytest.mark.parametrize("case", MULTI_CASES, ids=[c[0] for c in MULTI_CASES])
def test_engine_multi_episode_vs_reference(case):
    """Runner::RunEpisode x N on one engine against the reference: window sums surviving
    ClearWindows (Q7), leftover State on the first step (Q19), HandleTerminal."""
    name, algo, n_events, book, _extra = case
    fx = np.load(os.path.join(GOLD, "multi_%s.npz" % name))
    g = engine.default_gen_params()
    g.n_events = n_events
    rec = engine.gen_stream_host(g, 5, 2, book, 1)
    eng = engine.Engine(_params_for({}, algo, book), 1)
    eng.load_events(rec)

    def rec_fn():
        r = np.zeros(1, dtype=ol.STEP_DTYPE)[0]
        r["book"] = dumps_to_np(eng.get_books())[0]
        r["action"] = eng.last_actions()[0]
        r["reward"] = eng.last_rewards()[0]
        r["td"] = eng.last_td()[0]
        r["rng_ctr"] = eng.rng_counters()[0]
        r["vars"][:8] = eng.learner_state()[0] if eng.stepped()[0] else eng.get_state()[0]
        return r

    replay_multi(fx, eng.reset, lambda: eng.td_step(1), eng.clear_inventory, eng.handle_terminal, rec_fn, name)
    th = eng.theta(0)
    nz = np.nonzero(th)[0]
    np.testing.assert_array_equal(nz, fx["theta_idx"])
    np.testing.assert_array_equal(th[nz], fx["theta_val"])


from tests.test_oracle_golden import DOUBLE_Q_CASES, _check_sparse  # noqa: E402


@pytest.mark.parametrize("case", DOUBLE_Q_CASES, ids=[c[0] for c in DOUBLE_Q_CASES])
def test_engine_double_q_vs_reference(case):
    """rl::DoubleQLearn: both weight vectors, (Qa+Qb)/2 action values, and the update coin drawn
    from a device-side std::mt19937_64 that must reproduce libstdc++'s stream."""
    name, n_events, book = case[:3]
    fx = np.load(os.path.join(GOLD, "traj_%s.npz" % name))
    traj = fx["traj"]
    g = engine.default_gen_params()
    g.n_events = n_events
    rec = engine.gen_stream_host(g, 5, 2, book, 1)
    p = _params_for({}, "sarsa", book)
    p.algo = abi.ALGO_DOUBLE_Q
    if len(case) > 3:  # rl::DoubleRLearn
        p.algo, p.beta = case[3], case[4]
    eng = engine.Engine(p, 1)
    eng.load_events(rec)
    eng.reset()
    for i in range(1, len(traj)):
        eng.td_step(1)
        assert eng.last_actions()[0] == traj[i]["action"], "%s step %d action" % (name, i)
        assert eng.last_td()[0] == traj[i]["td"], "%s step %d td" % (name, i)
        assert eng.rng_counters()[0] == traj[i]["rng_ctr"]
    _check_sparse(eng.theta(0), fx["theta_idx"], fx["theta_val"])
    _check_sparse(eng.theta(1), fx["theta_b_idx"], fx["theta_b_val"])
