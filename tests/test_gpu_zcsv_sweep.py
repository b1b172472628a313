"""Three sweeps the engine runs last (file name sorts after the other GPU files): crafted CSV days, the edges of the
trading window, random configurations over venues and session placement.

The HIP engine on crafted CSV days (tests/test_convert_ref_sweep.py: same-timestamp rows, crossed books, rows the
reference's readers drop or never get past, a time-and-sales file that runs dry first), step by step against the
oracle -- which the CPU suite pins on the same files against the unmodified reference.  Covers the two places where the
record stream carries the reference's streamer semantics into the kernels: LOB_EVT_FLAG_TAS_DRY (no event starts at a
flagged row: out of data with nothing applied) and Initialise's closing SkipUntil (trades dropped when the last warm-up
event ran through an invalid state)."""
import os
import tempfile

import numpy as np
import pytest

from rl_markets_amd import abi, engine
from tests import oracle_lib as ol
from tests.parity import compare_learner_step, dumps_to_np, assert_books_equal
from tests.test_convert_ref_sweep import craft_csvs, T_SLOTS
from tests.test_oracle_golden import _params_for

pytestmark = pytest.mark.gpu


def run_day(seed, algo, tag):
    with tempfile.TemporaryDirectory() as td:
        md, tas = os.path.join(td, "md.csv"), os.path.join(td, "tas.csv")
        craft_csvs(seed, md, tas, ulps=False)   # (prices one float off the grid: pinned on the oracle, not yet run on the GPU)
        try:
            rec = engine.convert_csv(md, tas, T_SLOTS)
        except engine.LobError:
            return None      # fewer than two usable depth rows
    p = _params_for({}, algo, seed % 1000)
    p.memory_size = 1 << 18
    p.max_trades = T_SLOTS
    eng = engine.Engine(p, 1)
    eng.load_events(rec)
    orc = ol.Oracle(p, rec)
    eng.reset()
    orc.reset()
    n = 0
    for step in range(rec.shape[1] + 2):
        eng.td_step(1)
        orc.td_step(1)
        compare_learner_step(eng, orc, "%s step %d" % (tag, step))
        if eng.get_terminal()[0] != 0:
            break
        n += 1
    assert eng.get_terminal()[0] != 0, tag
    assert eng.counters()[0] == orc.counters()[0], tag
    eng.clear_inventory()
    orc.clear_inventory()
    assert_books_equal(dumps_to_np(eng.get_books()), orc.recs()["book"], tag + " after ClearInventory")
    np.testing.assert_array_equal(eng.theta(0), orc.theta(0), err_msg=tag)
    dry = bool((rec[0, :, 1] & abi.EVT_FLAG_TAS_DRY).any())
    eng.close()
    orc.close()
    return n, dry


def test_engine_on_crafted_csv_days():
    steps, dry_days, failed_init = 0, 0, 0
    for k in range(24):
        out = run_day(91000 + k, "sarsa" if k % 2 else "q_learn", "csv day %d" % k)
        if out is None:
            continue
        steps += out[0]
        dry_days += out[1]
        failed_init += out[0] == 0
    # the sample really contains what it is for
    assert steps > 500 and dry_days >= 3 and failed_init >= 1


@pytest.mark.parametrize("fuse", [0, 1])
@pytest.mark.parametrize("ticker,where,algo,theta_mode", [
    ("HSBA.L", "closes", abi.ALGO_QLAMBDA, abi.THETA_SHARED),
    ("CRDI.MI", "closes", abi.ALGO_SARSA, abi.THETA_PRIVATE),
    ("NOKIA.HE", "opens", abi.ALGO_QLAMBDA, abi.THETA_SHARED),
    ("AIR.PA", "opens", abi.ALGO_DOUBLE_Q, abi.THETA_PRIVATE),
])
def test_trading_window_edges(monkeypatch, ticker, where, algo, theta_mode, fuse):
    """Market::IsOpen (market.cpp:67-70: open + 30 min < t < close − 30 min) at both ends, on several venues' session
    times and tick tables: a stream that starts before the window (Initialise's `while not IsOpen` loop eats the early
    rows) and one that runs past it (isTerminal ends the step loop and the episode: terminal = 1, the reference's
    normal end of a day), two episodes each, against the oracle (pinned on the same cases against the reference by
    tests/test_oracle_ref_sweep.py).  `fuse`: action selection inside the env kernel, whose terminal check is its own."""
    if fuse:
        monkeypatch.setenv("LOB_FUSE_ACT", "1")
    B = 12
    p = engine.default_params()
    p.memory_size = 1 << 20
    p.algo, p.theta_mode = algo, theta_mode
    p.book_id_offset = 300
    assert abi.load().lob_market_preset(ticker.encode(), p.market) == 0
    g = engine.default_gen_params()
    g.n_events = 420
    lo, hi = p.market.open_ms + 30 * 60000, p.market.close_ms - 30 * 60000
    g.t0_ms = int(lo - 25 * g.dt_ms) if where == "opens" else int(hi - 160 * g.dt_ms)
    rec = engine.gen_stream_host(g, 5, 2, p.book_id_offset, B)
    eng = engine.Engine(p, B)
    eng.load_events(rec)
    orc = ol.Oracle(p, rec)
    exact = theta_mode == abi.THETA_PRIVATE
    from tests.parity import compare_env
    for episode in range(2):
        eng.reset()
        orc.reset()
        compare_env(eng, orc, "%s %s episode %d reset" % (ticker, where, episode))
        assert (dumps_to_np(eng.get_books())["time_ms"] > lo).all()      # Initialise ends inside the trading window
        for step in range(g.n_events + 2):
            eng.td_step(1)
            orc.td_step(1)
            compare_learner_step(eng, orc, "%s %s episode %d step %d" % (ticker, where, episode, step), exact=exact, rtol=1e-9)
            if (eng.get_terminal() != 0).all():
                break
        want = 1 if where == "closes" else 2
        assert (eng.get_terminal() == want).all(), eng.get_terminal()
        assert eng.counters()[0] == orc.counters()[0]
        eng.clear_inventory()
        orc.clear_inventory()
        compare_env(eng, orc, "%s %s episode %d after ClearInventory" % (ticker, where, episode))
        eng.handle_terminal()
        orc.handle_terminal()
    for which in range(B if exact else 1):
        if exact:
            np.testing.assert_array_equal(eng.theta(which), orc.theta(which))
        else:
            np.testing.assert_allclose(eng.theta(which), orc.theta(which), rtol=1e-9, atol=1e-12)
    if algo == abi.ALGO_DOUBLE_Q:       # theta_b of book b: which = b + n_books (private weights)
        for b in range(B):
            np.testing.assert_array_equal(eng.theta(b + B), orc.theta_b(b))
    eng.close()
    orc.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_configuration_venues_and_sessions(seed):
    """tests/test_gpu_fuzz.py's random configurations with two more dimensions drawn: the venue (session times, tick
    bands: ten of market.cpp's) and where the stream sits in the trading window (inside it, starting before it, running
    past its end) -- the dimensions tests/test_oracle_ref_sweep.py draws against the reference."""
    from tests.test_gpu_fuzz import random_case
    from tests.test_oracle_ref_sweep import TICKERS
    p, g, B = random_case(7000 + seed)
    r = np.random.default_rng(9000 + seed)
    ticker = str(r.choice(TICKERS))
    assert abi.load().lob_market_preset(ticker.encode(), p.market) == 0
    g.dt_ms = int(r.choice([100, 500, 2000]))
    lo, hi = p.market.open_ms + 30 * 60000, p.market.close_ms - 30 * 60000
    where = int(r.integers(0, 3))
    g.t0_ms = int(lo + 30 * 60000) if where == 0 else int(lo - int(r.integers(0, 40)) * g.dt_ms) if where == 1 else \
        int(hi - int(r.integers(30, 220)) * g.dt_ms)
    rec = engine.gen_stream_host(g, p.depth, p.max_trades, p.book_id_offset, B)
    eng = engine.Engine(p, B)
    eng.load_events(rec)
    orc = ol.Oracle(p, rec)
    exact = p.theta_mode == abi.THETA_PRIVATE or B == 1
    tag = "seed %d %s where %d" % (seed, ticker, where)
    for episode in range(2):
        eng.reset()
        orc.reset()
        for step in range(70):
            eng.td_step(1)
            orc.td_step(1)
            compare_learner_step(eng, orc, "%s episode %d step %d" % (tag, episode, step), exact=exact, rtol=1e-9)
        eng.clear_inventory(); orc.clear_inventory()
        eng.handle_terminal(); orc.handle_terminal()
    for which in range(B if p.theta_mode == abi.THETA_PRIVATE else 1):
        a, b = eng.theta(which), orc.theta(which)
        if exact:
            np.testing.assert_array_equal(a, b)
        else:
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)
    eng.close()
    orc.close()
