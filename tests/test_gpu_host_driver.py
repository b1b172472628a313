"""The C++ host adaptors (rl_markets_amd/host/lob_host.hpp: Config, BatchedIntraday,
Agent, Learner with the reference's class shapes) driven through the lob_run
executable on config/engine.yaml (the reference's example.yaml keys and defaults), checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from rl_markets_amd import abi, engine
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lob_run_single_book_matches_oracle(tmp_path):
    exe = os.path.join(ROOT, "rl_markets_amd", "host", "lob_run")
    theta_file = str(tmp_path / "theta.bin")
    events = 500
    stats_file = str(tmp_path / "test_stats.csv")
    out = subprocess.run([exe, "-c", os.path.join(ROOT, "config", "engine.yaml"), "-a", "q_learn", "-n", "1", "-e", "1",
                          "--events", str(events), "--theta", theta_file, "--stats-out", stats_file], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rows = out.stdout.strip().splitlines()
    assert rows[0] == "episode,episode_id,reward,pnl,n_steps,epsilon"
    ep, _id, reward, pnl, n_steps, eps = rows[1].split(",")

    p = engine.default_params()        # == config/engine.yaml (the reference defaults)
    p.algo = abi.ALGO_QLAMBDA
    g = engine.default_gen_params()
    g.n_events = events
    rec = engine.gen_stream_host(g, 5, 2, 0, 1)
    orc = ol.Oracle(p, rec)
    orc.reset()
    for _ in range(events):
        orc.td_step(1)
    orc.clear_inventory()
    r = orc.rec(0)
    assert int(n_steps) == r["book"]["total_ticks"]
    assert float(reward) == pytest.approx(r["book"]["episode_reward"], rel=1e-9)
    # Base::writeStats as the reference leaves it (quirk Q17: three writers truncate one path, the trade statistics survive) and
    # the nTr line of src/main.cpp:234-236
    bk = r["book"]
    ntr = int(bk["ask_transactions"] + bk["bid_transactions"] + bk["market_buys"] + bk["market_sells"])   # (as of the last decision: base.cpp:415-416)
    assert [l for l in rows if l.startswith("stats,")][0].split(",")[1] == str(ntr)
    assert open(stats_file).read().splitlines() == [
        "asks_placed,0", "bids_placed,0", "asks_cancelled,0", "bids_cancelled,0", "ask_transactions,%d" % bk["ask_transactions"],
        "bid_transactions,%d" % bk["bid_transactions"], "market_sells,%d" % bk["market_sells"], "market_buys,%d" % bk["market_buys"]]
    assert float(pnl) == pytest.approx(r["book"]["episode_pnl"], rel=1e-9)
    # EpsilonGreedy::HandleTerminal(0): eps = eps_init * (floor/init)^(0/T) = eps_init
    assert float(eps) == pytest.approx(0.8)
    th = np.fromfile(theta_file, dtype=np.float64)
    np.testing.assert_array_equal(th, orc.theta(0))


def test_lob_run_errors_like_the_reference():
    exe = os.path.join(ROOT, "rl_markets_amd", "host", "lob_run")
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 2 and "Unhandled Exception" in out.stderr
    out = subprocess.run([exe, "-c", os.path.join(ROOT, "config", "engine.yaml"), "-a", "td_zero"], capture_output=True, text=True)
    assert out.returncode == 2 and "Unknown learning algorithm" in out.stderr  # src/main.cpp:187-188
    # every algorithm main.cpp knows runs (the average-reward agents read learning.beta from the config)
    for algo in ("q_learn", "double_q_learn", "r_learn", "online_r_learn", "double_r_learn"):
        out = subprocess.run([exe, "-c", os.path.join(ROOT, "config", "engine.yaml"), "-a", algo, "-n", "1", "-e", "1", "--events", "300"],
                             capture_output=True, text=True)
        assert out.returncode == 0, (algo, out.stderr)


def test_lob_run_backtest_profit_log(tmp_path):
    """Backtester with the reference's profit_log schema (serial.cpp:101-107, intraday.cpp:438-451)
    after one training episode; rows checked against the oracle's greedy evaluation."""
    exe = os.path.join(ROOT, "rl_markets_amd", "host", "lob_run")
    log = str(tmp_path / "profit_log.csv")
    events = 400
    out = subprocess.run([exe, "-c", os.path.join(ROOT, "config", "engine.yaml"), "-a", "sarsa", "-n", "1", "-e", "1",
                          "--events", str(events), "--profit-log", log], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rows = open(log).read().strip().splitlines()
    assert rows[0] == "episode,step,action,position,midprice,spread,quoted_ask,quoted_bid,ask_level,bid_level,pnl_step,bandh_step"
    p = engine.default_params()
    p.algo = abi.ALGO_SARSA
    g = engine.default_gen_params()
    g.n_events = events
    rec = engine.gen_stream_host(g, 5, 2, 0, 1)
    orc = ol.Oracle(p, rec)
    orc.reset()
    for _ in range(events):
        orc.td_step(1)
    orc.clear_inventory()
    ol.load().oracle_handle_terminal(orc.h)
    orc.reset()
    n = 0
    last_bandh = 0.0
    while True:
        before = orc.counters()[0]
        orc.eval_step(1)
        if orc.counters()[0] == before:
            break
        r = orc.rec(0)
        cols = rows[1 + n].split(",")
        assert int(cols[1]) == r["book"]["time_ms"] and int(cols[2]) == r["action"] and int(cols[3]) == r["book"]["position"]
        assert float(cols[10]) == pytest.approx(r["book"]["pnl_step"], rel=1e-9, abs=1e-12)
        assert float(cols[11]) == pytest.approx(r["book"]["episode_bandh"] - last_bandh, rel=1e-6, abs=1e-9)
        last_bandh = r["book"]["episode_bandh"]
        n += 1
    assert n == len(rows) - 1 and n > 50


def test_lob_run_on_reference_csv_pair():
    """lob_run --md/--tas: the reference's own two CSV formats (crafted Q14 day: same-timestamp rows,
    a crossed book) converted and replayed; one book, one episode, against the oracle on the same
    converted records."""
    exe = os.path.join(ROOT, "rl_markets_amd", "host", "lob_run")
    gold = os.path.join(ROOT, "tests", "golden")
    md, tas = os.path.join(gold, "q14_md.csv"), os.path.join(gold, "q14_tas.csv")
    rec = engine.convert_csv(md, tas, 2)
    n = rec.shape[1]
    out = subprocess.run([exe, "-c", os.path.join(ROOT, "config", "engine.yaml"), "-a", "sarsa", "-n", "1", "-e", "1",
                          "--events", str(n), "--md", md, "--tas", tas], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    ep, _id, reward, pnl, n_steps, eps = out.stdout.strip().splitlines()[1].split(",")
    p = engine.default_params()
    p.algo = abi.ALGO_SARSA
    orc = ol.Oracle(p, rec)
    orc.reset()
    for _ in range(n):
        orc.td_step(1)
    orc.clear_inventory()
    r = orc.rec(0)
    assert int(n_steps) == r["book"]["total_ticks"] > 50
    assert float(reward) == pytest.approx(r["book"]["episode_reward"], rel=1e-9)
    assert float(pnl) == pytest.approx(r["book"]["episode_pnl"], rel=1e-9)
    # a missing file is an error, not a silent fall-back to synthetic data
    bad = subprocess.run([exe, "-c", os.path.join(ROOT, "config", "engine.yaml"), "--md", md, "--tas", md + ".nope"],
                         capture_output=True, text=True)
    assert bad.returncode == 2 and "Unhandled Exception" in bad.stderr
