"""The HIP engine on crafted CSV days (tests/test_convert_ref_sweep.py: same-timestamp rows, crossed books, rows the
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
        craft_csvs(seed, md, tas)
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
