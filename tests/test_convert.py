"""Recorded-data ingestion: the reference's CSV pair and LOBSTER files -> event
records.  CPU tests: converters round-trip generated streams; the oracle fed with the converted
records reproduces a reference run over CSV files containing same-timestamp depth rows, crossed
books and multi-row trade hand-over (quirk Q14)."""
import os

import numpy as np
import pytest

from rl_markets_amd import abi, engine
from tests import oracle_lib as ol
from tests.csv_io import write_lobster, write_reference_csvs
from tests.test_oracle_golden import GOLD, _params_for, compare_traj


def test_csv_round_trip(tmp_path):
    g = engine.default_gen_params()
    g.n_events = 300
    rec = engine.gen_stream_host(g, 5, 2, 3, 1)
    md, tas = str(tmp_path / "md.csv"), str(tmp_path / "tas.csv")
    write_reference_csvs(rec[0], 5, 2, md, tas)
    got = engine.convert_csv(md, tas, 2)
    np.testing.assert_array_equal(got, rec)


def test_csv_drops_bad_rows_and_reports_errors(tmp_path):
    g = engine.default_gen_params()
    g.n_events = 50
    rec = engine.gen_stream_host(g, 5, 2, 3, 1)
    md, tas = str(tmp_path / "md.csv"), str(tmp_path / "tas.csv")
    write_reference_csvs(rec[0], 5, 2, md, tas)
    lines = open(md).read().splitlines()
    cols = lines[10].split(",")
    cols[2] = "0"                      # ask price 0: the reference drops the row (src/data/basic.cpp:54-58)
    lines[10] = ",".join(cols)
    open(md, "w").write("\n".join(lines) + "\n")
    got = engine.convert_csv(md, tas, 2)
    assert got.shape[1] == 49
    with pytest.raises(engine.LobError):
        engine.convert_csv(md, str(tmp_path / "missing.csv"), 2)
    with pytest.raises(engine.LobError) as ei:
        engine.convert_csv(md, tas, 1)  # two trade price levels in one interval do not fit one slot
    assert ei.value.code == abi.LOB_EDATA


def test_csv_short_row_ends_the_day_and_dry_trades_are_flagged(tmp_path):
    """The reference's readers glue lines until a row has its 22 (4) columns (utilities/csv.cpp:31-53 appends,
    basic.cpp:31-43 / 138-150 wait for the exact count), so a line with a column missing is the last thing they ever
    deliver; and Streamer::LoadUntil needs two row groups later than a depth row to serve it (streamer.cpp:61-85)."""
    g = engine.default_gen_params()
    g.n_events = 60
    rec = engine.gen_stream_host(g, 5, 2, 3, 1)
    md, tas = str(tmp_path / "md.csv"), str(tmp_path / "tas.csv")
    write_reference_csvs(rec[0], 5, 2, md, tas)
    full = engine.convert_csv(md, tas, 2)
    assert full.shape[1] == 60 and not (full[0, :, 1] & abi.EVT_FLAG_TAS_DRY).any()   # two sentinel groups: never dry
    lib = abi.load()
    # (1) depth line 41 (row 40) loses its last three columns: rows 0..39 remain
    lines = open(md).read().splitlines()
    keep = lines[:]
    keep[41] = ",".join(keep[41].split(",")[:19])
    open(md, "w").write("\n".join(keep) + "\n")
    cut = engine.convert_csv(md, tas, 2)
    assert cut.shape[1] == 40 and b"depth file cut" in lib.lob_last_error()
    np.testing.assert_array_equal(cut[0, :, 2:], full[0, :40, 2:])
    open(md, "w").write("\n".join(lines) + "\n")
    # (2) without the sentinel groups the stream is dry from the last trade group but one on
    tl = open(tas).read().splitlines()
    assert tl[-1].endswith(",1.0,1") and tl[-2].endswith(",1.0,1")
    open(tas, "w").write("\n".join(tl[:-2]) + "\n")
    times = [l.split(",")[1] for l in tl[1:-2]]
    groups = sorted(set(times))
    dry = engine.convert_csv(md, tas, 2)
    flagged = (dry[0, :, 1] & abi.EVT_FLAG_TAS_DRY) != 0
    from tests.csv_io import ms_to_str
    want = np.array([ms_to_str(int(t)) >= groups[-2] for t in dry[0, :, 0].astype(np.int32)])
    np.testing.assert_array_equal(flagged, want)
    assert flagged.any() and not flagged.all() and b"LOB_EVT_FLAG_TAS_DRY" in lib.lob_last_error()
    # (3) a trade line with a column missing: the trade stream ends before it
    bad = tl[:]
    k = len(bad) // 2
    bad[k] = ",".join(bad[k].split(",")[:3])
    open(tas, "w").write("\n".join(bad) + "\n")
    groups3 = sorted(set(l.split(",")[1] for l in bad[1:k]))
    dry3 = engine.convert_csv(md, tas, 2)
    flagged3 = (dry3[0, :, 1] & abi.EVT_FLAG_TAS_DRY) != 0
    want3 = np.array([ms_to_str(int(t)) >= groups3[-2] for t in dry3[0, :, 0].astype(np.int32)])
    np.testing.assert_array_equal(flagged3, want3)
    assert b"time-and-sales file cut" in lib.lob_last_error()
    # (4) no trades at all: no event can start anywhere
    open(tas, "w").write(tl[0] + "\n")
    none = engine.convert_csv(md, tas, 2)
    assert ((none[0, :, 1] & abi.EVT_FLAG_TAS_DRY) != 0).all()


def test_lobster_round_trip(tmp_path):
    g = engine.default_gen_params()
    g.n_events = 200
    rec = engine.gen_stream_host(g, 10, 2, 9, 1)
    ob, msg = str(tmp_path / "ob.csv"), str(tmp_path / "msg.csv")
    write_lobster(rec[0], 10, 2, 10, ob, msg)
    got = engine.convert_lobster(ob, msg, 10, 10, 2)
    np.testing.assert_array_equal(got, rec)
    got5 = engine.convert_lobster(ob, msg, 10, 5, 2)   # keep the first 5 of 10 levels
    rec5 = engine.gen_stream_host(g, 10, 2, 9, 1)
    assert got5.shape == (1, 200, engine.record_words(5, 2))
    np.testing.assert_array_equal(got5[0, :, 2:7], rec5[0, :, 2:7])          # ask prices 1..5
    np.testing.assert_array_equal(got5[0, :, 12:17], rec5[0, :, 22:27])      # bid prices 1..5


def test_oracle_on_converted_csv_matches_reference():
    fx = np.load(os.path.join(GOLD, "csv_q14.npz"))
    rec = engine.convert_csv(os.path.join(GOLD, "q14_md.csv"), os.path.join(GOLD, "q14_tas.csv"), 2)
    assert (rec[0, :, 1] & 1).sum() == 4          # four rows flagged "same timestamp follows"
    o = ol.Oracle(_params_for({}, "sarsa", 31), rec)
    o.reset()
    compare_traj(lambda: o.td_step(1), lambda: o.rec(0), fx["traj"], "csv_q14")
    o.td_step(1)
    assert o.counters()[0] == int(fx["steps"])
    th = o.theta(0)
    nz = np.nonzero(th)[0]
    np.testing.assert_array_equal(nz, fx["theta_idx"])
    np.testing.assert_array_equal(th[nz], fx["theta_val"])


@pytest.mark.gpu
def test_engine_on_converted_csv_matches_reference():
    from tests.parity import dumps_to_np
    fx = np.load(os.path.join(GOLD, "csv_q14.npz"))
    traj = fx["traj"]
    rec = engine.convert_csv(os.path.join(GOLD, "q14_md.csv"), os.path.join(GOLD, "q14_tas.csv"), 2)
    eng = engine.Engine(_params_for({}, "sarsa", 31), 1)
    eng.load_events(rec)
    eng.reset()
    for i in range(1, len(traj)):
        eng.td_step(1)
        assert eng.last_actions()[0] == traj[i]["action"] and eng.last_td()[0] == traj[i]["td"], "step %d" % i
        got = dumps_to_np(eng.get_books())[0]
        for n in got.dtype.names:
            if n not in ("cursor",):
                assert np.array_equal(got[n], traj[i]["book"][n]), "step %d book.%s %r %r" % (i, n, got[n], traj[i]["book"][n])
    th = eng.theta(0)
    nz = np.nonzero(th)[0]
    np.testing.assert_array_equal(nz, fx["theta_idx"])
    np.testing.assert_array_equal(th[nz], fx["theta_val"])
