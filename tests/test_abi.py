"""The C-ABI library: loads, exports every symbol include/lob_engine.h declares,
host-side helpers work, and without a GPU the engine refuses to run (no CPU
fallback).  CPU only -- no compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rl_markets_amd import abi, engine
from tests.conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "lob_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lob_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    lib = abi.load()
    names = header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "liblob_engine.so does not export %s" % n
    # and the ctypes mirror covers the whole header
    assert set(names) == set(lib._declared)


def test_comm_library_exports_its_header():
    """include/lob_comm.h (multi-GPU weight exchange, RCCL) against liblob_comm.so and its ctypes mirror."""
    from rl_markets_amd import comm
    src = open(os.path.join(ROOT, "include", "lob_comm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(lob_[a-z_0-9]+)\s*\(", src)))
    lib = comm.load()
    assert "lob_theta_allreduce" in names and len(names) >= 10
    for n in names:
        assert hasattr(lib, n), "liblob_comm.so does not export %s" % n
    assert set(names) == set(lib._declared)


def test_struct_sizes_match_header():
    # compile a probe with the real header and compare sizeof with the ctypes mirror
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "probe.c")
        open(src, "w").write('#include <stdio.h>\n#include "lob_engine.h"\nint main(){printf("%zu %zu %zu %zu\\n",'
                             'sizeof(lob_market),sizeof(lob_params),sizeof(lob_gen_params),sizeof(lob_book_dump));return 0;}')
        exe = os.path.join(td, "probe")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        sizes = list(map(int, subprocess.check_output([exe]).split()))
    assert sizes == [C.sizeof(abi.Market), C.sizeof(abi.Params), C.sizeof(abi.GenParams), C.sizeof(abi.BookDump)]


def test_default_params_are_example_yaml():
    p = engine.default_params()
    assert (p.depth, p.n_vars, p.order_size, p.pos_lb, p.pos_ub) == (5, 8, 10, -50, 50)
    assert list(p.vars[:8]) == [abi.VAR_POS, abi.VAR_A_DIST, abi.VAR_B_DIST, abi.VAR_MPM, abi.VAR_SPD, abi.VAR_VOL,
                                abi.VAR_IMB, abi.VAR_SVL]
    assert (p.lb_mpm, p.lb_vlt, p.lb_svl, p.lb_spread) == (15, 60, 60, 45)
    assert p.memory_size == 20000000 and p.n_tilings == 32 and p.n_actions == 9
    assert list(p.group_weights) == [0.65, 0.25, 0.10]
    assert (p.gamma, p.lambda_, p.alpha, p.epsilon) == (0.975, 0.85, 0.001, 0.8)
    assert p.reward_measure == abi.REWARD_PNL_DAMPED and abs(p.damping_factor - 0.15) < 1e-7
    assert p.target_price == abi.TP_MICROPRICE  # quirk Q5: "midprice" instantiates MicroPrice
    assert p.market.open_ms == 8 * 3600000 and p.market.close_ms == 16 * 3600000 + 30 * 60000


def test_market_presets_and_errors():
    lib = abi.load()
    m = abi.Market()
    assert lib.lob_market_preset(b"HSBA.L", C.byref(m)) == 0 and m.n_bands == 10
    assert lib.lob_market_preset(b"CRDI.MI", C.byref(m)) == 0 and m.open_ms == 9 * 3600000
    assert lib.lob_market_preset(b"XXXX.L", C.byref(m)) == abi.LOB_EINVAL      # reference throws invalid_argument
    assert lib.lob_market_preset(b"HSBA.ZZ", C.byref(m)) == abi.LOB_EINVAL
    assert b"unknown venue" in lib.lob_last_error()
    lib.lob_market_preset(b"HSBA.L", C.byref(m))
    t = C.c_int32()
    assert lib.lob_to_ticks(C.byref(m), 702.1, C.byref(t)) == 0 and t.value == 46021  # test/test_Market.cpp:45
    assert lib.lob_to_ticks(C.byref(m), 702.5, C.byref(t)) == 0 and t.value == 46025
    assert lib.lob_to_ticks(C.byref(m), -1.0, C.byref(t)) == abi.LOB_EINVAL           # reference throws
    lib.lob_market_preset(b"AAL.L", C.byref(m))
    pr = C.c_double()
    assert lib.lob_to_ticks(C.byref(m), 2750.0, C.byref(t)) == 0 and t.value == 52500  # test_Market.cpp:26-28
    assert lib.lob_to_price(C.byref(m), 52500, C.byref(pr)) == 0 and pr.value == 2750.0


def test_stream_generator_is_deterministic_and_valid():
    g = engine.default_gen_params()
    g.n_events = 300
    a = engine.gen_stream_host(g, 10, 2, 5, 4)
    b = engine.gen_stream_host(g, 10, 2, 7, 2)
    np.testing.assert_array_equal(a[2:], b)          # book ids, not positions, seed the streams
    lib = abi.load()
    assert lib.lob_validate_stream(a.ctypes.data_as(C.c_void_p), 10, 2, 4, 300) == 0
    W = engine.record_words(10, 2)
    assert W == 48 and a.shape == (4, 300, W)
    px = a[..., 2:12].view(np.float32)
    assert (np.diff(px, axis=-1) > 0).all()           # asks ascending
    bad = a.copy()
    bad[0, 7, 2] = bad[0, 7, 3]                       # duplicate ask price key
    assert lib.lob_validate_stream(bad.ctypes.data_as(C.c_void_p), 10, 2, 4, 300) == abi.LOB_EDATA
    bad = a.copy()
    bad[1, 9, 0] = 5                                  # time goes backwards
    assert lib.lob_validate_stream(bad.ctypes.data_as(C.c_void_p), 10, 2, 4, 300) == abi.LOB_EDATA


def test_big_streams_are_validated_by_threads_and_report_the_first_book():
    """Streams of 2^18 records and more are checked by several host threads over contiguous book ranges: the offence reported is
    that of the lowest book (its first event), as the serial scan reports it."""
    g = engine.default_gen_params()
    g.n_events = 600
    a = engine.gen_stream_host(g, 10, 2, 0, 512)
    lib = abi.load()
    assert lib.lob_validate_stream(a.ctypes.data_as(C.c_void_p), 10, 2, 512, 600) == 0
    bad = a.copy()
    bad[400, 7, 2] = bad[400, 7, 3]      # duplicate ask price key, a book of a later range
    bad[37, 500, 0] = 5                  # time goes backwards
    bad[37, 9, 2] = bad[37, 9, 3]
    assert lib.lob_validate_stream(bad.ctypes.data_as(C.c_void_p), 10, 2, 512, 600) == abi.LOB_EDATA
    assert lib.lob_last_error().startswith(b"book 37 event 9 ")


def test_bad_params_rejected():
    lib = abi.load()
    p = engine.default_params()
    h = C.c_void_p()
    p.n_tilings = 16
    assert lib.lob_create(C.byref(p), 4, 0, C.byref(h)) == abi.LOB_EINVAL
    p = engine.default_params()
    p.depth = 11
    assert lib.lob_create(C.byref(p), 4, 0, C.byref(h)) == abi.LOB_EINVAL


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_gpu_means_no_engine():
    lib = abi.load()
    p = engine.default_params()
    h = C.c_void_p()
    assert lib.lob_create(C.byref(p), 4, 0, C.byref(h)) == abi.LOB_ENODEV
    assert b"no CPU fallback" in lib.lob_last_error()
    with pytest.raises(engine.LobError):
        engine.Engine(p, 4)


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_gpu_drivers_fail_loudly():
    """The C++ driver and the bench on a box without a GPU: an error that says so and a non-zero exit, not a CPU run."""
    import subprocess
    import sys
    exe = os.path.join(ROOT, "rl_markets_amd", "host", "lob_run")
    out = subprocess.run([exe, "-c", os.path.join(ROOT, "config", "engine.yaml"), "-a", "q_learn", "-n", "1", "-e", "1", "--events", "300"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "no usable HIP device" in out.stderr and out.stdout.strip() == ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--books", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "no usable HIP device" in (out.stderr + out.stdout)
    assert '"metric"' not in out.stdout        # no result line without a GPU


def test_product_does_not_reference_the_oracle():
    """Nothing under rl_markets_amd/ may include, import, link or execute anything under oracle/ or tests/."""
    pat = re.compile(r"#\s*include[^\n]*oracle|liblob_oracle|oracle_lib|ref_harness|from\s+tests|import\s+tests|oracle/")
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "rl_markets_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not pat.search(txt), "%s reaches into the oracle" % f


def test_traffic_table_is_what_the_committed_counter_passes_give(tmp_path):
    """profiles/pmc_traffic*.json (bench.py's roofline.traffic, one file per quoted configuration) regenerated from the committed
    rocprofv3 counter summaries of the round."""
    import json
    import subprocess
    import sys
    for tag, name in (("r06", "pmc_traffic.json"), ("r06_sarsa", "pmc_traffic_sarsa.json"), ("r06_c2", "pmc_traffic_c2.json"),
                      ("r06_double_q", "pmc_traffic_double_q.json"), ("r06_c5", "pmc_traffic_c5.json"), ("r06_eps01", "pmc_traffic_eps01.json")):
        out = str(tmp_path / name)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_traffic.py"), tag, out], stdout=subprocess.DEVNULL)
        assert json.load(open(out)) == json.load(open(os.path.join(ROOT, "profiles", name))), name
