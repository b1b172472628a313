"""The two event passes of env_kernel against each other on the CPU.

tests/host_env/pass_diff.cpp compiles the engine's DEVICE header (rl_markets_amd/csrc/lob_env.h) as host code
through a stand-in for <hip/hip_runtime.h> (tests/host_env/shim -- test infrastructure, nothing in the product
includes it) and throws random order / inventory states at random rows and trade lists: after every pass the
general pass (next_state + step_event, the restatement pinned by the oracle since round 1) and the select-form
fast pass (pass_fast, what env_kernel<., 2, .> runs) must leave bit-identical EnvR and StepAgg.  The GPU suite
then compares the whole kernel with the oracle; this test localises a discrepancy to the pass and runs here."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_pass_equals_general_pass(tmp_path):
    exe = str(tmp_path / "pass_diff")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I" + os.path.join(ROOT, "tests", "host_env", "shim"),
                           "-o", exe, os.path.join(ROOT, "tests", "host_env", "pass_diff.cpp")])
    out = subprocess.run([exe, os.environ.get("LOB_PASS_DIFF_CASES", "300000")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "pass_diff OK" in out.stdout


def test_exp_restatement_equals_libm(tmp_path):
    """rl::Boltzmann::Sample calls std::exp(double); the engine restates glibc 2.35's algorithm on the device (lob_learn.h
    exp_glibc, table lob_exp_table.h).  tools/check_exp.c is the same arithmetic on the host against this libm's exp():
    ordinary Q / tau, the over- and underflow ranges, arbitrary bit patterns, the neighbourhood of 0 -- no difference allowed."""
    exe = str(tmp_path / "check_exp")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-mfma", "-ffp-contract=off", "-I" + os.path.join(ROOT, "rl_markets_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tools", "check_exp.c"), "-lm"])
    out = subprocess.run([exe, os.environ.get("LOB_EXP_CASES", "40000000")], capture_output=True, text=True)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout[-1500:]
    # the table in the header is the library's own (tools/gen_exp_table.py reads it back from libm.so.6)
    gen = subprocess.run(["python3", os.path.join(ROOT, "tools", "gen_exp_table.py")], capture_output=True, text=True)
    if gen.returncode == 0:   # (a libm without this __exp_data layout: nothing to compare)
        hdr = open(os.path.join(ROOT, "rl_markets_amd", "csrc", "lob_exp_table.h")).read()
        for line in gen.stdout.splitlines()[1:]:
            assert line.strip().rstrip("\\").rstrip().rstrip(",") in hdr


def test_same_cell_arithmetic_equals_tile_coord_and_implies_same_index(tmp_path):
    """trace_lane_kernel decides which tiles an old trace generation loses to a new state from the quantised coordinates
    alone (lob_tiles.h tile_same_cell_mask).  tests/host_env/cell_diff.cpp, compiled from the device header: the mask equals
    the tiling-by-tiling comparison through tile_coord for random pairs of triples (near, 2 048 k apart, at both ends of the
    plain range), and a same cell always means the same weight index under the real hash, for every action and table size;
    the converse (an index shared by different cells) is what the tile registry exists for."""
    exe = str(tmp_path / "cell_diff")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "tests", "host_env", "shim"),
                           "-o", exe, os.path.join(ROOT, "tests", "host_env", "cell_diff.cpp")])
    out = subprocess.run([exe, os.environ.get("LOB_CELL_DIFF_CASES", "200000")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "cell_diff OK" in out.stdout


def test_tile_registry_flags_every_shared_index(tmp_path):
    """The tile registry (lob_tiles.h tile_register -- the engine's own function --, registry_block's per-slot loop and
    registry_scan_block restated serially in tests/host_env/registry_diff.cpp): whenever two registered tiles share a weight
    index without being the same tile, both carry their bit in mk_amb -- after every step's registrations and scan, for random
    batches of memo slots in random order (cells that coincide, twins 2 048 apart, three and more tiles on one index, tables
    from 4 099 to 1 M weights) -- and no tile is flagged without reason.  That invariant is what allows trace_lane_kernel to
    compare tile indices only where the bits are set."""
    exe = str(tmp_path / "registry_diff")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "tests", "host_env", "shim"),
                           "-o", exe, os.path.join(ROOT, "tests", "host_env", "registry_diff.cpp")])
    out = subprocess.run([exe, os.environ.get("LOB_REGISTRY_DIFF_TRIALS", "40")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "registry_diff OK" in out.stdout


def test_ticks_hint_equals_to_ticks(tmp_path):
    """The pre-pass converts five prices per event with lobh::to_ticks_hint (band hint instead of the band search, the loop's first
    iteration written out).  tests/host_env/ticks_diff.cpp: against lobh::to_ticks_t -- the restatement of Market::ToTicks pinned
    on the reference's known answers -- on all 14 venue tables, band boundaries +- fractions of a tick, float32 prices, tick
    counts used as prices, non-finite inputs, for every possible hint."""
    exe = str(tmp_path / "ticks_diff")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "host_env", "ticks_diff.cpp"),
                           os.path.join(ROOT, "rl_markets_amd", "csrc", "lob_host.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "ticks_diff OK" in out.stdout, out.stdout[-1500:]
