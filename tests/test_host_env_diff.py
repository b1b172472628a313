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
