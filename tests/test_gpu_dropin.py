"""The drop-in, demonstrated: the reference's UNMODIFIED experiment::serial::Learner::RunEpisode and
rl::SARSA / rl::QLearn (compiled from /root/reference into oracle/_ref/ref_dropin, see oracle/Makefile)
drive environment::GpuIntraday -- a subclass of the reference's own environment::Base
(rl_markets_amd/host/ref_binding/gpu_intraday.h) whose book lives in the HIP engine -- through the very
call sites they use for environment::Intraday<>, and reproduce, step for step and bit for bit, the
trajectories the all-CPU reference produced (tests/golden/traj_*.npz): actions, rewards (the
reference's non-virtual Base::getReward() evaluated on the engine's mirrored members), TD errors, state
variables, RNG draw counts, the full book / order / position state, and the learned weights."""
import json
import os
import subprocess

import numpy as np
import pytest

from rl_markets_amd import engine
from tests import oracle_lib as ol
from tests.golden.make_golden import TRAJ_CASES, gen_for
from tests.test_oracle_golden import GOLD

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "oracle", "_ref", "ref_dropin")

# the fixtures whose reward Base::getReward() can compute from mirrored members, default state variables
CASES = [c for c in TRAJ_CASES if c[0] in ("sarsa_b0", "qlearn_b3", "sarsa_mm_linear_b11", "sarsa_tight_bounds_b7", "sarsa_book_quotes_b9",
                                          "qlearn_mm_div_b16", "sarsa_lovol_b18", "sarsa_mm_exp_b20", "sarsa_boltzmann_b23")]


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/ref_dropin is built where the reference checkout is (make -C oracle dropin)")
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_reference_learner_drives_the_gpu_environment(tmp_path, case):
    name, algo, n_events, book, extra, over = case
    fx = np.load(os.path.join(GOLD, "traj_%s.npz" % name))
    want = fx["traj"][1:]  # (the fixture's first row is the state right after the reset)
    rec = engine.gen_stream_host(gen_for(n_events, over), 5, 2, book, 1)
    sp, out, th = str(tmp_path / "s.bin"), str(tmp_path / "t.traj"), str(tmp_path / "theta.bin")
    rec[0].tofile(sp)
    cmd = [DROPIN, "dropin", "--stream", sp, "--events", str(n_events), "--book", "0", "--depth", "5", "--trades", "2",
           "--algo", algo, "--mem", str(1 << 20), "--seed", "1994", "--rng_stream", str(book), "--eps", "0.8", "--out", out,
           "--theta_out", th, "--tmp", str(tmp_path / "h")]
    for k, v in extra.items():
        cmd += ["--" + k, str(v)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["sizeof_steprec"] == ol.STEP_DTYPE.itemsize and info["ok"] == 1
    got = np.fromfile(out, dtype=ol.STEP_DTYPE)
    assert len(got) == len(want) == int(fx["steps"]) == info["steps"]
    for f in ("action", "reward", "td", "rng_ctr", "n_vars"):
        np.testing.assert_array_equal(got[f], want[f], err_msg="%s: %s" % (name, f))
    np.testing.assert_array_equal(got["vars"], want["vars"], err_msg="%s: state variables" % name)
    for f in want["book"].dtype.names:
        if f == "cursor":
            continue
        assert np.array_equal(got["book"][f], want["book"][f]), "%s: book.%s first differs at step %d" % (
            name, f, int(np.argwhere(got["book"][f] != want["book"][f])[0][0]))
    # the weights the reference's agent learned from the GPU environment's rewards and states
    raw = np.fromfile(th, dtype=np.uint8)
    n = int(np.frombuffer(raw[:8].tobytes(), dtype=np.int64)[0])
    pairs = np.frombuffer(raw[8:8 + 16 * n].tobytes(), dtype=[("i", np.int64), ("v", np.float64)])
    np.testing.assert_array_equal(pairs["i"], fx["theta_idx"])
    np.testing.assert_array_equal(pairs["v"], fx["theta_val"])
    # Base::getEpisodeReward() / getEpisodePnL(): the reference's getters over the mirrored members
    # (RunEpisode's epilogue, ClearInventory through a Base&, is the reference's no-op on Base's own empty books)
    assert info["episode_reward"] == want["book"]["episode_reward"][-1]


def test_binding_rejects_window_rewards(tmp_path):
    """`spread` / `normed` read Base's rolling windows inside the NON-virtual Base::getReward(): the binding
    refuses them instead of returning a reward computed on empty windows."""
    if not os.path.exists(DROPIN):
        pytest.skip("no ref_dropin")
    rec = engine.gen_stream_host(gen_for(200, {}), 5, 2, 0, 1)
    sp = str(tmp_path / "s.bin")
    rec[0].tofile(sp)
    res = subprocess.run([DROPIN, "dropin", "--stream", sp, "--events", "200", "--book", "0", "--algo", "sarsa", "--mem", "4096",
                          "--reward", "spread", "--tmp", str(tmp_path / "h")], capture_output=True, text=True)
    assert res.returncode != 0 and "getReward" in res.stderr
