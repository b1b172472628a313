"""lob::Config (rl_markets_amd/host/lob_host.hpp) on CPU: the yaml keys of the reference's
example.yaml folded into lob_params the way Base / Intraday / Agent / main.cpp read them
(src/environment/base.cpp:14-115, src/rl/agent.cpp:13-60, src/main.cpp:140-189).
Compiles a probe against the header and the C-ABI library; no device call is made."""
import os
import subprocess

import pytest

from rl_markets_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rl_markets_amd", "csrc")
YAML = os.path.join(ROOT, "config", "engine.yaml")

PROBE = r'''
#include <cstdio>
#include <exception>
#include "lob_host.hpp"
int main(int argc, char** argv) {
    try {
        lob::Config c(argv[1]);
        for (int i = 2; i + 1 < argc; i += 2) c.set(argv[i], argv[i + 1]);
        lob_params p = c.to_params("HSBA.L");
        std::printf("%d %d %.17g %.17g %.17g %.17g %.17g %d %.17g %d %d %lld %d %d %d %d %d %d %d %d\n", (int)p.algo, (int)p.policy, p.beta, p.epsilon,
                    p.tau, p.gamma, p.alpha, (int)p.target_price, (double)p.group_weights[2], (int)p.n_tilings, (int)p.n_actions,
                    (long long)p.memory_size, (int)p.quote_mode, (int)p.market.n_bands, (int)p.lb_rsi, (int)p.lb_vwap,
                    (int)p.lb_pnl, (int)p.lb_spread, (int)p.lb_target, (int)p.random_init);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "EXC %s\n", e.what());
        return 3;
    }
    return 0;
}
'''


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    td = tmp_path_factory.mktemp("cfgprobe")
    src = td / "probe.cpp"
    src.write_text(PROBE)
    exe = str(td / "probe")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "rl_markets_amd", "host"), "-o", exe, str(src),
                           "-L" + CSRC, "-llob_comm", "-llob_engine", "-Wl,-rpath," + CSRC, "-L/opt/rocm/lib",
                           "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link," + CSRC + ":/opt/rocm/lib"])

    def run(yaml=YAML, **over):
        args = [exe, yaml]
        for k, v in over.items():
            args += [k.replace("__", "."), str(v)]
        return subprocess.run(args, capture_output=True, text=True)
    return run


def fields(out):
    assert out.returncode == 0, out.stderr
    f = out.stdout.split()
    return dict(algo=int(f[0]), policy=int(f[1]), beta=float(f[2]), epsilon=float(f[3]), tau=float(f[4]), gamma=float(f[5]),
                alpha=float(f[6]), target_price=int(f[7]), gw2=float(f[8]), n_tilings=int(f[9]), n_actions=int(f[10]),
                memory_size=int(f[11]), quote_mode=int(f[12]), n_bands=int(f[13]), lb_rsi=int(f[14]), lb_vwap=int(f[15]),
                lb_pnl=int(f[16]), lb_spread=int(f[17]), lb_target=int(f[18]), random_init=int(f[19]))


def test_engine_yaml_is_the_reference_example(probe):
    f = fields(probe())
    assert f["algo"] == abi.ALGO_DOUBLE_Q and f["policy"] == 0
    assert f["epsilon"] == float.fromhex("0x1.99999a0000000p-1")   # eps_init read as float (policy.cpp:58-67)
    assert (f["gamma"], f["alpha"]) == (0.975, 0.001)
    assert f["target_price"] == abi.TP_MICROPRICE                  # quirk Q5
    assert abs(f["gw2"] - 0.10) < 1e-12
    assert (f["n_tilings"], f["n_actions"], f["memory_size"], f["n_bands"]) == (32, 9, 20000000, 10)


@pytest.mark.parametrize("name,code", [("sarsa", "ALGO_SARSA"), ("q_learn", "ALGO_QLAMBDA"), ("double_q_learn", "ALGO_DOUBLE_Q"),
                                       ("r_learn", "ALGO_R_LEARN"), ("online_r_learn", "ALGO_ONLINE_R_LEARN"),
                                       ("double_r_learn", "ALGO_DOUBLE_R_LEARN")])
def test_algorithm_names_of_main_cpp(probe, name, code):
    f = fields(probe(learning__algorithm=name, learning__beta=0.0125))
    assert f["algo"] == getattr(abi, code)
    # beta is only read by the average-reward agents (src/rl/agent.cpp RLearn ctor)
    assert f["beta"] == (0.0125 if "r_learn" in name else 0.005)


def test_unknown_algorithm_and_policy_throw(probe):
    out = probe(learning__algorithm="td_zero")
    assert out.returncode == 3 and "Unknown learning algorithm" in out.stderr    # src/main.cpp:187-188
    out = probe(policy__type="softmax")
    assert out.returncode == 3 and "valid policy" in out.stderr                  # src/main.cpp:164-165


def test_r_learning_requires_beta(probe, tmp_path):
    text = "".join(l for l in open(YAML) if not l.strip().startswith("beta:"))
    y = tmp_path / "nobeta.yaml"
    y.write_text(text)
    assert fields(probe(str(y), learning__algorithm="q_learn"))["algo"] == abi.ALGO_QLAMBDA
    out = probe(str(y), learning__algorithm="r_learn")
    assert out.returncode == 3 and "learning.beta" in out.stderr   # YAML::Node::as<double>() on a missing key throws


def test_random_init_is_read_like_agent_cpp(probe):
    """learning.random_init (src/rl/agent.cpp:37-39: c["learning"]["random_init"].as<bool>(false)): absent = false; yaml-cpp's
    boolean spellings; anything else throws as YAML's bad conversion does."""
    assert fields(probe())["random_init"] == 0
    for v, want in (("true", 1), ("True", 1), ("yes", 1), ("on", 1), ("false", 0), ("no", 0), ("OFF", 0)):
        assert fields(probe(learning__random_init=v))["random_init"] == want, v
    out = probe(learning__random_init="maybe")
    assert out.returncode == 3 and "learning.random_init" in out.stderr


def test_policy_factory(probe):
    assert fields(probe(policy__type="greedy"))["epsilon"] == 0.0
    assert fields(probe(policy__type="random"))["epsilon"] == 1.0
    f = fields(probe(policy__type="boltzmann", policy__tau_init=0.3))
    assert f["policy"] == abi.POLICY_BOLTZMANN and f["tau"] == float.fromhex("0x1.3333340000000p-2")  # float, main.cpp:156-157


def test_target_price_quirk_and_quote_mode(probe):
    # anything but "midprice" builds MidPrice (base.cpp:101-112); "book" quotes off the book (intraday.cpp:64)
    assert fields(probe(**{"market__target_price__type": "microprice"}))["target_price"] == abi.TP_MIDPRICE
    f = fields(probe(**{"market__target_price__type": "book"}))
    assert f["quote_mode"] == abi.QUOTE_BOOK


def test_lookbacks_are_clamped_like_base_cpp(probe):
    """Every window is max(lookback, 1) (base.cpp:35-50: example.yaml's rsi / vwap / pnl look-backs are 0) except the
    target price's (base.cpp:102), which goes through as written -- lob_create refuses 0 where the reference has no price
    to quote at."""
    f = fields(probe())
    assert (f["lb_rsi"], f["lb_vwap"], f["lb_pnl"], f["lb_spread"], f["lb_target"]) == (1, 1, 1, 45, 1)
    f = fields(probe(**{"policy__spread_lookback": 0, "market__target_price__lookback": 0, "state__lookback__rsi": 14}))
    assert (f["lb_spread"], f["lb_target"], f["lb_rsi"]) == (1, 0, 14)


REF_YAML = "/root/reference/config/example.yaml"


@pytest.mark.skipif(not os.path.exists(REF_YAML), reason="needs the reference checkout")
def test_the_references_own_example_yaml_reads_the_same(tmp_path):
    """lob::Config on the reference's config/example.yaml, unedited: the same lob_params, byte for byte, as on
    config/engine.yaml (which documents the keys) -- a user's existing config file works as it is."""
    src = tmp_path / "cmp.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstring>
#include "lob_host.hpp"
int main(int, char** argv) {
    lob::Config a(argv[1]), b(argv[2]);
    lob_params pa = a.to_params("HSBA.L"), pb = b.to_params("HSBA.L"), pd;
    lob_default_params(&pd);
    std::printf("%d %d\n", std::memcmp(&pa, &pb, sizeof pa), std::memcmp(&pa, &pd, sizeof pa));
    return 0;
}
''')
    exe = str(tmp_path / "cmp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "rl_markets_amd", "host"), "-o", exe, str(src),
                           "-L" + CSRC, "-llob_comm", "-llob_engine", "-Wl,-rpath," + CSRC, "-L/opt/rocm/lib",
                           "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link," + CSRC + ":/opt/rocm/lib"])
    same_as_engine_yaml, same_as_defaults = subprocess.check_output([exe, REF_YAML, YAML]).split()
    assert int(same_as_engine_yaml) == 0
    # lob_default_params is example.yaml except for its algorithm (double_q_learn there, SARSA in the struct's defaults)
    p = abi.Params()
    abi.load().lob_default_params(p)
    assert p.algo == abi.ALGO_SARSA
