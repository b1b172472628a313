import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rl_markets_amd import abi, engine
p = engine.default_params(); p.depth, p.max_trades = 10, 2; p.algo = abi.ALGO_QLAMBDA; p.theta_mode = abi.THETA_SHARED; p.memory_size = 20000000
g = engine.default_gen_params(); g.n_events = 2112
eng = engine.Engine(p, 65536); eng.gen_events(g)
eng.kernel_timing(True)
try:
    eng.reset()
except Exception as ex:
    print("reset raised", ex)
eng.sync() if False else None
print(os.environ.get("LOB_DBG_PREPASS"), os.environ.get("LOB_PREPASS_ROLES"), "reset_ms", eng.kernel_time_ms("reset_kernel"))
