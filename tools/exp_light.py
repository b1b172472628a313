"""How many books take act_light_kernel per step after a reset, and when an update voids the hit lists (`dirty`)."""
import ctypes, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from rl_markets_amd import abi, engine
p = engine.default_params(); p.depth = 10; p.algo = abi.ALGO_QLAMBDA; p.theta_mode = abi.THETA_SHARED
g = engine.default_gen_params(); g.n_events = 600
B = 65536
eng = engine.Engine(p, B); eng.gen_events(g); eng.reset()
eng.lib.lob_debug_light.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
out = (ctypes.c_int64 * 2)()
prev = 0
for s in range(40):
    t0 = time.perf_counter(); eng.td_step(1); eng.sync(); dt = time.perf_counter() - t0
    eng.lib.lob_debug_light(eng.h, out)
    print("step %2d  light %6d  dirty_sid %3d  %.3f ms" % (s + 1, out[0] - prev, out[1], dt * 1e3))
    prev = out[0]
