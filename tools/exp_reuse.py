"""How often can act_kernel reuse learn_kernel's verdicts?  Prints, along a training run at the
headline configuration, the number of weights written for the first time per update."""
import sys, time, ctypes
sys.path.insert(0, '.')
from rl_markets_amd import abi, engine
p = engine.default_params(); p.depth = 10; p.algo = abi.ALGO_QLAMBDA
g = engine.default_gen_params(); g.n_events = 2112
eng = engine.Engine(p, 65536); eng.gen_events(g); eng.reset()
lib = eng.lib
lib.lob_debug_new_weights.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]
out = (ctypes.c_int32 * 2)()
done = 0
eng.kernel_timing(True)
for chunk in (1, 4, 16, 32, 64, 128, 256, 256, 256):
    t = time.perf_counter(); eng.td_step(chunk); eng.sync(); dt = time.perf_counter() - t
    done += chunk
    lib.lob_debug_new_weights(eng.h, out)
    print('steps', done, 'ms/step', round(dt / chunk * 1e3, 3), 'new weights (last two updates)', out[0], out[1],
          {k[:-7]: round(eng.kernel_time_ms(k)[0], 3) for k in ('act_kernel', 'env_kernel', 'learn_kernel', 'update_kernel')}, flush=True)
    eng.kernel_timing(True)
eng.close()
