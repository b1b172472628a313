"""Phase breakdown of env_kernel from an instrumented build (clock64 per phase, summed over lanes):
see DESIGN.md.  The instrumented library is built by hand into rl_markets_amd/csrc/_abl/prof.so."""
import sys, ctypes, shutil, os
sys.path.insert(0, '.')
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, 'rl_markets_amd', 'csrc', 'liblob_engine.so')
shutil.copy(lib, '/tmp/keep.so'); shutil.copy(os.path.join(root, 'rl_markets_amd', 'csrc', '_abl', 'prof.so'), lib)
try:
    from rl_markets_amd import abi, engine
    p = engine.default_params(); p.depth = 10; p.algo = abi.ALGO_QLAMBDA
    g = engine.default_gen_params(); g.n_events = 1200
    eng = engine.Engine(p, 65536); eng.gen_events(g); eng.reset()
    eng.td_step(100); eng.sync()
    out = (ctypes.c_int64 * 32)()
    eng.lib.lob_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
    eng.lib.lob_debug_counters(eng.h, out); a = list(out)
    eng.td_step(100); eng.sync()
    eng.lib.lob_debug_counters(eng.h, out); b = list(out)
    d = [y - x for x, y in zip(a, b)]
    steps = d[0]; names = ['env_load', 'do_action', 'event loop', 'pnl windows', ' ev: track', ' ev: load_trades', ' ev: match', ' ev: order volumes+update', ' ev: adverse+rest', ' events', 'vars+store', 'total']
    print('lane-steps', steps, 'events/step', d[1] / steps)
    for i, n in enumerate(names):
        print('%-28s %10.0f cycles per lane-step' % (n, d[8 + i] / steps))
finally:
    shutil.copy('/tmp/keep.so', lib)
