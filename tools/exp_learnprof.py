"""Phase breakdown of learn_kernel from an instrumented build (clock64 at phase boundaries, lane 0 of
every wave, summed).  The instrumented library is built by hand into rl_markets_amd/csrc/_abl/prof.so."""
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
    eng.td_step(150); eng.sync()
    out = (ctypes.c_int64 * 32)()
    eng.lib.lob_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
    eng.lib.lob_debug_counters(eng.h, out); a = list(out)
    eng.td_step(100); eng.sync()
    eng.lib.lob_debug_counters(eng.h, out); b = list(out)
    d = [y - x for x, y in zip(a, b)]
    n = d[16]
    names = ['header + LDS staging + barrier', 'F (group-0 tiles of s) + argmax(qs_last)', 'LDS map: init + 288 inserts', 'old generations scan + stores + claim issue',
             'new generation + claim issue', 'Q(s\', .) incl. verdict store', 'argmax / delta / header stores', 'claim finish']
    tot = sum(d[8:16])
    print('waves', n, 'clocks per wave', tot / n)
    for i, nm in enumerate(names):
        print('%-48s %8.0f  %5.1f %%' % (nm, d[8 + i] / n, 100.0 * d[8 + i] / tot))
finally:
    shutil.copy('/tmp/keep.so', lib)
