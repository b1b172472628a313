"""Phase breakdown of act_kernel / learn_kernel from an instrumented build (clock64 at phase boundaries,
lane 0 of every wave; lob_learn.h `Prof`).  Builds rl_markets_amd/csrc/_prof/liblob_engine.so with
-DLOB_PROF (here or on the GPU box), runs the headline configuration through it, prints clocks per wave.
    python tools/exp_prof.py [--build-only]"""
import ctypes
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
csrc = os.path.join(root, "rl_markets_amd", "csrc")
out_dir = os.path.join(csrc, "_prof")
lib = os.path.join(out_dir, "liblob_engine.so")
os.makedirs(out_dir, exist_ok=True)
srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".h"))]
if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
    import __graft_entry__ as ge
    ge.build_engine(lib, ("-DLOB_PROF",), os.path.join(out_dir, "_obj"))
if "--build-only" in sys.argv:
    sys.exit(0)

from rl_markets_amd import abi
abi.LIB_PATH = lib
from rl_markets_amd import engine

p = engine.default_params(); p.depth = 10; p.algo = abi.ALGO_QLAMBDA
g = engine.default_gen_params(); g.n_events = 1200
B = 65536
eng = engine.Engine(p, B); eng.gen_events(g); eng.reset()
eng.td_step(60); eng.sync()
eng.lib.lob_debug_prof.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
out = (ctypes.c_int64 * 32)()
assert eng.lib.lob_debug_prof(eng.h, out) == 0
a = list(out)
N = 100
eng.td_step(N); eng.sync()
assert eng.lib.lob_debug_prof(eng.h, out) == 0
d = [(y - x) / (N * B) for x, y in zip(a, list(out))]
for i in range(20, 32):
    d[i] *= 64.0   # the lane kernels stamp once per 64-book wave
names = {0: "act: headers, memo records, state variables", 1: "act: tile hashing", 2: "act: coarse filter (LDS)", 3: "act: exact map for the coarse hits",
         4: "act: written weights + ordered continuation", 5: "act: policy + stores",
         8: "trace: header, Q(s,.), state variables", 9: "trace: group-0 tiles of s", 10: "trace: argmax + LDS map (288 inserts)",
         11: "trace: old generations (scan, stores, claim issue)", 12: "trace: new generation + claim issue", 18: "trace: header stores, previous book's claims",
         13: "learn_q: headers, memo records, state variables", 14: "learn_q: tile hashing", 15: "learn_q: coarse filter (LDS)",
         16: "learn_q: exact map for the coarse hits", 17: "learn_q: written weights + ordered continuation", 19: "learn_q: argmax / delta / header stores"}
names.update({30: "env: action selection (hit list replay, policy, header stores)", 20: "env: launch, round 1 (loads addressed by the book id), scalars to LDS",
              31: "env: round 2 (memo record, listed weights, track entries, rows) issued and in",
              21: "env: DoAction (quotes, tick conversions, queue position)", 22: "env: hot copy LDS -> registers, order keys",
              23: "env: per pass (lane 0's own): loop top, next entry / row requested", 24: "env: per pass (lane 0's own): the pass",
              25: "env: waiting for the wave's slowest lane", 26: "env: hot copy back", 27: "env: PnL windows",
              28: "env: state variables (track entry, 8 variables, memo look-up issued)", 29: "env: agent scalars out, memo claim"})
print("(clocks per wave ITERATION: the Q kernels take LOB_FAST_NB books per iteration and stamp the batch on its first book's row)")
for lo, hi, nm, idx in ((20, 32, "env_kernel (per WAVE-step: 64 books; the per-pass rows are summed over lane 0's passes)", (20, 31, 30, 21, 22, 23, 24, 25, 26, 27, 28, 29)), (0, 8, "act_kernel", range(0, 8)), (8, 20, "trace_kernel", (8, 9, 10, 11, 12, 18)), (13, 20, "learn_q kernel", (13, 14, 15, 16, 17, 19))):
    tot = sum(d[i] for i in idx)
    print("%s: %.0f clocks per wave" % (nm, tot))
    for i in idx:
        if i in names:
            print("  %-58s %8.0f  %5.1f %%" % (names[i], d[i], 100.0 * d[i] / max(tot, 1e-9)))
