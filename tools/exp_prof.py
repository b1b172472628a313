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
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                           "-Wno-unused-value", "-DLOB_PROF", "-o", lib, os.path.join(csrc, "lob_engine.hip"), os.path.join(csrc, "lob_host.cpp")])
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
names = {0: "act: header, memo record, LDS staging, barrier", 1: "act: tile hashing", 2: "act: map words", 3: "act: weights of the maybe-written tiles",
         4: "act: ordered continuation", 5: "act: policy + stores",
         8: "learn: header, Q(s,.), memo record, LDS staging, barrier", 9: "learn: group-0 tiles of s", 10: "learn: argmax + LDS map (288 inserts)",
         11: "learn: old generations (scan, stores, claim issue)", 12: "learn: new generation + claim issue", 13: "learn: tile hashing",
         14: "learn: map words", 15: "learn: weights of the maybe-written tiles", 16: "learn: ordered continuation",
         17: "learn: argmax / delta / header stores", 18: "learn: claim finish"}
for lo, hi, nm in ((0, 8, "act_kernel"), (8, 20, "learn_kernel")):
    tot = sum(d[lo:hi])
    print("%s: %.0f clocks per wave" % (nm, tot))
    for i in range(lo, hi):
        if i in names:
            print("  %-58s %8.0f  %5.1f %%" % (names[i], d[i], 100.0 * d[i] / max(tot, 1e-9)))
