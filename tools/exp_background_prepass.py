"""Probe (round 6): what a market pre-pass running BESIDE the learner steps costs them.  Two engines on one GPU, each on its own
stream: engine A steps 65 536 books, engine B (a thread of its own) runs lob_reset -- the 16 ms pre-pass, 1 024 waves that own a
SIMD's whole register file -- over and over.  A's step rate with and without B, and B's reset time with and without A: the
worst case (no co-residency of the two kernels' waves) of hiding the next episode's pre-pass behind the current episode.
    python tools/exp_background_prepass.py [books_B]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rl_markets_amd import abi, engine

BB = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
p = engine.default_params()
p.depth, p.max_trades, p.algo, p.theta_mode = 10, 2, abi.ALGO_QLAMBDA, abi.THETA_SHARED
g = engine.default_gen_params()
g.n_events = 64 + 2048
A = engine.Engine(p, 65536)
A.gen_events(g)
A.reset()
A.td_step(40)
A.sync()
B = engine.Engine(p, BB)
B.gen_events(g)
B.reset()
B.sync()


def rate(n=300):
    A.sync()
    t0 = time.perf_counter()
    A.td_step(n)
    A.sync()
    return (time.perf_counter() - t0) / n * 1e3


def reset_ms(n=5):
    B.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        B.reset()
    B.sync()
    return (time.perf_counter() - t0) / n * 1e3


print("A alone: %.4f ms per step" % rate())
print("B alone: %.2f ms per lob_reset (%d books)" % (reset_ms(), BB))
stop = False
resets = [0]


def loop():
    while not stop:
        B.reset()
        resets[0] += 1


th = threading.Thread(target=loop)
th.start()
time.sleep(0.05)
r0 = resets[0]
t0 = time.perf_counter()
ms = rate(600)
dt = time.perf_counter() - t0
n = resets[0] - r0
stop = True
th.join()
print("A beside B's back-to-back resets: %.4f ms per step; B: %d resets in %.3f s = %.2f ms each" % (ms, n, dt, dt / max(n, 1) * 1e3))
print("A alone again: %.4f ms per step" % rate())
