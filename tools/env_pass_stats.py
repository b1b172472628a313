"""What the event passes of the env step do to the agent's two orders, counted on the CPU oracle (debug counters of
oracle/lob_oracle.cpp) for a sample of books of the headline configuration: how many passes / steps are QUIET (no trade at
or through an order price, no cancellation ahead of the order, both order levels still in the applied row, no adverse
selection) -- the share a light env pass with a hand-back list could serve.

    python tools/env_pass_stats.py [--books 128] [--steps 400] [--depth 10] [--eps 0.8]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_markets_amd import abi, engine  # noqa: E402  (host-side generator only)
from tests import oracle_lib as ol     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--books", type=int, default=128)
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--depth", type=int, default=10)
ap.add_argument("--eps", type=float, default=0.8)
ap.add_argument("--alpha", type=float, default=0.001)
a = ap.parse_args()

p = engine.default_params()
p.depth, p.max_trades = a.depth, 2
p.algo, p.theta_mode = abi.ALGO_QLAMBDA, abi.THETA_SHARED
p.memory_size = 1 << 22
p.epsilon, p.alpha = a.eps, a.alpha
g = engine.default_gen_params()          # the bench's streams
rec = engine.gen_stream_host(g, a.depth, 2, 0, a.books)
o = ol.Oracle(p, rec)
lib = ol.load()
lib.oracle_debug_pass_stats.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_longlong * 16)()
o.reset()
lib.oracle_debug_pass_stats(buf, 1)      # drop what Initialise's warm-up counted
o.td_step(a.steps)
lib.oracle_debug_pass_stats(buf, 1)
s = list(buf)
passes, steps = s[0], s[11]
names = ["trade at / through an order price", "fill", "order executed", "cancellation ahead of the order", "order level gone from the row",
         "order level not in the previous row", "more volume behind (simple)", "adverse selection", "executed order erased"]
print("books %d  steps %d  passes %d  (%.2f passes per step, %.0f %% of steps take more than one)" % (
    a.books, steps, passes, passes / max(1, steps), 100.0 * s[14] / max(1, steps)))
for i, n in enumerate(names):
    print("  %-40s %6.2f %% of passes" % (n, 100.0 * s[1 + i] / max(1, passes)))
print("quiet passes (nothing, or only more volume behind): %.1f %%" % (100.0 * s[10] / max(1, passes)))
print("steps whose passes are all quiet:                    %.1f %%" % (100.0 * s[12] / max(1, steps)))
print("steps without fill / execution / adverse selection / vanished level: %.1f %%" % (100.0 * s[13] / max(1, steps)))
for w in (16, 32, 64):
    q = s[12] / max(1, steps)
    print("  a wave of %2d books is all-quiet in a step with probability %.3g" % (w, q ** w))
