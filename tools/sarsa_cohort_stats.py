"""How many additions SARSA(lambda)'s update would need if the trace generations of one creation step were kept together,
sorted by identity: counted on the CPU oracle for a sample of books of the headline configuration.

Today accumulate_kernel makes one f64 atomic addition per live (book, generation); all books that hold the SAME
(generation, alive tiles) add into one slot.  Generations of equal age were created in the same step; sorted by identity
at creation, equal slots would be adjacent in a wave and could be summed there first.  This prints, per step, the live
(book, generation) pairs, the distinct (tiles) sets among them (= slots), and the sum over ages of the distinct sets of
that age (= additions after perfect in-cohort combining).

    python tools/sarsa_cohort_stats.py [--books 4096] [--steps 80]
"""
import argparse
import os
import sys
from collections import Counter, defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ORACLE_THREADS", str(max(1, min(32, (os.cpu_count() or 1)))))
from rl_markets_amd import abi, engine  # noqa: E402  (host-side generator only)
from tests import oracle_lib as ol     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--books", type=int, default=4096)
ap.add_argument("--steps", type=int, default=80)
a = ap.parse_args()

p = engine.default_params()               # the bench's configuration: D = 10, M = 20 M, eps 0.8 ...
p.depth, p.max_trades = 10, 2
p.algo, p.theta_mode = abi.ALGO_SARSA, abi.THETA_SHARED
g = engine.default_gen_params()
rec = engine.gen_stream_host(g, 10, 2, 0, a.books)
o = ol.Oracle(p, rec)
o.reset()
for step in range(a.steps):
    o.td_step(1)
    if step < a.steps - 3:
        continue
    pairs = 0
    slots = set()
    per_age = defaultdict(set)
    for b in range(a.books):
        idx, e = o.traces(b)
        by_e = defaultdict(list)
        for i, x in zip(idx.tolist(), e.tolist()):
            by_e[x].append(i)             # one eligibility value per age
        for x, tiles in by_e.items():
            key = tuple(sorted(tiles))
            pairs += 1
            slots.add(key)
            per_age[x].add(key)
    cohort = sum(len(v) for v in per_age.values())
    print("step %3d: %7d live (book, generation) pairs = additions today; %6d distinct tile sets (slots); "
          "%6d additions with in-cohort combining (%.1f x fewer); %d ages" % (step, pairs, len(slots), cohort, pairs / max(1, cohort), len(per_age)))
