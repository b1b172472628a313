"""Wall time of each of an episode's first learner steps (host clock around td_step(1) + sync: ~20 us of synchronisation in every
figure, the shape is what matters), with the path statistics of the step: what the driver's `--steps 20 --warmup 5` leg times
that the 200-step leg does not.
    python tools/exp_first_steps.py [n_steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rl_markets_amd import abi, engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
p = engine.default_params()
p.depth, p.max_trades, p.algo, p.theta_mode = 10, 2, abi.ALGO_QLAMBDA, abi.THETA_SHARED
g = engine.default_gen_params()
g.n_events = 64 + 2048
eng = engine.Engine(p, 65536)
eng.gen_events(g)
for episode in range(2):
    eng.reset()
    eng.sync()
    prev = eng.path_stats()
    out = []
    for i in range(n):
        t0 = time.perf_counter()
        eng.td_step(1)
        eng.sync()
        dt = (time.perf_counter() - t0) * 1e3
        ps = eng.path_stats()
        out.append("%2d %.3f ms  slots %d new-in-step %d  handed_on %d  act_full %d rest %d" % (
            i, dt, ps[2], ps[5], ps[0] - prev[0], ps[6] - prev[6], ps[7] - prev[7]))
        prev = ps
    print("episode %d" % episode)
    print("\n".join(out))
    eng.clear_inventory()
    eng.handle_terminal()
