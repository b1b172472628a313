"""How many weights get written at the headline configuration?  (sizes the coarse 'written' filter)"""
import sys
sys.path.insert(0, '.')
import numpy as np
from rl_markets_amd import abi, engine
p = engine.default_params(); p.depth = 10; p.algo = abi.ALGO_QLAMBDA
g = engine.default_gen_params(); g.n_events = 2112
eng = engine.Engine(p, 65536); eng.gen_events(g); eng.reset()
done = 0
for chunk in (20, 200, 800, 1000):
    eng.td_step(chunk); eng.sync(); done += chunk
    th = eng.theta(0)
    nzi = np.flatnonzero(th)
    out = {'steps': done, 'written': int(nzi.size)}
    for gsz in (32, 64, 128, 256, 512):
        nb = (th.size + gsz - 1) // gsz
        out['dirty_%d' % gsz] = round(np.unique(nzi // gsz).size / nb, 4)
    print(out, flush=True)
eng.close()
