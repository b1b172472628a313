"""SARSA(lambda): how much would combining the theta updates of neighbouring books save?
Ratio (trace entries) / (distinct weights) over blocks of consecutive books."""
import sys
sys.path.insert(0, '.')
import numpy as np
from rl_markets_amd import abi, engine
p = engine.default_params(); p.depth = 10; p.algo = abi.ALGO_SARSA
g = engine.default_gen_params(); g.n_events = 1500
eng = engine.Engine(p, 65536); eng.gen_events(g); eng.reset()
eng.td_step(400); eng.sync()
N = 512
tr = [eng.traces(b)[0] for b in range(N)]
print('mean traces per book', np.mean([len(t) for t in tr]))
for blk in (4, 16, 32, 64, 128, 256, 512):
    tot = dis = 0
    for s in range(0, N, blk):
        cat = np.concatenate(tr[s:s + blk])
        tot += cat.size; dis += np.unique(cat).size
    print('books per block', blk, 'entries/distinct', round(tot / dis, 2), 'distinct per block', dis // (N // blk))
eng.close()
