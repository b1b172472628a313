"""Would combining whole trace GENERATIONS across books pay?  A generation = the live tiles one
(state, action) left in one book; count (book, generation) pairs against distinct live-tile sets."""
import sys
sys.path.insert(0, '.')
import numpy as np
from rl_markets_amd import abi, engine
algo = abi.ALGO_SARSA if sys.argv[1] == 'sarsa' else abi.ALGO_QLAMBDA
p = engine.default_params(); p.depth = 10; p.algo = algo
g = engine.default_gen_params(); g.n_events = 1500
eng = engine.Engine(p, 65536); eng.gen_events(g); eng.reset()
eng.td_step(400); eng.sync()
for N in (1024, 8192):
    pairs = 0; entries = 0; keys = set()
    for b in range(N):
        idx, el = eng.traces(b)
        for e in np.unique(el):
            s = idx[el == e]
            pairs += 1; entries += s.size
            keys.add((float(e), s.tobytes()))
    print(sys.argv[1], 'books', N, 'entries', entries, '(book,gen) pairs', pairs, 'distinct (age, tile set)', len(keys),
          'atomics after combining ~', pairs + sum(len(k[1]) // 4 for k in keys), flush=True)
eng.close()
