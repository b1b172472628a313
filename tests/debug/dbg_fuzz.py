"""Debugging aid (test infrastructure: it drives the oracle beside the engine); run from the repository root."""
import sys
sys.path.insert(0, '.')
import numpy as np
from tests.test_gpu_fuzz import random_case
from rl_markets_amd import abi, engine
from tests import oracle_lib as ol
seed = int(sys.argv[1]); upto = int(sys.argv[2])
p, g, B = random_case(1000 + seed)
rec = engine.gen_stream_host(g, p.depth, p.max_trades, p.book_id_offset, B)
eng = engine.Engine(p, B); eng.load_events(rec); orc = ol.Oracle(p, rec)
eng.reset(); orc.reset()
print('algo', p.algo, 'M', p.memory_size, 'V', p.n_vars, 'eps', p.epsilon, 'gl', p.gamma * p.lambda_)
for step in range(upto + 1):
    eng.td_step(1); orc.td_step(1)
    for b in range(B):
        ei, ee = eng.traces(b); oi, oe = orc.traces(b)
        se, so = dict(zip(ei.tolist(), ee.tolist())), dict(zip(oi.tolist(), oe.tolist()))
        if se != so:
            print('step', step, 'book', b, 'action', eng.last_actions()[b], orc.recs()['action'][b], 'n', len(se), len(so))
            print('  only engine', {k: v for k, v in se.items() if k not in so})
            print('  only oracle', {k: v for k, v in so.items() if k not in se})
            print('  differ', {k: (se[k], so[k]) for k in se if k in so and se[k] != so[k]})
            v = eng.learner_state()[b]
            print('  state', v)
            f = eng.features(np.array([v]))[0]   # [9][96]
            for k in list({k for k in se if k not in so} | {k for k in so if k not in se}):
                print('  idx', k, 'in current-state group-0 tiles of actions', [a for a in range(9) if k in f[a, :32].tolist()])
            sys.exit(0)
print('no difference')
