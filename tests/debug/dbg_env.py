"""Debugging aid (test infrastructure: it drives the oracle beside the engine); run from the repository root."""
import sys
sys.path.insert(0, '.')
import numpy as np
from tests.test_gpu_fuzz import random_case
from rl_markets_amd import abi, engine
from tests import oracle_lib as ol
from tests.parity import dumps_to_np
seed = int(sys.argv[1])
p, g, B = random_case(5000 + seed)
g.n_events = 150
rec = engine.gen_stream_host(g, p.depth, p.max_trades, p.book_id_offset, B)
eng = engine.Engine(p, B); eng.load_events(rec); orc = ol.Oracle(p, rec)
eng.reset(); orc.reset()
rng = np.random.default_rng(seed)
for step in range(14):
    if step % 7 == 6:
        eng.eval_step(1); orc.eval_step(1)
        r = orc.recs()
        print('eval step', step, 'engine actions', eng.last_actions(), 'oracle', r['action'], 'rng', eng.rng_counters(), r['rng_ctr'])
        print(' engine state', eng.learner_state()[0], ' oracle vars', r['vars'][0][:eng.V])
        eb = dumps_to_np(eng.get_books()); ob = r['book']
        print(' cursor', eb['cursor'], ob['cursor'], 'terminal', eng.get_terminal())
    else:
        acts = rng.integers(0, 9, size=B).astype(np.int32)
        eng.step(acts); orc.env_step(acts)
