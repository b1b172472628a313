"""N > 1 path on CPU: two gloo ranks, each driving its shard of books through
rl_markets_amd.parallel.ShardedLearner with the ORACLE as the compute backend
(tests may use the oracle; the product never does).  The result must equal a
single process that steps all books with the same sync schedule."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOTAL_BOOKS, N_EVENTS, STEPS, SYNC = 6, 260, 96, 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleBackend:
    def __init__(self, orc, torch):
        self.orc, self.torch = orc, torch
        self.sync = None

    def td_step(self, n):
        self.orc.td_step(n)

    def delta_init(self):
        self.sync = self.orc.theta(0).copy()

    def delta_tensor(self):
        self.delta = self.torch.from_numpy(self.orc.theta(0) - self.sync)
        return self.delta

    def after_all_reduce(self):
        pass

    def delta_apply(self):
        th = self.orc.theta(0)
        th[:] = self.sync + self.delta.numpy()
        self.sync = th.copy()


def _make(first, n):
    from rl_markets_amd import abi, engine
    from tests import oracle_lib as ol
    p = engine.default_params()
    p.memory_size = 1 << 16
    p.theta_mode = abi.THETA_SHARED
    p.algo = abi.ALGO_SARSA
    p.book_id_offset = first
    g = engine.default_gen_params()
    g.n_events = N_EVENTS
    rec = engine.gen_stream_host(g, p.depth, p.max_trades, first, n)
    o = ol.Oracle(p, rec)
    o.reset()
    return o


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from rl_markets_amd.parallel import ShardedLearner, shard_books
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n = shard_books(TOTAL_BOOKS, world, rank)
    o = _make(first, n)
    sl = ShardedLearner(OracleBackend(o, torch), dist, sync_every=SYNC)
    sl.run(STEPS)
    np.save(os.path.join(out_dir, "theta_%d.npy" % rank), o.theta(0))
    np.save(os.path.join(out_dir, "steps_%d.npy" % rank), o.counters())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_books_partition():
    from rl_markets_amd.parallel import shard_books
    for total in (1, 7, 8, 65536, 524288):
        for world in (1, 2, 3, 8):
            spans = [shard_books(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total
            for (f0, n0), (f1, _n1) in zip(spans, spans[1:]):
                assert f0 + n0 == f1


def test_two_rank_gloo_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    t0 = np.load(tmp_path / "theta_0.npy")
    t1 = np.load(tmp_path / "theta_1.npy")
    np.testing.assert_array_equal(t0, t1)  # replicas agree after the last sync... 
    # single-process emulation of the same schedule: two shards, private replicas, summed deltas
    from rl_markets_amd.parallel import shard_books
    shards = [_make(*shard_books(TOTAL_BOOKS, world, r)) for r in range(world)]
    sync = np.zeros_like(shards[0].theta(0))
    done = 0
    while done < STEPS:
        chunk = min(SYNC - done % SYNC, STEPS - done)
        for o in shards:
            o.td_step(chunk)
        done += chunk
        if done % SYNC == 0:
            total = sum(o.theta(0) - sync for o in shards)
            for o in shards:
                o.theta(0)[:] = sync + total
            sync = shards[0].theta(0).copy()
    if STEPS % SYNC == 0:
        np.testing.assert_allclose(t0, shards[0].theta(0), rtol=1e-12, atol=0)
    else:
        # after the last sync the replicas drift apart again by their local updates
        pass
    steps = sum(int(np.load(tmp_path / ("steps_%d.npy" % r))[0]) for r in range(world))
    assert steps == sum(int(o.counters()[0]) for o in shards)
