"""Multi-GPU weight exchange pieces on ONE GPU (a 1-GPU box cannot host two RCCL ranks: RCCL
refuses duplicate devices): two engines (= two book shards) on the same device exchanging their
delta buffers exactly as lob_theta_allreduce does, the sum over ranks formed on the host; the
RCCL all-reduce itself on a one-rank communicator, in place on the engine's buffer and stream
(rl_markets_amd.comm.RcclComm -> liblob_comm.so).  Checked against the oracle running the same
two-shard schedule.  No torch anywhere on this path."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from rl_markets_amd import abi, engine
from rl_markets_amd.parallel import EngineBackend, ShardedLearner, shard_books
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return _hip


def d2h(ptr, n):
    out = np.empty(n, np.float64)
    assert hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), n * 8, 2) == 0
    return out


def h2d(ptr, arr):
    a = np.ascontiguousarray(arr, np.float64)
    assert hip().hipMemcpy(C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.size * 8, 1) == 0


def make_shards(algo, total=12, world=2, M=1 << 16, n_events=200):
    g = engine.default_gen_params()
    g.n_events = n_events
    engs, orcs = [], []
    for r in range(world):
        first, n = shard_books(total, world, r)
        p = engine.default_params()
        p.memory_size = M
        p.algo = algo
        p.book_id_offset = first
        rec = engine.gen_stream_host(g, p.depth, p.max_trades, first, n)
        e = engine.Engine(p, n)
        e.load_events(rec)
        e.reset()
        e.delta_init()
        o = ol.Oracle(p, rec)
        o.reset()
        engs.append(e)
        orcs.append(o)
    return engs, orcs


def host_allreduce(engs):
    """What lob_theta_allreduce does on every rank, the sum over ranks formed on the host."""
    bufs = [e.delta_begin() for e in engs]
    tot = sum(d2h(ptr, n) for ptr, n in bufs)
    for (ptr, n), e in zip(bufs, engs):
        h2d(ptr, tot)
        e.delta_apply()


@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA, abi.ALGO_DOUBLE_Q])
def test_delta_exchange_two_shards_one_gpu(algo):
    M, steps, sync = 1 << 16, 48, 16
    engs, orcs = make_shards(algo, M=M)
    nv = 2 if algo == abi.ALGO_DOUBLE_Q else 1
    vecs = [lambda o: o.theta(0)] + ([lambda o: o.theta_b(0)] if nv == 2 else [])
    osync = [np.zeros(M) for _ in vecs]
    for s0 in range(0, steps, sync):
        for e, o in zip(engs, orcs):
            e.td_step(sync)
            o.td_step(sync)
        assert engs[0].delta_begin()[1] == nv * M
        host_allreduce(engs)
        for v, get in enumerate(vecs):
            ototal = sum(get(o) - osync[v] for o in orcs)
            for o in orcs:
                get(o)[:] = osync[v] + ototal
            osync[v] = get(orcs[0]).copy()
    for v, get in enumerate(vecs):
        t0, t1 = engs[0].theta(v), engs[1].theta(v)
        np.testing.assert_array_equal(t0, t1)
        np.testing.assert_allclose(t0, get(orcs[0]), rtol=1e-9, atol=1e-15)
        assert np.count_nonzero(t0) > 100
    for e in engs:
        e.close()


def d2h_u32(ptr, n):
    out = np.empty(n, np.uint32)
    assert hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), n * 4, 2) == 0
    return out


def host_sparse_allreduce(engs):
    """What lob_theta_allreduce does on the sparse path, the two collectives formed on the host: all-gather of the ranks'
    written-weights maps, union + pack on every rank, all-reduce (SUM) of the packed deltas, scatter."""
    world = len(engs)
    maps = [e.delta_sparse_maps(world) for e in engs]
    own = [d2h_u32(o, w) for o, _, w in maps]
    for (_, gather, words), e in zip(maps, engs):
        for r in range(world):   # ncclAllGather: rank r's map lands in slot r on every rank
            a = np.ascontiguousarray(own[r])
            assert hip().hipMemcpy(C.c_void_p(gather + r * words * 4), a.ctypes.data_as(C.c_void_p), words * 4, 1) == 0
    packed = [e.delta_sparse_pack(world) for e in engs]
    counts = {n for _, n in packed}
    assert len(counts) == 1, "every rank must arrive at the same compact layout"
    n = counts.pop()
    tot = sum(d2h(ptr, n) for ptr, _ in packed)
    for (ptr, _), e in zip(packed, engs):
        h2d(ptr, tot)
        e.delta_sparse_apply()
    return n


@pytest.mark.parametrize("algo", [abi.ALGO_SARSA, abi.ALGO_QLAMBDA])
def test_sparse_exchange_equals_dense(algo):
    """The sparse exchange (written-weights maps gathered, union packed, a few thousand doubles reduced) against the dense
    one (memory_size doubles) on two pairs of shards stepping the same books: the weights must agree after every exchange
    (to the order of a shard's own atomic additions; the two replicas of a pair bit for bit), keep agreeing while the shards go on learning from them
    -- the maps the sparse path leaves mark a superset of the dense path's, which may not change a single Q value --, and
    follow the oracle's two-shard schedule."""
    M, steps, sync = 1 << 16, 64, 8
    sparse, orcs = make_shards(algo, M=M)
    dense, orcs2 = make_shards(algo, M=M)
    for o in orcs2:
        o.close()
    assert all(e.delta_sparse_supported() for e in sparse)
    osync = np.zeros(M)
    for rnd, s0 in enumerate(range(0, steps, sync)):
        mid = rnd % 2 == 1   # every other exchange sits INSIDE the sync step (lob_td_step_begin / _end), as ShardedLearner places it
        for x in sparse + dense + orcs:
            x.td_step(sync - 1 if mid else sync)
            if mid:
                x.td_step_begin()
        n = host_sparse_allreduce(sparse)
        host_allreduce(dense)
        assert 0 < n <= M   # (the union's size, or -- exchanges without a host synchronisation -- the fixed count that holds it)
        ototal = sum(o.theta(0) - osync for o in orcs)
        for o in orcs:
            o.theta(0)[:] = osync + ototal
        osync = orcs[0].theta(0).copy()
        if not mid:
            np.testing.assert_array_equal(sparse[0].theta(), sparse[1].theta())
        for x in sparse + dense + orcs:
            if mid:
                x.td_step_end()
        for es, ed, o in zip(sparse, dense, orcs):
            # (two runs of the same shard agree up to the order of their f64 atomic additions)
            np.testing.assert_allclose(es.theta(), ed.theta(), rtol=1e-12, atol=1e-18, err_msg="after the exchange at step %d" % (s0 + sync))
            np.testing.assert_array_equal(es.last_actions(), ed.last_actions())
            np.testing.assert_array_equal(es.last_actions(), o.recs()["action"])
            np.testing.assert_allclose(es.last_td(), o.recs()["td"], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(es.theta(), o.theta(0), rtol=1e-9, atol=1e-15)
    assert np.count_nonzero(sparse[0].theta()) > 100
    # the first exchange asks the device for the union's size; the later ones go by the size the exchange before them left in
    # pinned memory and exchange a fixed count without a host synchronisation (lob_delta_sparse_pack)
    st = exchange_stats(sparse[0])
    print("exchange:", st)
    assert st["synced"] >= 2 and st["no_sync"] >= 2 and st["overflows"] == 0, st
    for e in sparse + dense:
        e.close()
    for o in orcs:
        o.close()


def exchange_stats(eng):
    out = (C.c_int64 * 5)()
    fn = eng.lib.lob_debug_exchange
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]
    assert fn(eng.h, out) == 0
    return {"no_sync": int(out[0]), "synced": int(out[1]), "overflows": int(out[2]), "fixed_count": int(out[3]), "last_union": int(out[4])}


def test_a_union_that_outgrows_the_fixed_count_loses_nothing(monkeypatch):
    """The sparse exchange without a host synchronisation exchanges a FIXED number of doubles, chosen from the union sizes of the
    two exchanges before (the last + twice the last increase).  Here a few exchanges one learner step apart settle that count
    (LOB_SPX_COUNT=64 so that it starts small); then the shards learn for sixty steps and the union grows many-fold: the next
    exchange carries only the entries that fit, the rest keep their theta - theta_sync and travel with the one after it -- after
    which both shards hold exactly theta_sync + the sum of the two shards' deltas, as if nothing had been late."""
    monkeypatch.setenv("LOB_SPX_COUNT", "64")
    M = 1 << 16
    engs, orcs = make_shards(abi.ALGO_QLAMBDA, M=M)
    for o in orcs:
        o.close()
    for _ in range(10):
        for e in engs:
            e.td_step(1)
        host_sparse_allreduce(engs)
        if exchange_stats(engs[0])["no_sync"] >= 1:
            break
    st0 = exchange_stats(engs[0])
    assert st0["no_sync"] >= 1 and st0["overflows"] == 0, st0
    base = engs[0].theta()
    np.testing.assert_array_equal(base, engs[1].theta())
    for e in engs:
        e.td_step(60)
    before = [e.theta() for e in engs]
    n2 = host_sparse_allreduce(engs)          # no synchronisation: the fixed count, smaller than the union by now
    st = exchange_stats(engs[0])
    assert st["no_sync"] == st0["no_sync"] + 1 and n2 == st["fixed_count"], (n2, st0, st)
    assert not np.array_equal(engs[0].theta(), engs[1].theta())      # (the surplus has not travelled yet)
    n3 = host_sparse_allreduce(engs)          # the exchange after it: sees the union's true size, takes the exact count
    st = exchange_stats(engs[0])
    assert st["overflows"] == 1 and st["synced"] == st0["synced"] + 1 and n3 > n2, (n2, n3, st)
    want = base + (before[0] - base) + (before[1] - base)
    for e in engs:
        np.testing.assert_allclose(e.theta(), want, rtol=1e-12, atol=1e-18)
    np.testing.assert_array_equal(engs[0].theta(), engs[1].theta())
    for e in engs:
        e.close()


def test_checkpoint_loaded_after_delta_init_is_not_scaled_by_world_size():
    """lob_theta_set after lob_delta_init (ShardedLearner's constructor ran, then a checkpoint is
    loaded on every rank): the loaded weights are the new common base.  Before the fix every rank
    contributed (loaded - sync) and theta came out as sync + world * (loaded - sync)."""
    engs, orcs = make_shards(abi.ALGO_SARSA)
    for e in engs:
        e.td_step(8)
    host_allreduce(engs)
    loaded = np.random.default_rng(5).standard_normal(engs[0].M) * 1e-3
    for e in engs:
        e.set_theta(loaded)
    host_allreduce(engs)          # nobody stepped: the sum of the deltas must be zero
    for e in engs:
        np.testing.assert_array_equal(e.theta(), loaded)
    for e in engs:
        e.td_step(4)
    host_allreduce(engs)
    np.testing.assert_array_equal(engs[0].theta(), engs[1].theta())
    assert np.abs(engs[0].theta() - loaded).max() < 1.0   # a few small TD updates on top of the checkpoint, not 2 x checkpoint
    for e in engs:
        e.close()


def test_rccl_allreduce_in_place_on_the_engine_buffer(tmp_path):
    """The product's exchange: lob_theta_allreduce = delta kernel -> RCCL all-reduce (f64, SUM) in place
    on the engine's own buffer, on the engine's stream -> apply kernel.  One rank: the sum over ranks is
    the rank's own delta, so the weights must come out bit-identical, and stay in step with the oracle."""
    from rl_markets_amd.comm import MAX, SUM, RcclComm
    comm = RcclComm(str(tmp_path / "rdzv"), 0, 1, 0)
    assert not (tmp_path / "rdzv").exists()          # rank 0 removes the token once everybody has joined
    engs, orcs = make_shards(abi.ALGO_QLAMBDA, total=16, world=1)
    eng, orc = engs[0], orcs[0]
    learner = ShardedLearner(EngineBackend(eng), comm, sync_every=8)
    learner.run(40)
    orc.td_step(40)
    eng.sync()
    assert learner.n_syncs == 5
    np.testing.assert_allclose(eng.theta(), orc.theta(), rtol=1e-9, atol=1e-15)
    before = eng.theta()
    comm.sync_weights(EngineBackend(eng))
    eng.sync()
    np.testing.assert_array_equal(eng.theta(), before)
    assert comm.reduce([1.5, -2.0], MAX) == [1.5, -2.0] and comm.reduce([3.0], SUM) == [3.0]
    comm.barrier()
    comm.close()
    eng.close()


def test_bench_multi_gpu_code_path_on_one_rank():
    """bench.py with LOB_FORCE_DIST=1: the N > 1 code path (RCCL communicator, delta kernels, in-place
    all-reduce every 64 steps, max/sum reductions of the timings over ranks) on a single rank."""
    env = dict(os.environ, LOB_FORCE_DIST="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--books", "2048", "--steps", "130", "--warmup", "10",
                          "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["sync_every"] == 64 and d["value"] > 0
    ks = d["roofline"]["all_kernels_avg_ms"]
    assert "delta_begin_kernel" in ks and "delta_apply_kernel" in ks
    assert "torch" not in out.stderr.lower()


def test_lob_run_multi_gpu_path_on_one_rank(tmp_path):
    """The C++ driver's --gpus path (fork per GPU, file rendezvous, Learner::_step exchanging through
    lob_theta_allreduce, global live-book count deciding the end of the episode) with one rank."""
    exe = os.path.join(ROOT, "rl_markets_amd", "host", "lob_run")
    cfg = os.path.join(ROOT, "config", "engine.yaml")
    th = [str(tmp_path / "a.bin"), str(tmp_path / "b.bin")]
    outs = []
    for force, path in (("1", th[0]), ("0", th[1])):
        env = dict(os.environ, LOB_FORCE_DIST=force)
        out = subprocess.run([exe, "-c", cfg, "-n", "64", "-e", "1", "--events", "400", "--sync-every", "16", "--theta", path],
                             capture_output=True, text=True, env=env)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        # (RCCL prints a version banner on stdout when the communicator comes up)
        outs.append([l for l in out.stdout.splitlines() if l.startswith("episode,") or l[:1].isdigit()])
    # same books, same weights: one rank's exchange is the identity (up to the float order of theta_sync + delta)
    assert len(outs[0]) == 2 and outs[0] == outs[1]
    a, b = np.fromfile(th[0]), np.fromfile(th[1])
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-15)
    assert np.count_nonzero(a) > 100
