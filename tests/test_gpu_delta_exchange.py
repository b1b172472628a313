"""Multi-GPU weight exchange pieces on ONE GPU: two engines (= two book shards)
on the same device, the delta buffers handed to torch through
__cuda_array_interface__ exactly as bench.py does for the RCCL all-reduce, the
all-reduce itself emulated by a torch sum.  Checked against the oracle running
the same two-shard schedule.

Runs in a fresh interpreter that imports torch BEFORE liblob_engine.so is
loaded (as bench.py does for N > 1): torch bundles its own libamdhip64 with the
same SONAME as /opt/rocm's, and whichever HIP runtime is loaded first must be
the only one in the process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_delta_exchange_two_shards_one_gpu():
    out = subprocess.run([sys.executable, os.path.abspath(__file__)], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "DELTA-EXCHANGE-OK" in out.stdout


def exchange(torch, algo):
    """Two shards, one device: train, exchange, compare with the oracle doing the same."""
    total, world, steps, sync = 12, 2, 48, 16
    M = 1 << 16
    nv = 2 if algo == abi.ALGO_DOUBLE_Q else 1
    g = engine.default_gen_params()
    g.n_events = 200
    engs, orcs, backs = [], [], []
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
        backs.append(EngineBackend(e, torch, "cuda:0"))
    vecs = [lambda o: o.theta(0)] + ([lambda o: o.theta_b(0)] if nv == 2 else [])
    osync = [np.zeros(M) for _ in vecs]
    for s0 in range(0, steps, sync):
        for e, o in zip(engs, orcs):
            e.td_step(sync)
            o.td_step(sync)
        ts = [b.delta_tensor() for b in backs]
        assert ts[0].dtype == torch.float64 and ts[0].is_cuda and ts[0].numel() == nv * M
        tot = ts[0] + ts[1]          # stands in for all_reduce(SUM)
        for t in ts:
            t.copy_(tot)
        for b in backs:
            b.after_all_reduce()   # staged tensor -> the engine's buffer
        for e in engs:
            e.delta_apply()
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
    return engs, backs


def main():
    import torch
    assert torch.cuda.is_available()
    exchange(torch, abi.ALGO_DOUBLE_Q)
    engs, backs = exchange(torch, abi.ALGO_SARSA)
    # RCCL itself (one rank: the only collective a 1-GPU box can run): f64 all-reduce on the staged
    # tensor and on the engine's own hipMalloc'ed buffer
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    engs[0].td_step(4)
    t = backs[0].delta_tensor()
    want = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(backs[0].view, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    assert torch.equal(t, want) and torch.equal(backs[0].view, want) and int((want != 0).sum()) > 0
    backs[0].after_all_reduce()
    engs[0].delta_apply()
    dist.destroy_process_group()


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401  (first: see module docstring)
    import numpy as np
    from rl_markets_amd import abi, engine
    from rl_markets_amd.parallel import EngineBackend, shard_books
    from tests import oracle_lib as ol
    main()
    print("DELTA-EXCHANGE-OK")


def test_bench_multi_gpu_code_path_on_one_rank():
    """bench.py with LOB_FORCE_DIST=1: the N > 1 code path (torch first, RCCL process group, delta
    kernels, staged all-reduce every 64 steps, max/sum reductions of the timings) on a single rank."""
    import json
    env = dict(os.environ, LOB_FORCE_DIST="1", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--books", "2048", "--steps", "130", "--warmup", "10",
                          "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["sync_every"] == 64 and d["value"] > 0
    ks = d["roofline"]["all_kernels_avg_ms"]
    assert "delta_begin_kernel" in ks and "delta_apply_kernel" in ks
