"""More than one real RCCL rank (skipped below two visible GPUs; the same harness runs with ONE rank on a one-GPU box).

Every rank is a process of its own (rl_markets_amd.launch.spawn_ranks, as `bench.py --gpus N` starts them), drives its
shard of books through rl_markets_amd.parallel.ShardedLearner on the HIP engine and exchanges the shared weight vector with
lob_theta_allreduce (include/lob_comm.h: the ranks' written-weights maps all-gathered, the packed deltas of their union
all-reduced over xGMI -- or the whole vector, LOB_DENSE_EXCHANGE=1) every SYNC steps, inside the sync step.  The ranks'
final weights must equal (1e-9) the oracle's run of the same schedule -- N shards, private replicas, deltas summed at the
same points (tests/test_dist_gloo.py runs that schedule on the CPU over gloo) -- and RCCL must really have seen N ranks
(ncclCommCount).  Reference ancestor: the Hogwild threads sharing one rl::Agent, src/main.cpp:196-206."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

BOOKS_PER_RANK, N_EVENTS, STEPS, SYNC, MEM = 1024, 700, 130, 64, 1 << 20


def _params(first, algo):
    from rl_markets_amd import abi, engine
    p = engine.default_params()
    p.depth, p.max_trades = 10, 2
    p.memory_size = MEM
    p.theta_mode = abi.THETA_SHARED
    p.algo = algo
    p.book_id_offset = first
    g = engine.default_gen_params()
    g.n_events = N_EVENTS
    return p, g


def _worker(out_dir, algo):
    """One rank: RANK / LOCAL_RANK / WORLD_SIZE / LOB_RDZV from spawn_ranks."""
    sys.path.insert(0, ROOT)
    from rl_markets_amd import abi, engine, launch
    from rl_markets_amd.comm import RcclComm
    from rl_markets_amd.parallel import EngineBackend, ShardedLearner
    rank, local_rank, world = launch.rank_env()
    p, g = _params(rank * BOOKS_PER_RANK, int(algo))
    rec = engine.gen_stream_host(g, p.depth, p.max_trades, rank * BOOKS_PER_RANK, BOOKS_PER_RANK)
    eng = engine.Engine(p, BOOKS_PER_RANK, device=local_rank)
    eng.load_events(rec)
    comm = RcclComm(launch.rendezvous_path(), rank, world, local_rank)
    eng.reset()
    sl = ShardedLearner(EngineBackend(eng), comm, sync_every=SYNC)
    sl.run(STEPS)
    eng.sync()
    st = comm.exchange_stats()
    np.save(os.path.join(out_dir, "theta_%d.npy" % rank), eng.theta())
    if int(algo) == abi.ALGO_DOUBLE_Q:
        np.save(os.path.join(out_dir, "theta_b_%d.npy" % rank), eng.theta(1))
    json.dump({"stats": st, "counters": [int(v) for v in eng.counters()], "n_syncs": sl.n_syncs}, open(os.path.join(out_dir, "info_%d.json" % rank), "w"))
    comm.barrier()
    comm.close()
    eng.close()
    print("rank %d of %d done" % (rank, world))


def _oracle_schedule(world, algo):
    """The same schedule on the CPU: `world` shards with private replicas, deltas summed inside every SYNC-th step."""
    from rl_markets_amd import engine
    from tests import oracle_lib as ol
    shards = []
    for r in range(world):
        p, g = _params(r * BOOKS_PER_RANK, algo)
        rec = engine.gen_stream_host(g, p.depth, p.max_trades, r * BOOKS_PER_RANK, BOOKS_PER_RANK)
        o = ol.Oracle(p, rec)
        o.reset()
        shards.append(o)
    from rl_markets_amd import abi
    double = algo == abi.ALGO_DOUBLE_Q   # DoubleAgent::theta_b (src/rl/agent.cpp:185-264) travels with theta: [theta | theta_b] in one exchange
    sync = np.zeros_like(shards[0].theta(0))
    sync_b = np.zeros_like(sync)
    done = 0
    while done < STEPS:
        chunk = min(SYNC - done % SYNC, STEPS - done)
        sync_now = (done + chunk) % SYNC == 0
        for o in shards:
            o.td_step(chunk - 1 if sync_now else chunk)
        done += chunk
        if sync_now:
            for o in shards:
                o.td_step_begin()
            total = sum(o.theta(0) - sync for o in shards)
            for o in shards:
                o.theta(0)[:] = sync + total
            sync = shards[0].theta(0).copy()
            if double:
                total_b = sum(o.theta_b(0) - sync_b for o in shards)
                for o in shards:
                    o.theta_b(0)[:] = sync_b + total_b
                sync_b = shards[0].theta_b(0).copy()
            for o in shards:
                o.td_step_end()
    return shards


def _run(tmp_path, world, algo, extra_env=None):
    from rl_markets_amd import launch
    env = dict(os.environ, **(extra_env or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    rc = launch.spawn_ranks([sys.executable, os.path.abspath(__file__), "worker", str(tmp_path), str(algo)], world, timeout=900, env=env)
    assert rc == 0
    shards = _oracle_schedule(world, algo)
    for r in range(world):
        info = json.load(open(tmp_path / ("info_%d.json" % r)))
        assert info["stats"]["rccl_world"] == world          # RCCL's own count of the communicator's ranks
        assert info["stats"]["exchanges"] == STEPS // SYNC == info["n_syncs"]
        assert info["counters"][0] == int(shards[r].counters()[0])
        th = np.load(tmp_path / ("theta_%d.npy" % r))
        assert np.count_nonzero(th) > 1000
        np.testing.assert_allclose(th, shards[r].theta(0), rtol=1e-9, atol=1e-12)
        if os.path.exists(tmp_path / ("theta_b_%d.npy" % r)):
            thb = np.load(tmp_path / ("theta_b_%d.npy" % r))
            assert np.count_nonzero(thb) > 500
            np.testing.assert_allclose(thb, shards[r].theta_b(0), rtol=1e-9, atol=1e-12)
    for o in shards:
        o.close()
    return [json.load(open(tmp_path / ("info_%d.json" % r))) for r in range(world)]


@pytest.mark.parametrize("exchange,algo_name", [("sparse", "q_lambda"), ("dense", "q_lambda"), ("dense", "double_q")])
def test_one_rank_through_the_multi_rank_harness(tmp_path, exchange, algo_name):
    """The harness itself, on any box with a GPU: one rank, a one-rank RCCL communicator, the whole exchange path -- double Q
    (two weight vectors in one exchange, always dense) included, against the oracle schedule's theta AND theta_b."""
    from rl_markets_amd import abi
    algo = {"q_lambda": abi.ALGO_QLAMBDA, "double_q": abi.ALGO_DOUBLE_Q}[algo_name]
    info = _run(tmp_path, 1, algo, {"LOB_DENSE_EXCHANGE": "1"} if exchange == "dense" else None)
    assert info[0]["stats"]["sparse"] == (0 if exchange == "dense" else STEPS // SYNC)


@pytest.mark.parametrize("exchange", ["sparse", "dense"])
@pytest.mark.parametrize("algo_name", ["q_lambda", "sarsa", "double_q"])
def test_two_ranks_over_rccl(tmp_path, exchange, algo_name):
    """Two GPUs, two processes, RCCL over xGMI: the first time lob_theta_allreduce runs with world > 1 is here."""
    from rl_markets_amd import abi
    from tests.conftest import gpu_count
    if gpu_count() < 2:
        pytest.skip("needs two visible GPUs")
    algo = {"q_lambda": abi.ALGO_QLAMBDA, "sarsa": abi.ALGO_SARSA, "double_q": abi.ALGO_DOUBLE_Q}[algo_name]
    # (double Q carries two vectors and has no sparse form: the ranks agree on the dense exchange by themselves, lob_comm.cpp)
    dense = exchange == "dense" or algo == abi.ALGO_DOUBLE_Q
    info = _run(tmp_path, 2, algo, {"LOB_DENSE_EXCHANGE": "1"} if exchange == "dense" else None)
    assert info[0]["stats"]["sparse"] == (0 if dense else STEPS // SYNC)
    assert info[0]["stats"]["bytes_per_exchange"] == info[1]["stats"]["bytes_per_exchange"] > 0


def test_bench_on_two_gpus_reports_two_ranks():
    """`bench.py --gpus 2` end to end: two ranks, RCCL world 2, two exchanges in 130 steps."""
    from tests.conftest import gpu_count
    if gpu_count() < 2:
        pytest.skip("needs two visible GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--books", "4096", "--steps", "130", "--warmup", "0",
                          "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["exchange"]["rccl_world"] == 2
    assert line["config"]["exchange"]["exchanges"] == 2
    assert line["config"]["env_steps"] > 2 * 4096 * 100


if __name__ == "__main__" and len(sys.argv) == 4 and sys.argv[1] == "worker":
    _worker(sys.argv[2], sys.argv[3])
