#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched LOB + TD(lambda) hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: every live book of every
rank performs one Learner::_step equivalent (act -> performAction -> features ->
TD update).  Workload at N=1 = BASELINE.json configs[2] (C3, the configuration
the metric is quoted on): 65 536 parallel synthetic 10-level books, Q(lambda)
with eligibility traces, memory_size 20 M, on one MI355X.  For N>1 every rank
owns its own 65 536-book shard (weak scaling, C4) and the shared weight vector
is exchanged by an RCCL all-reduce of delta-theta every SYNC_EVERY steps.

Prints ONE JSON line (rank 0).  `value` counts env-steps actually performed
(device counter), inputs are generated in HBM before the timed region.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SYNC_EVERY = 64
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# Algorithmic bytes per book per launch (DESIGN.md "Kernels and rooflines";
# SURVEY.md §8d terms regrouped per kernel; n_live = live traces, measured).
def algorithmic_bytes(kernel, depth, trades, n_vars, n_live, events_per_step):
    rec = 4 * ((2 + 4 * depth + 2 * trades + 3) // 4 * 4)
    hdr = 64                          # learner header, one 64-byte record per book
    if kernel == "act_kernel":        # header + 3 state slots in, 9*96 weights (f64), Q(last,.) + header out
        return hdr + 192 + 9 * 96 * 8 + 9 * 8 + 32
    if kernel == "learn_kernel":      # header + slots + Q(last,.) in, 9*96 weights, trace index list r/w, header out
        return hdr + 192 + 9 * 8 + 9 * 96 * 8 + n_live * 4 + 32 * 4 + 26 * 4 + 32
    if kernel == "update_kernel":     # per live trace: index (4) + theta read-modify-write (8 + 8); header + masks
        return hdr + n_live * (4 + 8 + 8) + 26 * 4
    if kernel in ("accumulate_kernel", "apply_kernel"):
        # the combined update: per live trace the same index + theta read-modify-write as update_kernel;
        # summing per distinct generation first only changes how many atomics reach theta
        return (hdr + n_live * (4 + 8 + 8) + 26 * 4) / 2.0
    if kernel == "env_kernel":        # per event: track entry (96) + trade slots + the two snapshots' levels at the order
        per_event = 96 + 2 * trades * 4 + 2 * 2 * depth * 8
        return 2 * 232 + hdr + events_per_step * per_event + 2 * 96 + 3 * 4 * n_vars   # agent scalars r/w + quotes + vars out
    return 0


def hip_device_sync():
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipDeviceSynchronize()
    if rc != 0:
        raise RuntimeError("hipDeviceSynchronize failed: %d" % rc)


def cpu_baseline():
    """Time the UNMODIFIED reference (oracle/_ref/ref_harness, built from the
    reference's own sources) on a bounded sample of the workload on this
    host's cores: config C1 -- one 5-level book (the reference has no 10-level
    book), Q(lambda), memory_size 20 M, one thread."""
    from rl_markets_amd import engine
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(harness):
        return None
    g = engine.default_gen_params()
    g.n_events = 61200      # one synthetic day at 500 ms from 08:00
    g.t0_ms = 8 * 3600000   # the reference skips to open + 30 min itself
    rec = engine.gen_stream_host(g, 5, 2, 0, 1)
    episodes = 5
    with tempfile.TemporaryDirectory() as td:
        sp = os.path.join(td, "s.bin")
        rec.tofile(sp)
        out = subprocess.run([harness, "learner", "--stream", sp, "--events", str(g.n_events), "--book", "0",
                              "--algo", "q_learn", "--mem", "20000000", "--episodes", str(episodes),
                              "--tmp", os.path.join(td, "h")], capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            return {"error": out.stderr[-300:]}
        info = json.loads(out.stdout.strip().splitlines()[-1])
        best, steps = info["best_sec"], info["steps_per_episode"]
        if not best:
            return None
        # whole-host figure: one independent reference process per host core, 3 episodes
        # each, all started together; value = total steps / slowest process' wall time
        ncores = min(os.cpu_count() or 1, 64)
        host = None
        if ncores > 1:
            t0 = time.perf_counter()
            procs = [subprocess.Popen([harness, "learner", "--stream", sp, "--events", str(g.n_events), "--book", "0",
                                       "--algo", "q_learn", "--mem", "20000000", "--episodes", "3",
                                       "--tmp", os.path.join(td, "h%d" % i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                     for i in range(ncores)]
            done = 0
            for pr in procs:
                try:
                    o, _ = pr.communicate(timeout=300)
                    if pr.returncode == 0:
                        done += 3 * json.loads(o.strip().splitlines()[-1])["steps_per_episode"]
                except subprocess.TimeoutExpired:
                    pr.kill()
            wall = time.perf_counter() - t0
            if done:
                host = {"value": done / wall, "cores": ncores, "note": "N independent single-thread reference processes, wall time incl. process start and CSV writing"}
    return {"value": steps / best, "unit": "env-steps/s", "cores": 1, "kind": "reference", "whole_host": host,
            "sample": "C1 slice: 1 book, 5-level (reference maximum), %d-event synthetic day, Q(lambda) rl::QLearn, "
                      "memory_size 20M, Learner::RunEpisode incl. CSV parsing, best of %d episodes (%d steps each)"
                      % (g.n_events, episodes, steps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--books", type=int, default=65536, help="books per GPU")
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--algo", default="q_lambda", choices=["q_lambda", "sarsa", "double_q"])
    ap.add_argument("--memory-size", type=int, default=20000000)
    ap.add_argument("--events", type=int, default=0, help="events per book (0 = 64 warm-up + 2048)")
    ap.add_argument("--replay", type=int, default=0, metavar="N_TOTAL",
                    help="config 5 instead of the headline: every book replays ONE recorded stream of N_TOTAL events "
                         "from its own phase (lob_load_events_shared); not the headline workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    dist = torch = None
    force_dist = os.environ.get("LOB_FORCE_DIST") == "1"   # run the N > 1 code path with one rank (1-GPU boxes)
    if force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29543")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        # torch first: it brings its own HIP runtime and must be the one liblob_engine.so binds to
        import torch
        import torch.distributed as dist
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "rl_markets_amd", "csrc", "liblob_engine.so")):
        ge.build()
    from rl_markets_amd import abi, engine

    p = engine.default_params()
    p.depth, p.max_trades = args.depth, 2
    p.algo = {"q_lambda": abi.ALGO_QLAMBDA, "sarsa": abi.ALGO_SARSA, "double_q": abi.ALGO_DOUBLE_Q}[args.algo]
    p.theta_mode = abi.THETA_SHARED
    p.memory_size = args.memory_size
    p.book_id_offset = rank * args.books
    g = engine.default_gen_params()
    g.n_events = args.events if args.events else 64 + 2048
    need = 64 + 6 * (args.steps + args.warmup)
    if g.n_events < 64 + 3 * (args.steps + args.warmup):
        g.n_events = need

    eng = engine.Engine(p, args.books, device=local_rank)
    if args.replay:
        import numpy as np
        g1 = engine.default_gen_params()
        g1.n_events = max(args.replay, g.n_events)
        day = engine.gen_stream_host(g1, args.depth, 2, 0, 1)[0]
        phase = np.random.default_rng(1994 + rank).integers(0, g1.n_events - g.n_events + 1, size=args.books)
        eng.load_events_shared(day, phase, g.n_events)
    else:
        eng.gen_events(g)   # synthetic streams generated directly in HBM (never timed)
    t_reset = time.perf_counter()
    eng.reset()             # Initialise(): includes the once-per-episode market pre-pass over the whole stream
    eng.sync()
    reset_ms = (time.perf_counter() - t_reset) * 1e3

    from rl_markets_amd.parallel import EngineBackend, ShardedLearner
    learner = ShardedLearner(EngineBackend(eng, torch, "cuda:%d" % local_rank), dist, sync_every=SYNC_EVERY, single_rank_sync=force_dist)

    def run(n_steps, first):
        learner.run(n_steps)

    def barrier():
        eng.sync()
        if world > 1 or force_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        else:
            hip_device_sync()

    run(args.warmup, 0)
    barrier()
    c0 = eng.counters()
    if not args.no_kernel_timing:
        eng.kernel_timing(True)
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    barrier()
    t1 = time.perf_counter()
    c1 = eng.counters()
    elapsed = t1 - t0
    steps_done = int(c1[0] - c0[0])
    events_done = int(c1[1] - c0[1])

    ktimes = {}
    if not args.no_kernel_timing:
        for k in ("act_kernel", "env_kernel", "learn_kernel", "update_kernel", "accumulate_kernel", "apply_kernel",
                  "delta_begin_kernel", "delta_apply_kernel"):
            ms, n = eng.kernel_time_ms(k)
            if n:
                ktimes[k] = {"avg_ms": ms, "launches": n}
        eng.kernel_timing(False)
    books = eng.get_books(0, min(args.books, 4096))
    n_live = float(sum(b.n_traces for b in books)) / len(books)

    if world > 1 or force_dist:
        t = torch.tensor([elapsed, float(steps_done), float(events_done)], dtype=torch.float64, device="cuda:%d" % local_rank)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        steps_done = int(tsum[1])
        events_done = int(tsum[2])

    if rank == 0:
        eps = events_done / max(steps_done, 1)
        roofline = None
        if ktimes:
            dom = max((k for k in ktimes if k.endswith("_kernel") and not k.startswith("delta")),
                      key=lambda k: ktimes[k]["avg_ms"])
            per_book = algorithmic_bytes(dom, args.depth, 2, p.n_vars, n_live, eps)
            live_books = steps_done / world / ktimes[dom]["launches"]   # books one launch of that kernel covers
            bytes_per_launch = per_book * live_books
            achieved = bytes_per_launch / (ktimes[dom]["avg_ms"] * 1e-3) / 1e9
            traffic = None
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tf):
                try:
                    traffic = json.load(open(tf)).get(dom, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "algorithmic_bytes_per_book": round(per_book, 1), "books_per_launch": round(live_books, 1),
                        "avg_launch_ms": round(ktimes[dom]["avg_ms"], 4),
                        "all_kernels_avg_ms": {k: round(v["avg_ms"], 4) for k, v in ktimes.items()}}
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only (the contract); N > 1 runs report null
            try:
                cpu = cpu_baseline()
            except Exception as ex:  # the baseline is reported, never required
                cpu = {"error": str(ex)}
        out = {
            "metric": "env-steps/sec (whole node) at 65 536 parallel books",
            "value": steps_done / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (("C5: %d books replaying one recorded %d-level stream from per-book phases, reward pnl_damped, %s, "
                              if args.replay else "C3: %d parallel synthetic %d-level books per GPU, %s with eligibility traces, ") +
                             "tile-coded linear Q (32 tilings x 3 groups x 9 actions), memory_size %d, "
                             "shared theta, synchronous-batch TD") % (args.books, args.depth,
                                                                   {"q_lambda": "Q(lambda)", "sarsa": "SARSA(lambda)", "double_q": "double Q(lambda)"}[args.algo],
                                                                   args.memory_size),
                "books_per_gpu": args.books, "depth": args.depth, "events_per_book": g.n_events,
                "events_per_step": round(eps, 4), "live_traces_per_book": round(n_live, 1),
                "env_steps": steps_done, "reset_ms_per_episode": round(reset_ms, 2), "sync_every": SYNC_EVERY if (world > 1 or force_dist) else None,
                "parallelism": "%d book shard(s), dense RCCL all-reduce of delta-theta" % world if world > 1 else "1 shard",
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        result = json.dumps(out)
    if world > 1 or force_dist:
        try:
            ctypes.CDLL(None).fflush(None)   # every rank: whatever RCCL buffered on stdout goes out before the barrier
        except OSError:
            pass
        sys.stdout.flush()
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if rank == 0:
        # last, after RCCL is torn down (it prints a version banner on stdout): the JSON line is the
        # final line of output
        # (RCCL writes it through C stdio, which is block-buffered on a pipe and would otherwise be
        # flushed at process exit, after this line)
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(result, flush=True)


if __name__ == "__main__":
    main()
