#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched LOB + TD(lambda) hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: every live book of every
rank performs one Learner::_step equivalent (act -> performAction -> features ->
TD update).  Workload at N=1 = BASELINE.json configs[2] (C3, the configuration
the metric is quoted on): 65 536 parallel synthetic 10-level books, Q(lambda)
with eligibility traces, memory_size 20 M, on one MI355X.  For N>1 every rank
owns its own 65 536-book shard (weak scaling, C4) and the shared weight vector
is exchanged by an RCCL all-reduce of delta-theta every SYNC_EVERY steps
(liblob_comm.so, in place on the engine's buffer and stream; no torch in this
process).  Without a launcher `--gpus N` starts the N ranks itself; under
`python -m torch.distributed.run` it takes RANK / LOCAL_RANK / WORLD_SIZE from
the environment.

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
ISSUE_CLOCK_HZ = 2.4e9  # peak engine clock; the issue roof of a kernel = launch time x clock x CUs x 4 SIMDs x 1 instruction / clock
N_SIMDS = 256 * 4

# Algorithmic bytes per book per launch (DESIGN.md "Kernels and rooflines";
# SURVEY.md §8d terms regrouped per kernel; n_live = live traces, measured).
def algorithmic_bytes(kernel, depth, trades, n_vars, n_live, events_per_step):
    rec = 4 * ((2 + 4 * depth + 2 * trades + 3) // 4 * 4)
    hdr = 64                          # learner header, one 64-byte record per book
    if kernel == "trace_kernel":      # UpdateTraces: header + state slots + Q(last,.) in; trace index list r/w, new generation + masks, header out
        return hdr + 192 + 9 * 8 + n_live * 4 + 32 * 4 + 26 * 4 + 32   # (all books: the step's trace work, whichever kernels share it)
    if kernel == "act_kernel":        # header + 3 state slots in, 9*96 weights (f64), Q(last,.) + header out
        return hdr + 192 + 9 * 96 * 8 + 9 * 8 + 32
    if kernel == "learn_kernel":      # Q(s', .) + TD error: header + slots in, 9*96 weights, header out (the traces are trace_kernel's)
        return hdr + 192 + 9 * 96 * 8 + 32
    if kernel == "update_kernel":     # per live trace: index (4) + theta read-modify-write (8 + 8); header + masks
        return hdr + n_live * (4 + 8 + 8) + 26 * 4
    if kernel in ("accumulate_kernel", "apply_kernel"):
        # the combined update: per live trace the same index + theta read-modify-write as update_kernel;
        # summing per distinct generation first only changes how many atomics reach theta
        return (hdr + n_live * (4 + 8 + 8) + 26 * 4) / 2.0
    if kernel == "env_kernel":        # per event: track entry (128: the merged trades ride in it) + the applied row's four level arrays
        per_event = 128 + 2 * 2 * depth * 4
        # agent scalars r/w (248 B) + header + events + the entries of the quotes / the state (2 x 128) + the current snapshot + vars out
        return 2 * 248 + hdr + events_per_step * per_event + 2 * 128 + 2 * 2 * depth * 4 + 3 * 4 * n_vars
    if kernel == "reset_kernel":      # per event of the stream: the record in, the track entry out (events_per_step = events per book here)
        return events_per_step * (rec + 96) + 2 * 232
    return 0


# timer name (lob_kernel_time_ms) -> kernel function(s) launched under it, as rocprofv3 names them
TIMER_KERNELS = {"act_kernel": ("act_light_kernel", "act_fast_kernel", "act_kernel"), "env_kernel": ("env_step_kernel", "env_step16_kernel", "env_kernel"), "trace_kernel": ("trace_lane_kernel", "trace_fast_kernel"),
                 "trace_light_kernel": ("trace_light_kernel",),
                 "learn_kernel": ("learn_q_pair_kernel", "learn_q_lane_kernel", "learn_q_fast_kernel", "learn_kernel"), "act_rest_kernel": (), "learn_rest_kernel": ("learn_q_rest_kernel",),
                 "accumulate_kernel": ("trace_rest_kernel", "accumulate_block_kernel", "accumulate_kernel")}


# ... and the timers that bracket two launches (the lane trace kernel and the wave-per-book kernel behind it): their figures add up
TIMER_SUMS = {"trace_kernel": ("trace_lane_kernel", "trace_fast_kernel"), "accumulate_kernel": ("accumulate_dense_kernel", "reduce_dense_kernel")}


def traffic_of(traffic_file, timer, key="hbm_bytes_per_launch"):
    """A per-launch counter figure of the kernel behind a timer, from profiles/pmc_traffic.json (None if absent)."""
    if timer in TIMER_SUMS:
        vals = [traffic_file.get(fn, {}).get(key) for fn in TIMER_SUMS[timer] if isinstance(traffic_file.get(fn), dict)]
        vals = [v for v in vals if v]
        if len(vals) == len(TIMER_SUMS[timer]):
            return sum(vals)
    for fn in TIMER_KERNELS.get(timer, (timer,)):
        v = traffic_file.get(fn, {}).get(key) if isinstance(traffic_file.get(fn), dict) else None
        if v:
            return v
    return None


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
    ap.add_argument("--move-prob", type=float, default=None,
                    help="experiment: probability that an event of the synthetic stream moves the touch (default: the generator's 0.35); 1.0 makes every env-step one event long")
    ap.add_argument("--algo", default="q_lambda", choices=["q_lambda", "sarsa", "double_q"])
    ap.add_argument("--memory-size", type=int, default=20000000)
    ap.add_argument("--events", type=int, default=0, help="events per book (0 = 64 warm-up + 2048)")
    ap.add_argument("--replay", type=int, default=0, metavar="N_TOTAL",
                    help="config 5 instead of the headline: every book replays ONE recorded stream of N_TOTAL events "
                         "from its own phase (lob_load_events_shared); not the headline workload")
    ap.add_argument("--sustained", type=int, default=3, metavar="E",
                    help="after the timed leg, on the same engine: finish episode 1, then E whole episodes with theta carried over "
                         "(ClearInventory -> HandleTerminal -> schedules -> reset), each timed INCLUDING lob_reset (0 = off; N = 1 only)")
    ap.add_argument("--dense", type=int, default=200, metavar="STEPS",
                    help="... and one leg from a dense theta (learning.random_init: 2u - 1 for every weight, reference "
                         "src/rl/agent.cpp:37-39): STEPS timed steps of a fresh episode (0 = off; N = 1 only)")
    ap.add_argument("--epsilon", type=float, default=None, help="exploration rate of the timed leg (default: example.yaml's eps_init 0.8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    from rl_markets_amd import launch
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: be one.  N ranks of this very command, one GPU each; rank 0 prints the line.
        sys.exit(launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    rank, local_rank, world = launch.rank_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        sys.stderr.write("bench.py rank %d of %d on GPU %d\n" % (rank, world, local_rank))
        sys.stderr.flush()

    libs = [os.path.join(ROOT, "rl_markets_amd", "csrc", n) for n in ("liblob_engine.so", "liblob_comm.so")]
    if not all(os.path.exists(f) for f in libs):
        if rank == 0:
            import __graft_entry__ as ge
            ge.build()
        for _ in range(3000):                       # the other ranks wait for rank 0's build
            if all(os.path.exists(f) for f in libs):
                break
            time.sleep(0.1)
    from rl_markets_amd import abi, engine

    # run the N > 1 code path with one rank (1-GPU boxes): delta kernels + a one-rank RCCL all-reduce
    force_dist = os.environ.get("LOB_FORCE_DIST") == "1" and world == 1
    comm = None
    if world > 1 or force_dist:
        from rl_markets_amd.comm import MAX, SUM, RcclComm
        rdzv = launch.rendezvous_path() if world > 1 else os.path.join(tempfile.gettempdir(), "lob_rdzv_single_%d" % os.getpid())
        comm = RcclComm(rdzv, rank, world, local_rank)
        from rl_markets_amd.comm import pin_host_thread
        pinned_cpus = pin_host_thread(local_rank)   # this rank's launches / event reads stay on its GPU's socket

    p = engine.default_params()
    p.depth, p.max_trades = args.depth, 2
    p.algo = {"q_lambda": abi.ALGO_QLAMBDA, "sarsa": abi.ALGO_SARSA, "double_q": abi.ALGO_DOUBLE_Q}[args.algo]
    p.theta_mode = abi.THETA_SHARED
    p.memory_size = args.memory_size
    p.book_id_offset = rank * args.books
    if args.epsilon is not None:
        p.epsilon = args.epsilon
    g = engine.default_gen_params()
    g.n_events = args.events if args.events else 64 + 2048
    if args.move_prob is not None:
        g.move_prob_q16 = min(65535, int(args.move_prob * 65536))
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
    eng.kernel_timing(True)
    eng.reset()             # Initialise(): includes the once-per-episode market pre-pass over the whole stream
    eng.sync()
    reset_ms, _ = eng.kernel_time_ms("reset_kernel")
    eng.kernel_timing(False)

    from rl_markets_amd.parallel import EngineBackend, ShardedLearner
    learner = ShardedLearner(EngineBackend(eng), comm, sync_every=SYNC_EVERY)

    def barrier():
        eng.sync()
        hip_device_sync()
        if comm is not None:
            comm.barrier()

    tw0 = time.perf_counter()
    learner.run(args.warmup)
    barrier()
    warmup_s = time.perf_counter() - tw0
    if comm is not None:
        comm.exchange_stats()   # (drop the warm-up's)
    c0 = eng.counters()
    ps0 = eng.path_stats()
    if not args.no_kernel_timing:
        # HIP events around the kernels of every n-th step of the timed region (2-8 sampled steps): timing every
        # launch costs 9 % of a step (two event records per launch, eight launches per step)
        eng.kernel_timing(max(8, args.steps // 8))
    t0 = time.perf_counter()
    learner.run(args.steps)
    barrier()
    t1 = time.perf_counter()
    c1 = eng.counters()
    ps1 = eng.path_stats()
    elapsed = t1 - t0
    steps_done = int(c1[0] - c0[0])
    events_done = int(c1[1] - c0[1])

    ktimes = {}
    if not args.no_kernel_timing:
        for k in KERNELS:
            ms, n = eng.kernel_time_ms(k)
            if n:
                ktimes[k] = {"avg_ms": ms, "launches": n}
        eng.kernel_timing(False)
    books = eng.get_books(0, min(args.books, 4096))
    n_live = float(sum(b.n_traces for b in books)) / len(books)

    # ---- the regime the reference trains in (src/main.cpp:53-77: n_episodes over ONE agent): whole episodes with theta carried
    # over, timed including lob_reset; then a dense theta (learning.random_init).  After the driver-timed leg, same engine.
    sustained = dense = None
    if world == 1 and comm is None and not args.replay:
        if args.sustained > 0:
            sustained = sustained_leg(eng, learner, args, g, reset_ms, warmup_s + elapsed, int(c1[0]))
        if args.dense > 0 and args.algo != "double_q":
            dense = dense_leg(eng, learner, args, p)

    exchange = None
    if comm is not None:
        exchange = comm.exchange_stats()                  # HIP events on the engine stream around the exchange's phases (this rank)
        exchange["pinned_host_cpus"] = pinned_cpus
        exchange["ms_per_exchange"] = round(exchange["pack_ms"] + exchange["collectives_ms"] + exchange["apply_ms"], 4)
        exchange["sparse_counts"] = eng.exchange_debug()   # (whole run: how many exchanges went without a host synchronisation)
        elapsed = comm.reduce([elapsed], MAX)[0]          # the slowest rank's clock
        steps_done, events_done = (int(v) for v in comm.reduce([steps_done, events_done], SUM))

    result = None
    if rank == 0:
        eps = events_done / max(steps_done, 1)
        steps_per_episode = max((g.n_events - 64) / max(eps, 1e-9), 1.0)   # an episode = the stream after the 64-event warm-up
        traffic_file = {}
        # counter figures of THIS configuration's kernels (tools/final_validation.sh -> profiles/pmc_traffic_<cfg>.json; the headline's
        # file keeps its old name); a configuration without a tracked counter file reports traffic null, never another one's bytes
        cfg_key = config_key(args)
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json" if cfg_key == "c3" else "pmc_traffic_%s.json" % cfg_key)
        if os.path.exists(tf):
            try:
                traffic_file = json.load(open(tf))
            except Exception:
                traffic_file = {}
        roofline = None
        if ktimes:
            per_kernel = {}
            # Agent::UpdateTraces of a step is shared by up to three kernels (the light case inside the lane-per-book learn
            # kernel or in trace_light_kernel, the rest in the wave-per-book trace kernel): the step's trace bytes are set
            # against the time of the kernels that do nothing else, the learn kernel keeps its own yardstick
            trace_ms = sum(ktimes[k]["avg_ms"] for k in ("trace_kernel", "trace_light_kernel") if k in ktimes)
            for k, v in ktimes.items():
                if k.startswith("delta"):
                    continue
                per_book = algorithmic_bytes(k, args.depth, 2, p.n_vars, n_live, eps)
                if k == "trace_kernel" and trace_ms > 0:
                    per_book *= v["avg_ms"] / trace_ms
                elif k == "trace_light_kernel" and trace_ms > 0:
                    per_book = algorithmic_bytes("trace_kernel", args.depth, 2, p.n_vars, n_live, eps) * v["avg_ms"] / trace_ms
                live_books = steps_done / world / args.steps   # books one launch covers
                ach = per_book * live_books / (v["avg_ms"] * 1e-3) / 1e9
                tr = traffic_of(traffic_file, k)
                insts = traffic_of(traffic_file, k, "insts_per_launch")
                sec = v["avg_ms"] * 1e-3
                per_kernel[k] = {"avg_ms": round(v["avg_ms"], 4), "launches": v["launches"],
                                 # what the counters say this kernel moved / issued per launch (profiles/, same command), over THIS run's time
                                 "traffic": tr,
                                 "hbm_frac_counter": round(tr / sec / 1e9 / HBM_PEAK_GBS, 5) if tr else None,
                                 "insts_per_launch": insts,
                                 "issue_frac": round(insts / (sec * ISSUE_CLOCK_HZ * N_SIMDS), 5) if insts else None,
                                 # SURVEY.md 8(d)'s yardstick: the reference algorithm's bytes for this piece of the step
                                 "algorithmic_bytes_per_book": round(per_book, 1), "yardstick_GBps": round(ach, 1),
                                 "yardstick_frac": round(ach / HBM_PEAK_GBS, 5)}
            per_kernel["reset_kernel"] = {
                "avg_ms": round(reset_ms, 3), "launches": 1, "note": "once per episode, outside `value`; see value_amortised",
                "algorithmic_bytes_per_book": round(algorithmic_bytes("reset_kernel", args.depth, 2, p.n_vars, 0, g.n_events), 1),
                "yardstick_GBps": round(algorithmic_bytes("reset_kernel", args.depth, 2, p.n_vars, 0, g.n_events) * args.books / (reset_ms * 1e-3) / 1e9, 1) if reset_ms else None,
                "traffic": traffic_of(traffic_file, "reset_kernel")}
            # the dominant kernel = the largest share of a STEP: a kernel launched once per 64 steps (the track-ring refill) counts
            # with a 64th of its duration (its timer is on at every launch, the others' at the sampled steps only)
            sampled_steps = max(ktimes["env_kernel"]["launches"], 1) if "env_kernel" in ktimes else max(v["launches"] for v in ktimes.values())
            for k, v in ktimes.items():
                per_kernel.get(k, {})["ms_per_step"] = round(v["avg_ms"] * v["launches"] / (args.steps if k in ALWAYS_TIMED else sampled_steps), 5)
            dom = max((k for k in ktimes if not k.startswith("delta")), key=lambda k: per_kernel[k]["ms_per_step"])
            d = per_kernel[dom]
            # the whole step against the same roof: SURVEY.md 8(d)'s bytes per env-step (event read + book state + scalars +
            # Q-value gathers + n_live traces' worth of index / eligibility / theta read-modify-write) x env-steps per second
            step_bytes = eps * (2 * args.depth * 8 + 2 * 8) + eps * 2 * args.depth * 8 + 256 + 9 * 3 * 32 * 8 + n_live * (4 + 4 + 4 + 8 + 8)
            dsec = d["avg_ms"] * 1e-3
            # `achieved` / `frac`: the dominant kernel's COUNTER bytes (FETCH_SIZE + WRITE_SIZE of the rocprofv3 --pmc passes kept
            # under profiles/) over its HIP-event time in this run, against the HBM roof; the SURVEY yardstick under its own key.
            # Without a counter file: the yardstick.
            use_counter = d["traffic"] is not None
            roofline = {"bound": "hbm", "kernel": dom,
                        "achieved": round(d["traffic"] / dsec / 1e9, 1) if use_counter else d["yardstick_GBps"],
                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": d["hbm_frac_counter"] if use_counter else d["yardstick_frac"],
                        "frac_is": "counter bytes / launch time / peak" if use_counter else "SURVEY 8(d) yardstick bytes / launch time / peak",
                        "traffic": d["traffic"],
                        "traffic_source": traffic_file.get("_source", "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command; static file, not measured in this run)") if d["traffic"] else None,
                        "issue_frac": d["issue_frac"],
                        "issue_note": "wave instructions of all kinds per launch (SQ_INSTS_*) / (launch time x 2.4 GHz x 1024 SIMDs x 1 instruction per clock): what these latency-chain kernels are nearest to is neither roof -- one wave per SIMD waiting on dependent memory round trips (DESIGN.md section 4)",
                        "yardstick": {"algorithmic_bytes_per_book": d["algorithmic_bytes_per_book"], "GBps": d["yardstick_GBps"], "frac": d["yardstick_frac"]},
                        "whole_step": {"algorithmic_bytes_per_env_step": round(step_bytes, 1),
                                       "yardstick_GBps": round(step_bytes * steps_done / elapsed / 1e9, 1),
                                       "yardstick_frac": round(step_bytes * steps_done / elapsed / 1e9 / HBM_PEAK_GBS, 5),
                                       "counter_bytes_per_step": round(sum(v["traffic"] * v["launches"] / max(per_kernel[dom]["launches"], 1) for k_, v in per_kernel.items()
                                                                           if v.get("traffic") and k_ != "reset_kernel"), 1),
                                       "note": "yardstick = SURVEY.md 8(d) (the reference algorithm's bytes per env-step); counter_bytes_per_step = sum over the step's kernels of FETCH_SIZE + WRITE_SIZE per launch: the memo / hit-list kernels move far fewer bytes than the yardstick assumes"},
                        "books_per_launch": round(steps_done / world / args.steps, 1),
                        "avg_launch_ms": d["avg_ms"],
                        "all_kernels_avg_ms": {k: round(v["avg_ms"], 4) for k, v in ktimes.items()},
                        "per_kernel": per_kernel}
            ws = roofline["whole_step"]
            if ws["counter_bytes_per_step"]:
                ws["hbm_frac_counter"] = round(ws["counter_bytes_per_step"] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5)
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only (the contract); N > 1 runs report null
            try:
                cpu = cpu_baseline()
            except Exception as ex:  # the baseline is reported, never required
                cpu = {"error": str(ex)}
        ms_step_only = elapsed / args.steps * 1e3
        # `value` is the whole-episode figure: the once-per-episode lob_reset (the market pre-pass, into which the market half of
        # every NextState was hoisted) is charged to every step at reset_ms / steps_per_episode, on top of the measured time of
        # the K timed steps.  The step-only figure stays under its own keys.
        ms_per_step = ms_step_only + reset_ms / steps_per_episode
        out = {
            "metric": "env-steps/sec (whole node) at 65 536 parallel books",
            "value": steps_done / (ms_per_step * args.steps * 1e-3),
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "value_is": "env-steps of the K timed steps / (their wall time + K x reset_ms_per_episode / steps_per_episode): whole-episode throughput",
            "value_step_only": steps_done / elapsed,
            "ms_per_step_step_only": ms_step_only,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "value_amortised": steps_done / (ms_per_step * args.steps * 1e-3),   # (= value since round 4; kept for older readers)
            "sustained": sustained,
            "dense_theta": dense,
            "config": {
                "workload": (workload_label(args, world) + ": " +
                             ("%d books replaying one recorded %d-level stream from per-book phases, reward pnl_damped, %s, "
                              if args.replay else "%d parallel synthetic %d-level books per GPU, %s with eligibility traces, ") +
                             "tile-coded linear Q (32 tilings x 3 groups x 9 actions), memory_size %d, "
                             "shared theta, synchronous-batch TD") % (args.books, args.depth,
                                                                   {"q_lambda": "Q(lambda)", "sarsa": "SARSA(lambda)", "double_q": "double Q(lambda)"}[args.algo],
                                                                   args.memory_size),
                "traffic_profile": cfg_key,
                "books_per_gpu": args.books, "depth": args.depth, "events_per_book": g.n_events,
                "events_per_step": round(eps, 4), "live_traces_per_book": round(n_live, 1),
                "move_prob": args.move_prob,   # (None: the generator's default; an experiment knob otherwise)
                "env_steps": steps_done, "reset_ms_per_episode": round(reset_ms, 2), "steps_per_episode": round(steps_per_episode, 1),
                # diagnostics of the fast paths (lob_get_path_stats), per timed step where cumulative
                "paths": {"trace_lane_handed_back_per_step": round(float(ps1[0] - ps0[0]) / max(args.steps, 1), 1),
                          "memo_slots_registered": int(ps1[2]), "ambiguous_indices": int(ps1[3]), "registry_overflow": int(ps1[4]),
                          "memo_slots_in_step": int(ps1[5])},
                "sync_every": SYNC_EVERY if comm is not None else None,
                "exchange": exchange,
                "parallelism": ("%d book shard(s), one process per GPU, every %d steps the ranks' written-weights maps all-gathered and the packed delta-theta of their union all-reduced over RCCL" % (world, SYNC_EVERY))
                               if comm is not None else "1 shard",
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        result = json.dumps(out)
    if comm is not None:
        comm.barrier()
        comm.close()
    eng.close()
    if rank == 0:
        # last line of output, after RCCL is torn down (it may print through C stdio, which is
        # block-buffered on a pipe): flush that first
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(result, flush=True)


def workload_label(args, world):
    """Which of BASELINE.json's configurations this run is -- or that it is none of them."""
    plain = args.depth == 10 and args.memory_size == 20000000 and args.epsilon is None and args.move_prob is None
    if args.replay:
        return "C5" if plain and args.books == 65536 else "C5-shaped (not a BASELINE configuration)"
    if plain and args.books == 4096 and args.algo == "sarsa" and world == 1:
        return "C2"
    if plain and args.books == 65536 and args.algo == "q_lambda":
        return "C3" if world == 1 else "C4" if world == 8 else "C4-shaped (65 536 books per GPU on %d GPUs)" % world
    return "variant of C3 (not a BASELINE configuration)"


def config_key(args):
    """The tracked counter profile this run's `traffic` figures come from (profiles/pmc_traffic*.json)."""
    if args.replay:
        return "c5"
    if args.books == 4096 and args.algo == "sarsa":
        return "c2"
    if args.books != 65536 or args.move_prob is not None:
        return "other"
    if args.epsilon is not None:
        return "eps%s" % ("%g" % args.epsilon).replace("0.", "").replace(".", "_") if args.algo == "q_lambda" else "other"
    return {"q_lambda": "c3", "sarsa": "sarsa", "double_q": "double_q"}[args.algo]


ALWAYS_TIMED = ("prepass_extend_kernel", "delta_begin_kernel", "delta_apply_kernel", "reset_kernel")


def episode_schedule(ep):
    """Agent::HandleTerminal(ep) + EpsilonGreedy::HandleTerminal(ep) with config/example.yaml's constants
    (src/rl/agent.cpp:103-109, src/rl/policy.cpp:79-82): alpha, epsilon of episode ep + 1."""
    return max(0.001, 0.001 * 1.0 ** ep), 0.8 * (0.0001 / 0.8) ** (ep / 800.0)


def path_delta(a, b):
    return {"hit_list_replay": int(b[1] - a[1]), "act_in_full_in_kernel": int(b[6] - a[6]), "learn_q_rest": int(b[7] - a[7]),
            "trace_lane_handed_back": int(b[0] - a[0])}


def fp_brief(st):
    return {k: v for k, v in st.items() if k != "hist"}


def run_to_end(eng, learner, first_step, chunk=32, samples=(256, 1024)):
    """Learner steps until no book is live (Runner::RunEpisode's loop, the live count read every `chunk` steps).
    Returns (steps run, fast-path samples at the given episode steps)."""
    n, out = 0, {}
    while True:
        learner.run(chunk)
        n += chunk
        for sp in samples:
            if first_step + n - chunk < sp <= first_step + n:
                out["step_%d" % sp] = fp_brief(eng.fastpath_stats())
        if eng.counters()[2] == 0:
            return n, out


def kernel_times(eng):
    out = {}
    for k in KERNELS:
        ms, n = eng.kernel_time_ms(k)
        if n:
            out[k] = round(ms, 4)
    return out


def sustained_leg(eng, learner, args, g, reset_ms, ep1_seconds, ep1_env_steps_so_far):
    """Episode 1 to its end, then args.sustained whole episodes on the same agent."""
    episodes = []
    t0 = time.perf_counter()
    c0, ps0 = eng.counters(), eng.path_stats()
    n, smp = run_to_end(eng, learner, args.warmup + args.steps)
    eng.sync(); hip_device_sync()
    t1 = time.perf_counter()
    c1, ps1 = eng.counters(), eng.path_stats()
    sec = reset_ms * 1e-3 + ep1_seconds + (t1 - t0)
    episodes.append({"episode": 1, "env_steps": int(c1[0]), "learner_steps": args.warmup + args.steps + n,
                     "seconds_incl_reset": round(sec, 4), "reset_ms": round(reset_ms, 2),
                     "env_steps_per_s_incl_reset": round(c1[0] / sec, 1),
                     "note": "reset + warm-up + timed leg + the rest of the episode (the timed leg's sampled kernel timers included)",
                     "paths_after_timed_leg": path_delta(ps0, ps1), "fastpath": smp})
    for ep in range(1, args.sustained + 1):
        last = ep == args.sustained
        t0 = time.perf_counter()
        eng.clear_inventory()                 # Runner::RunEpisode epilogue (serial.cpp:31)
        eng.handle_terminal()                 # Agent::HandleTerminal: traces.decay(0), schedules
        alpha, eps = episode_schedule(ep - 1)
        eng.set_alpha(alpha)
        if args.epsilon is None:
            eng.set_epsilon(eps)
        eng.kernel_timing(True)
        eng.reset()
        eng.sync()
        r_ms, _ = eng.kernel_time_ms("reset_kernel")
        eng.kernel_timing(16 if last else False)   # the last episode: per-kernel times of every 16th step
        cs, ps0 = eng.counters(), eng.path_stats()
        n, smp = run_to_end(eng, learner, 0)
        eng.sync(); hip_device_sync()
        t1 = time.perf_counter()
        ce, ps1 = eng.counters(), eng.path_stats()
        e = {"episode": ep + 1, "env_steps": int(ce[0] - cs[0]), "learner_steps": n, "seconds_incl_reset": round(t1 - t0, 4),
             "reset_ms": round(r_ms, 2), "epsilon": round(eps, 5) if args.epsilon is None else args.epsilon,
             "env_steps_per_s_incl_reset": round((ce[0] - cs[0]) / (t1 - t0), 1),
             "memo_slots_registered": int(ps1[2]), "paths": path_delta(ps0, ps1), "fastpath": smp}
        if last:
            e["kernels_avg_ms"] = kernel_times(eng)
            eng.kernel_timing(False)
        episodes.append(e)
    v1 = episodes[0]["env_steps_per_s_incl_reset"]
    return {"episodes": episodes, "last_over_first": round(episodes[-1]["env_steps_per_s_incl_reset"] / v1, 4) if v1 else None,
            "what": "one agent across episodes (reference src/main.cpp:53-77, config/example.yaml n_episodes): every figure includes "
                    "lob_reset, the host's end-of-episode calls and the steps in which only a tail of the books is still live"}


def dense_leg(eng, learner, args, p):
    """learning.random_init (reference src/rl/agent.cpp:37-39): every weight 2u - 1.  A fresh episode from that theta."""
    import numpy as np
    th = np.random.default_rng(1994).random(args.memory_size) * 2.0 - 1.0
    eng.clear_inventory()
    eng.handle_terminal()
    eng.set_theta(th)
    del th
    eng.kernel_timing(True)
    eng.reset()
    eng.sync()
    r_ms, _ = eng.kernel_time_ms("reset_kernel")
    eng.kernel_timing(False)
    learner.run(20)
    eng.sync(); hip_device_sync()
    c0, ps0 = eng.counters(), eng.path_stats()
    eng.kernel_timing(max(8, args.dense // 8))
    t0 = time.perf_counter()
    learner.run(args.dense)
    eng.sync(); hip_device_sync()
    t1 = time.perf_counter()
    c1, ps1 = eng.counters(), eng.path_stats()
    kt = kernel_times(eng)
    eng.kernel_timing(False)
    steps, events = int(c1[0] - c0[0]), int(c1[1] - c0[1])
    eps = events / max(steps, 1)
    # with every weight non-zero the 9 x 96 weight gathers of a Q evaluation really move (8 974 B per env-step
    # of the yardstick are mostly these): bytes per env-step as in the main line's whole_step
    step_bytes = eps * (2 * args.depth * 8 + 2 * 8) + eps * 2 * args.depth * 8 + 256 + 9 * 3 * 32 * 8 + 44 * (4 + 4 + 4 + 8 + 8)
    sec = t1 - t0
    return {"env_steps_per_s": round(steps / sec, 1), "ms_per_step": round(sec / args.dense * 1e3, 4), "steps": args.dense,
            "reset_ms": round(r_ms, 2), "paths": path_delta(ps0, ps1), "fastpath": fp_brief(eng.fastpath_stats()), "kernels_avg_ms": kt,
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_env_step": round(step_bytes, 1),
                         "achieved": round(step_bytes * steps / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(step_bytes * steps / sec / 1e9 / HBM_PEAK_GBS, 5)},
            "what": "theta = 2u - 1 for all %d weights, loaded through lob_theta_set before a fresh lob_reset; 20 untimed + %d timed learner steps" % (args.memory_size, args.dense)}


KERNELS = ("act_kernel", "act_rest_kernel", "env_kernel", "env_rest_kernel", "memo_kernel", "trace_light_kernel", "trace_kernel", "learn_kernel", "learn_rest_kernel",
           "update_kernel", "accumulate_kernel", "apply_kernel", "prepass_extend_kernel", "delta_begin_kernel", "delta_apply_kernel")

if __name__ == "__main__":
    main()
