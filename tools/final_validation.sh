set -u
# Runs ON THE GPU BOX (through gpurun), end of round 4: the driver's command twice, the profiles of the same command (no
# sustained / dense legs under the profiler), the secondary configurations.  Outputs under gpurun_out/; copy what is kept to profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r4_driver1.json 2>gpurun_out/bench_r4_driver1.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r4_driver2.json 2>/dev/null
export PROF_CMD="python $GRAFT_REPO_ROOT/bench.py --gpus 1 --no-cpu-baseline --no-kernel-timing --sustained 0 --dense 0"
bash tools/profile.sh r04 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" > gpurun_out/profile_r4.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_r4_200.json 2>/dev/null
python bench.py --gpus 1 --algo sarsa --steps 200 --warmup 20 --no-cpu-baseline --dense 0 > gpurun_out/bench_r4_sarsa.json 2>/dev/null
python bench.py --gpus 1 --algo double_q --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_r4_dq.json 2>/dev/null
python bench.py --gpus 1 --books 4096 --algo sarsa --steps 200 --warmup 20 --no-cpu-baseline --sustained 0 --dense 0 > gpurun_out/bench_r4_c2.json 2>/dev/null
python bench.py --gpus 1 --replay 61200 --events 50000 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_r4_c5.json 2>/dev/null
python bench.py --gpus 1 --books 262144 --events 300 --steps 200 --warmup 20 --no-cpu-baseline --sustained 0 --dense 0 > gpurun_out/bench_r4_262k.json 2>/dev/null
LOB_FORCE_DIST=1 python bench.py --gpus 1 --steps 256 --warmup 64 --no-cpu-baseline > gpurun_out/bench_r4_dist.json 2>/dev/null
python bench.py --gpus 1 --epsilon 0.01 --steps 200 --warmup 20 --no-cpu-baseline --sustained 0 --dense 0 > gpurun_out/bench_r4_eps001.json 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_r4_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.1f M  step-only %.1f M  ms/step %.4f" % (d["value"] / 1e6, d["value_step_only"] / 1e6, d["ms_per_step"]), d["roofline"]["all_kernels_avg_ms"] if d.get("roofline") else "")
    except Exception as ex:
        print(f, "FAILED", ex)
PY
