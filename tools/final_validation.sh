set -u
# Runs ON THE GPU BOX (through gpurun), end of a round (R below): the driver's command twice, the counter profiles of the same command (no
# sustained / dense legs under the profiler), kernel statistics + FETCH_SIZE / WRITE_SIZE of every other configuration the
# documents quote (-> profiles/r05_<cfg>_*), the benches.  Outputs under gpurun_out/; tools/collect_round.sh copies what is kept
# to profiles/.   usage: bash tools/final_validation.sh [quick]   (quick: no full test suite)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
R=${ROUND:-r06}
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${R}_driver1.json 2>gpurun_out/bench_${R}_driver1.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${R}_driver2.json 2>/dev/null
# ---- the headline configuration: kernel statistics + six counter passes of the driver's command ----
export PROF_CMD="python $GRAFT_REPO_ROOT/bench.py --gpus 1 --no-cpu-baseline --no-kernel-timing --sustained 0 --dense 0"
bash tools/profile.sh ${R} "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" > gpurun_out/profile_${R}.log 2>&1
# ---- every other configuration the documents quote: kernel statistics + HBM bytes ----
prof_cfg() {  # name, bench flags
  local name=$1; shift
  export PROF_CMD="python $GRAFT_REPO_ROOT/bench.py --gpus 1 --no-cpu-baseline --no-kernel-timing --sustained 0 --dense 0 $*"
  bash tools/profile.sh ${R}_$name "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/profile_${R}_$name.log 2>&1
  rm -rf /tmp/prof_stats /tmp/prof_pmc_*
}
rm -rf /tmp/prof_stats /tmp/prof_pmc_*
prof_cfg sarsa --algo sarsa
prof_cfg c2 --books 4096 --algo sarsa
prof_cfg double_q --algo double_q
prof_cfg c5 --replay 61200 --events 50000
prof_cfg eps01 --epsilon 0.01
cd $GRAFT_REPO_ROOT
# ---- the benches (200 timed steps) ----
python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_${R}_200.json 2>/dev/null
python bench.py --gpus 1 --algo sarsa --steps 200 --warmup 20 --no-cpu-baseline --dense 0 > gpurun_out/bench_${R}_sarsa.json 2>/dev/null
python bench.py --gpus 1 --algo double_q --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_${R}_dq.json 2>/dev/null
python bench.py --gpus 1 --books 4096 --algo sarsa --steps 200 --warmup 20 --no-cpu-baseline --sustained 0 --dense 0 > gpurun_out/bench_${R}_c2.json 2>/dev/null
python bench.py --gpus 1 --replay 61200 --events 50000 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_${R}_c5.json 2>/dev/null
python bench.py --gpus 1 --books 262144 --events 300 --steps 200 --warmup 20 --no-cpu-baseline --sustained 0 --dense 0 > gpurun_out/bench_${R}_262k.json 2>/dev/null
LOB_FORCE_DIST=1 python bench.py --gpus 1 --steps 256 --warmup 64 --no-cpu-baseline > gpurun_out/bench_${R}_dist.json 2>/dev/null
timeout 600 python tools/exp_upload.py 16384 > gpurun_out/exp_upload_${R}.txt 2>&1
python bench.py --gpus 1 --epsilon 0.01 --steps 200 --warmup 20 --no-cpu-baseline --sustained 0 --dense 0 > gpurun_out/bench_${R}_eps01.json 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_${R}_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        s = d.get("sustained")
        print(f, "value %.1f M  step-only %.1f M  ms/step %.4f" % (d["value"] / 1e6, d["value_step_only"] / 1e6, d["ms_per_step"]),
              ("sustained " + " ".join("%.1f" % (e["env_steps_per_s_incl_reset"] / 1e6) for e in s["episodes"])) if s else "",
              ("dense %.1f M" % (d["dense_theta"]["env_steps_per_s"] / 1e6)) if d.get("dense_theta") else "",
              d["roofline"]["all_kernels_avg_ms"] if d.get("roofline") else "")
    except Exception as ex:
        print(f, "FAILED", ex)
PY
if [ "${1:-}" != quick ]; then
  python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" > gpurun_out/gputest_${R}.log
  cat gpurun_out/gputest_${R}.log
fi
