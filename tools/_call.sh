python -m pytest tests/test_gpu_steady.py -x -q -k "longer_than or come_on_and_go_off or preloaded or steady_state" 2>&1 | grep -E "passed|failed|FAILED|Error|assert|long lists" > gpurun_out/r5_t12.log
python -m pytest tests/test_gpu_parity.py -x -q -k "hit_list or q_lane or shared_theta or config3" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" >> gpurun_out/r5_t12.log
cat gpurun_out/r5_t12.log
python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --dense 0 > gpurun_out/r5_b12_q.json 2>/dev/null
python bench.py --gpus 1 --algo double_q --steps 200 --warmup 20 --no-cpu-baseline --dense 0 > gpurun_out/r5_b12_dq.json 2>/dev/null
python bench.py --gpus 1 --algo sarsa --steps 200 --warmup 20 --no-cpu-baseline --dense 0 > gpurun_out/r5_b12_sarsa.json 2>/dev/null
python - <<PY
import json
for f in ("q","dq","sarsa"):
    d = json.loads(open("gpurun_out/r5_b12_%s.json" % f).read().strip().splitlines()[-1])
    s = d["sustained"]
    print(f, "value %.1f step-only %.1f" % (d["value"]/1e6, d["value_step_only"]/1e6), "sustained", " ".join("%.1f" % (e["env_steps_per_s_incl_reset"]/1e6) for e in s["episodes"]), s["episodes"][-1].get("paths"), s["episodes"][-1].get("kernels_avg_ms"))
PY
