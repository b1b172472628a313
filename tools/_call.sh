python -m pytest tests/test_gpu_parity.py -x -q -k "trace_lane or q_lane or combined or hit_list or carry or memo" 2>&1 | grep -E "passed|failed|FAILED|Error" > gpurun_out/r5_t9.log
python -m pytest tests/test_gpu_steady.py tests/test_gpu_halfstep.py tests/test_gpu_delta_exchange.py -x -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert" >> gpurun_out/r5_t9.log
cat gpurun_out/r5_t9.log
for v in 1 0 1 0; do echo "== LOB_REST_MERGE=$v"; LOB_REST_MERGE=$v python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --dense 0 --sustained 0 2>/dev/null | python tools/benchline.py; done
for v in 1 0; do echo "== dq LOB_REST_MERGE=$v"; LOB_REST_MERGE=$v python bench.py --gpus 1 --algo double_q --steps 200 --warmup 20 --no-cpu-baseline --dense 0 --sustained 0 2>/dev/null | python tools/benchline.py; done
