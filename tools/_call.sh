python -m pytest tests/test_gpu_steady.py -x -q -k "come_on_and_go_off or dense_slot_ids or (steady_state and qlambda)" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" > gpurun_out/r5_t11.log
cat gpurun_out/r5_t11.log
for v in 1 0 1 0; do echo "== LOB_ACC_DENSE=$v"; LOB_ACC_DENSE=$v python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --dense 0 --sustained 0 2>/dev/null | python tools/benchline.py; done
