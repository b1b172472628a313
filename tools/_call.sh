python -m pytest tests/test_gpu_steady.py -x -q -k "dense_slot_ids or steady_state" 2>&1 | tail -4 > gpurun_out/r5_t3.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -x -q -k "trace_lane or config2 or combined_update or model_log or random_init" 2>&1 | tail -4 >> gpurun_out/r5_t3.log
tail -8 gpurun_out/r5_t3.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sarsa -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --algo sarsa --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --dense 0 --sustained 0 > $GRAFT_REPO_ROOT/gpurun_out/r5_prof_sarsa.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_prof.py r5s /tmp/prof_sarsa $GRAFT_REPO_ROOT/gpurun_out/prof > /dev/null 2>&1
head -16 $GRAFT_REPO_ROOT/gpurun_out/prof/r5s_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --algo sarsa --steps 200 --warmup 20 --no-cpu-baseline --dense 0 --sustained 0 | python tools/benchline.py
python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --dense 0 --sustained 0 | python tools/benchline.py
python bench.py --gpus 1 --books 4096 --algo sarsa --steps 200 --warmup 20 --no-cpu-baseline --sustained 0 --dense 0 | python tools/benchline.py
LOB_ACC_DENSE=0 python bench.py --gpus 1 --books 4096 --algo sarsa --steps 200 --warmup 20 --no-cpu-baseline --sustained 0 --dense 0 | python tools/benchline.py
