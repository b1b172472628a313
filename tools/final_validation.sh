set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/gputest_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_final.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_final_driver1.json 2>gpurun_out/bench_final_driver1.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_final_driver2.json 2>/dev/null
bash tools/profile.sh r03 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" > gpurun_out/profile_final.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_final_200.json 2>/dev/null
python bench.py --gpus 1 --algo sarsa --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_final_sarsa.json 2>/dev/null
python bench.py --gpus 1 --algo double_q --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_final_dq.json 2>/dev/null
python bench.py --gpus 1 --books 4096 --algo sarsa --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_final_c2.json 2>/dev/null
python bench.py --gpus 1 --replay 61200 --events 50000 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_final_c5.json 2>/dev/null
python bench.py --gpus 1 --books 262144 --events 300 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_final_262k.json 2>/dev/null
LOB_FORCE_DIST=1 python bench.py --gpus 1 --steps 256 --warmup 64 --no-cpu-baseline > gpurun_out/bench_final_dist.json 2>/dev/null
python bench.py --gpus 1 --events 7264 --steps 1000 --warmup 200 --no-cpu-baseline > gpurun_out/bench_final_long.json 2>/dev/null
python bench.py --gpus 1 --algo sarsa --events 7264 --steps 1000 --warmup 200 --no-cpu-baseline > gpurun_out/bench_final_long_sarsa.json 2>/dev/null
cat gpurun_out/gputest_final.log gpurun_out/smoke_final.log
