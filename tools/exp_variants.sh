#!/bin/bash
# A/B of engine builds that differ in -D switches.
#   tools/exp_variants.sh build "name1:-DX=1 -DY=2" "name2:..."   (here: hipcc cross-compiles without a GPU)
#       -> rl_markets_amd/csrc/_var/<name>/liblob_engine.so (git-ignored; travels to the GPU box with the snapshot)
#   tools/exp_variants.sh run name1 name2 ...                      (GPU box: bench.py through each, LOB_ENGINE_LIB)
#   tools/exp_variants.sh "name1:-DX=1" ...                        (GPU box: build + run, as before)
# BENCH_ARGS adds bench.py flags.  Every run is under `timeout`.
cd ${GRAFT_REPO_ROOT:-/root/repo}
CS=rl_markets_amd/csrc
# (the library's translation units compiled in parallel and linked: __graft_entry__.build_engine; the kernel variants measured
# and lost -- env_compact_kernel, the two-wave pre-pass, 32-lane env_step_kernel, two book groups, LOB_ACC_REPS > 8,
# 16 / 32-lane reset kernels -- exist only with -DLOB_EXPERIMENTS:   tools/exp_variants.sh build "exp:-DLOB_EXPERIMENTS"
# then LOB_ENGINE_LIB=$PWD/rl_markets_amd/csrc/_var/exp/liblob_engine.so python -m pytest tests -m gpu -k "books_per_wave or two_waves")
build_one() {
  mkdir -p $CS/_var/$1
  python -c "import sys, __graft_entry__ as g; g.build_engine(sys.argv[1], sys.argv[2].split(), sys.argv[3])" \
     $PWD/$CS/_var/$1/liblob_engine.so "$2" $PWD/$CS/_var/$1/_obj 2>/dev/null
}
run_one() {
  echo -n "$1 "
  LOB_ENGINE_LIB=$PWD/$CS/_var/$1/liblob_engine.so timeout 120 python bench.py --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python tools/benchline.py || echo failed
}
mode=both
if [ "$1" = build ] || [ "$1" = run ]; then mode=$1; shift; fi
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  if [ $mode != run ]; then build_one $name "$flags" || { echo "$name build failed"; continue; }; fi
  if [ $mode != build ]; then run_one $name; fi
done
