#!/bin/bash
# A/B of engine builds that differ in -D switches.
#   tools/exp_variants.sh build "name1:-DX=1 -DY=2" "name2:..."   (here: hipcc cross-compiles without a GPU)
#       -> rl_markets_amd/csrc/_var/<name>/liblob_engine.so (git-ignored; travels to the GPU box with the snapshot)
#   tools/exp_variants.sh run name1 name2 ...                      (GPU box: bench.py through each, LOB_ENGINE_LIB)
#   tools/exp_variants.sh "name1:-DX=1" ...                        (GPU box: build + run, as before)
# BENCH_ARGS adds bench.py flags.  Every run is under `timeout`.
cd ${GRAFT_REPO_ROOT:-/root/repo}
CS=rl_markets_amd/csrc
build_one() {
  mkdir -p $CS/_var/$1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value $2 \
     -o $CS/_var/$1/liblob_engine.so $CS/lob_engine.hip $CS/lob_host.cpp 2>/dev/null
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
