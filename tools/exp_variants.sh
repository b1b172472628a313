#!/bin/bash
# GPU box: build liblob_engine.so with different -D switches into rl_markets_amd/csrc/_var/<name>/ and time
# bench.py through each (LOB_ENGINE_LIB).  Usage: tools/exp_variants.sh "name1:-DX=1 -DY=2" "name2:..." ...
# Every run is under `timeout`.
cd ${GRAFT_REPO_ROOT:-/root/repo}
CS=rl_markets_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  mkdir -p $CS/_var/$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value $flags \
     -o $CS/_var/$name/liblob_engine.so $CS/lob_engine.hip $CS/lob_host.cpp 2>/dev/null || { echo "$name build failed"; continue; }
  echo -n "$name [$flags] "
  LOB_ENGINE_LIB=$PWD/$CS/_var/$name/liblob_engine.so timeout 120 python bench.py --no-cpu-baseline $BENCH_ARGS 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), d['roofline']['all_kernels_avg_ms'])" || echo failed
done
