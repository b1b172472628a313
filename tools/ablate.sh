#!/bin/bash
# GPU box: time bench.py with each ablation build of the engine (tools: built into csrc/_abl by hand)
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp rl_markets_amd/csrc/liblob_engine.so /tmp/keep.so
for f in rl_markets_amd/csrc/_abl/*.so; do
  cp $f rl_markets_amd/csrc/liblob_engine.so
  echo -n "$(basename $f) "; python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), d['roofline']['all_kernels_avg_ms'])"
done
cp /tmp/keep.so rl_markets_amd/csrc/liblob_engine.so
