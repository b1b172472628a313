#!/bin/bash
# GPU box: time bench.py with each ablation build of the engine found in rl_markets_amd/csrc/_abl/
# (built by hand from a scratch copy of csrc/ with -DABL_* switches; never committed).  Every run is
# under `timeout`: an ablation that removes a loop's exit condition must not eat the GPU budget.
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp rl_markets_amd/csrc/liblob_engine.so /tmp/keep.so
for f in rl_markets_amd/csrc/_abl/*.so; do
  cp $f rl_markets_amd/csrc/liblob_engine.so
  echo -n "$(basename $f) "; timeout 90 python bench.py --no-cpu-baseline $ABL_ARGS 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), d['roofline']['all_kernels_avg_ms'])" || echo failed
done
cp /tmp/keep.so rl_markets_amd/csrc/liblob_engine.so
