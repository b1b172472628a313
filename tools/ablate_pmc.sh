#!/bin/bash
# GPU box: dynamic instruction counts (rocprofv3 --pmc) of act_kernel / learn_kernel for each ablation
# build in rl_markets_amd/csrc/_abl/ (see tools/ablate.sh)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cp $REPO/rl_markets_amd/csrc/liblob_engine.so /tmp/keep.so
for f in $REPO/rl_markets_amd/csrc/_abl/*.so; do
  cp $f $REPO/rl_markets_amd/csrc/liblob_engine.so
  n=$(basename $f .so)
  timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d /tmp/ab_$n -- python $REPO/bench.py --steps 10 --warmup 150 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  python - "$n" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/ab_%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("act_kernel") or k.startswith("learn_kernel"):
            a = acc[(k.split("<")[0], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(sys.argv[1], {k[0][:5] + ":" + k[1][9:]: round(v[0] / v[1] / 65536) for k, v in sorted(acc.items())})
PY
done
cp /tmp/keep.so $REPO/rl_markets_amd/csrc/liblob_engine.so
