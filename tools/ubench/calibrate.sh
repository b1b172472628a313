#!/bin/bash
# GPU box: calibrate FETCH_SIZE on this code base's own access pattern (random 8-byte gathers from a
# 160 MB table): tools/ubench/gather issues a known number of lane-loads per launch.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE "TCC_MISS_sum TCC_REQ_sum"; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/cal_$(echo $c | cut -c1-5) -- $REPO/tools/ubench/gather > /dev/null 2>&1
done
python - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("/tmp/cal_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gather" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
# launch order in gather.hip: dep 0/1 x active 64/32/16/4, 5 repetitions each
for name, v in acc.items():
    per = [sum(v[i*5:(i+1)*5])/5 for i in range(len(v)//5)]
    print(name, [round(x, 1) for x in per])
loads = [65536*a*16*(1+d) for d in (0, 1) for a in (64, 32, 16, 4)]
print("lane_loads", loads)
if "FETCH_SIZE" in acc:
    v = acc["FETCH_SIZE"]; per = [sum(v[i*5:(i+1)*5])/5 for i in range(len(v)//5)]
    print("FETCH_SIZE bytes per lane-load", [round(p*1024/l, 1) for p, l in zip(per, loads)])
PY
