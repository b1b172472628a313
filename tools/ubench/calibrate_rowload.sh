#!/bin/bash
# GPU box: what does FETCH_SIZE count for the access pattern of env_step_kernel / reset_kernel -- every lane reading its OWN 224-byte
# record with twelve 16-byte loads (tools/ubench/rowload mode 0), against the 14-lanes-per-record cooperative form (mode 1)?
# The 8-byte-gather calibration (calibrate.sh, profiles/r01_fetch_calibration.txt) does not cover 16-byte loads; the guide says
# FETCH_SIZE halves wide coalesced streams (VERDICT r5 next #1e).  One --pmc pass per counter group, kernel-trace never combined.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
[ -x $REPO/tools/ubench/rowload ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $REPO/tools/ubench/rowload $REPO/tools/ubench/rowload.hip
for c in FETCH_SIZE "TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/calrow_$(echo $c | cut -c1-9 | tr ' ' '_') -- $REPO/tools/ubench/rowload > /dev/null 2>&1
done
python - <<'PY' | tee $REPO/gpurun_out/prof/rowload_fetch_calibration.txt
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("/tmp/calrow_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rowload" in k:
            m = re.search(r"rowload<(\d+),\s*(\d+)>", k)
            acc[(int(m.group(1)), int(m.group(2)))][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# tools/ubench/calibrate_rowload.sh on MI355X: 65 536 lanes (1 024 one-wave blocks), each reading R consecutive 224-byte records")
print("# of its own stream (12 x 16-byte loads per record in mode 0; 14 lanes x 16 bytes per record through LDS in mode 1).")
print("# algorithmic bytes per launch = 65 536 x R x 192 (mode 0 reads quads 1..12) or x 224 (mode 1 reads all 14 quads).")
for (mode, R), cs in sorted(acc.items()):
    alg = 65536 * R * (192 if mode == 0 else 224)
    line = "mode %d R %d: algorithmic %.1f MB" % (mode, R, alg / 1e6)
    for name, v in sorted(cs.items()):
        avg = sum(v) / len(v)
        if name in ("FETCH_SIZE", "WRITE_SIZE"):
            line += "; %s %.1f KiB = %.1f MB (x %.3f of algorithmic)" % (name, avg, avg * 1024 / 1e6, avg * 1024 / alg)
        else:
            line += "; %s %.0f (x 64 B = %.1f MB, x %.3f)" % (name, avg, avg * 64 / 1e6, avg * 64 / alg)
    print(line)
PY
