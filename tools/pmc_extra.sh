#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): extra rocprofv3 --pmc passes of the bench command (one counter group per pass, no trace
# domains), condensed by tools/summarize_prof.py into gpurun_out/prof/<tag>_pmc.csv.
#   usage: tools/pmc_extra.sh <tag> "<counter group 1>" "<counter group 2>" ...
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-extra}; shift
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD=${PROF_CMD:-"python $REPO/bench.py --gpus 1 --no-cpu-baseline --no-kernel-timing --sustained 0 --dense 0"}
export PROF_CMD_STR="$CMD"
i=0
for grp in "$@"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/px_${TAG}_$i -- $CMD > $OUT/${TAG}_pmc_$i.log 2>&1
    tail -2 $OUT/${TAG}_pmc_$i.log
done
mkdir -p /tmp/px_none
python $REPO/tools/summarize_prof.py $TAG /tmp/px_none $OUT /tmp/px_${TAG}_* > /dev/null
grep -E "env_step|learn_q_pair|trace_lane|reset_kernel|accumulate|apply|memo" $OUT/${TAG}_pmc.csv
