#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel-trace stats pass, then one rocprofv3 --pmc pass per
# counter group (never combined with a trace domain), then tools/summarize_prof.py writes the
# summaries under gpurun_out/prof/ -- copy the ones to keep into profiles/.
#   usage: tools/profile.sh <tag> "<counter group 1>" "<counter group 2>" ...
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}; shift
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the profiled command = the bench command itself (its own HIP-event timers off, no CPU baseline leg)
CMD=${PROF_CMD:-"python $REPO/bench.py --gpus 1 --no-cpu-baseline --no-kernel-timing"}
export PROF_CMD_STR="$CMD"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $CMD > $OUT/${TAG}_stats.log 2>&1
i=0
for grp in "$@"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/prof_pmc_$i -- $CMD > $OUT/${TAG}_pmc_$i.log 2>&1
done
python $REPO/tools/summarize_prof.py $TAG /tmp/prof_stats $OUT /tmp/prof_pmc_*
