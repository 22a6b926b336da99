#!/bin/bash
# rocprofv3 kernel-trace stats of one scene (no PMC): tools/gpu_kt.sh <scene> <steps> <tag>
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
SC=${1:-large_pyramid}; N=${2:-100}; TAG=${3:-kt}
cd /tmp; rm -rf /tmp/pr_kt
RP_NO_GRAPH=${RP_NO_GRAPH:-0} RP_PROF_TIMERS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pr_kt -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py $SC $N > $OUT/${TAG}_kt.log 2>&1
d=$(find /tmp/pr_kt -name '*.db' | head -1)
[[ -n "$d" ]] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $d > $OUT/${TAG}_kernel_stats.txt 2>&1
head -40 $OUT/${TAG}_kernel_stats.txt; grep steps/s $OUT/${TAG}_kt.log
