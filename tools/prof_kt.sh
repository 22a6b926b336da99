#!/bin/bash
# rocprofv3 kernel-trace stats of one tools/prof_run.py scene (no PMC passes).  usage: tools/prof_kt.sh <scene> <tag> [steps]
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
SC=$1; TAG=$2; N=${3:-200}
cd /tmp; rm -rf /tmp/pr_kt
RP_PROF_TIMERS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pr_kt -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py $SC $N > $OUT/${TAG}_kt.log 2>&1
d=$(find /tmp/pr_kt -name '*.db' | head -1)
[[ -n "$d" ]] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $d > $OUT/${TAG}_kernel_stats.txt 2>&1
head -40 $OUT/${TAG}_kernel_stats.txt; tail -2 $OUT/${TAG}_kt.log
