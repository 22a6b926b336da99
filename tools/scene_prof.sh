#!/bin/bash
# rocprofv3 kernel stats of one prof_run.py scene: tools/scene_prof.sh <scene> <steps>  ->  gpurun_out/<scene>_kernel_stats.txt
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
SC=$1; N=${2:-100}
cd /tmp; rm -rf /tmp/pr_sc
RP_PROF_TIMERS=0 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pr_sc -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py $SC $N > $OUT/${SC}_kt.log 2>&1
d=$(find /tmp/pr_sc -name '*.db' | head -1)
[[ -n "$d" ]] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $d > $OUT/${SC}_kernel_stats.txt 2>&1
head -20 $OUT/${SC}_kernel_stats.txt; tail -3 $OUT/${SC}_kt.log | head -2
