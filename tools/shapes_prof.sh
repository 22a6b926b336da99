#!/bin/bash
# rocprofv3 kernel stats of the shapes_rain scene (tools/shapes_bench.py's world, 150 steps: falling + landing + the start of settling)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd /tmp; rm -rf /tmp/pr_sh
RP_PROF_TIMERS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pr_sh -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py shapes_rain ${1:-150} > $OUT/shapes_kt.log 2>&1
d=$(find /tmp/pr_sh -name '*.db' | head -1)
[[ -n "$d" ]] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $d > $OUT/shapes_kernel_stats.txt 2>&1
head -24 $OUT/shapes_kernel_stats.txt; tail -3 $OUT/shapes_kt.log
