#!/bin/bash
# Round-2 GPU session N: fused rebuild kernels (broad phase, layout): whole suite, full-step cost (RP_NO_FAST=1), first-step colouring.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2n}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -12 $OUT/pytest_$TAG.log | cut -c1-200
for sc in many_pyramids many_pyramids_events large_pyramid joint_grid; do
  RP_NO_FAST=1 RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 500 2>&1 | head -1 | cut -c1-90
done
for sc in many_pyramids joint_grid large_pyramid; do
  rm -rf /tmp/pr_$sc
  (cd /tmp && RP_NO_FAST=1 RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_$sc -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py $sc 100 > $OUT/kt_${sc}_$TAG.log 2>&1)
  d=$(find /tmp/pr_$sc -name '*.db' | head -1)
  [[ -n "$d" ]] && python tools/rocpd_stats.py $d > $OUT/kstats_full_${sc}_$TAG.txt 2>&1
  head -22 $OUT/kstats_full_${sc}_$TAG.txt | cut -c1-150
done
