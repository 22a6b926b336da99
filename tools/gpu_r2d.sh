#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2d}
for pre in ${2:-0 2}; do
  for sc in ${3:-large_pyramid joint_grid}; do
    RP_FLOW_PRE=$pre RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 300 > $OUT/flow_${sc}_${TAG}_pre$pre.log 2>&1; echo "pre=$pre"; tail -2 $OUT/flow_${sc}_${TAG}_pre$pre.log | cut -c1-120,800-1400
  done
done
