#!/bin/bash
# bench + un-profiled timing of the feature variants of the headline scene
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
TAG=${1:-v}
timeout 400 python bench.py > $OUT/bench_$TAG.log 2>&1; tail -1 $OUT/bench_$TAG.log | cut -c1-900
for sc in many_pyramids_sleep many_pyramids_coulomb many_pyramids_events large_pyramid joint_grid; do
  RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 500 2>&1 | cut -c1-400 | tee -a $OUT/variants_$TAG.log
done
RP_NO_FAST=1 RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py many_pyramids 500 2>&1 | cut -c1-200 | sed 's/^/[RP_NO_FAST] /' | tee -a $OUT/variants_$TAG.log
