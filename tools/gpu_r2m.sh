#!/bin/bash
# Round-2 GPU session M: whole suite (joint-colouring wavefront, force events on the fast graph, cheaper CCD criterion), headline kernel time,
# joint_grid first-step colouring, events config.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2m}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -12 $OUT/pytest_$TAG.log | cut -c1-200
for sc in many_pyramids joint_grid many_pyramids_events; do
  rm -rf /tmp/pr_$sc
  (cd /tmp && RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_$sc -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py $sc 300 > $OUT/kt_${sc}_$TAG.log 2>&1)
  d=$(find /tmp/pr_$sc -name '*.db' | head -1)
  [[ -n "$d" ]] && python tools/rocpd_stats.py $d > $OUT/kstats_${sc}_$TAG.txt 2>&1
  head -8 $OUT/kstats_${sc}_$TAG.txt | cut -c1-150
  RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 1000 2>&1 | head -1 | cut -c1-80
done
