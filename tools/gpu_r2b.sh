#!/bin/bash
# Round-2 GPU session B: dataflow launch — correctness subset, steps/s + hand-off statistics, rocprofv3 kernel stats per scene.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2b}
SCENES=${2:-"large_pyramid joint_grid many_pyramids_coulomb"}
timeout 600 python -m pytest tests -m gpu -x -q -k "dataflow or large_pyramid or joint_grid or fuzz_pile" > $OUT/pytest_flow_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_flow_$TAG.log
tail -6 $OUT/pytest_flow_$TAG.log
for sc in $SCENES; do
  RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 300 > $OUT/flow_${sc}_$TAG.log 2>&1; tail -2 $OUT/flow_${sc}_$TAG.log | cut -c1-300
  rm -rf /tmp/pr_$sc
  (cd /tmp && RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_$sc -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py $sc 100 > $OUT/kt_${sc}_$TAG.log 2>&1)
  d=$(find /tmp/pr_$sc -name '*.db' | head -1)
  [[ -n "$d" ]] && python tools/rocpd_stats.py $d > $OUT/kstats_${sc}_$TAG.txt 2>&1
  head -14 $OUT/kstats_${sc}_$TAG.txt | cut -c1-140
done
