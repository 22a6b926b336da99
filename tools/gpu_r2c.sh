#!/bin/bash
# Round-2 GPU session C: dataflow launch variants (RP_FLOW_PRE = 0 / 1 / 2): correctness subset once per variant, steps/s + hand-off stats.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2c}
for pre in 0 1 2; do
  RP_FLOW_PRE=$pre timeout 600 python -m pytest tests -m gpu -x -q -k "dataflow or large_pyramid_bit_exact or joint_grid_bit_exact or fuzz_pile" > $OUT/pytest_flow_${TAG}_pre$pre.log 2>&1; echo "rc=$?" >> $OUT/pytest_flow_${TAG}_pre$pre.log
  tail -2 $OUT/pytest_flow_${TAG}_pre$pre.log
  for sc in large_pyramid joint_grid many_pyramids_coulomb; do
    RP_FLOW_PRE=$pre RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 300 > $OUT/flow_${sc}_${TAG}_pre$pre.log 2>&1; echo "pre=$pre"; tail -2 $OUT/flow_${sc}_${TAG}_pre$pre.log | cut -c1-200
  done
done
