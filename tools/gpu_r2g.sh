#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2g}
timeout 600 python -m pytest tests -m gpu -x -q -k "dataflow or large_pyramid_bit_exact or joint_grid_bit_exact or fuzz_pile" > $OUT/pytest_flow_${TAG}.log 2>&1; echo "rc=$?" >> $OUT/pytest_flow_${TAG}.log
tail -2 $OUT/pytest_flow_${TAG}.log
for cfg in "1 1" "1 2" "1 4" "2 1" "2 2" "2 4"; do
  set -- $cfg
  for sc in large_pyramid joint_grid; do
    RP_FLOW_WG_PER_CU=$1 RP_FLOW_LANE_STRIDE=$2 RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 300 > $OUT/flow_${sc}_${TAG}_w$1_s$2.log 2>&1; echo "wg/cu=$1 stride=$2"; tail -2 $OUT/flow_${sc}_${TAG}_w$1_s$2.log | cut -c1-60,800-1200
  done
done
