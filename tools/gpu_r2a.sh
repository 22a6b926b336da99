#!/bin/bash
# Round-2 GPU session A: first contact of the dataflow launch with the hardware (sanity under a short timeout), the whole
# -m gpu suite, then steps/s of the global-path scenes with the dataflow launch and with one launch per colour stage.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2a}
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dataflow or large_pyramid_bit_exact or joint_grid_bit_exact" > $OUT/pytest_flow_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_flow_$TAG.log
tail -15 $OUT/pytest_flow_$TAG.log
for sc in large_pyramid joint_grid many_pyramids_coulomb; do
  RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 300 > $OUT/flow_${sc}_$TAG.log 2>&1; tail -1 $OUT/flow_${sc}_$TAG.log | cut -c1-400
  RP_NO_FLOW=1 RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 300 > $OUT/noflow_${sc}_$TAG.log 2>&1; tail -1 $OUT/noflow_${sc}_$TAG.log | cut -c1-400
done
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -8 $OUT/pytest_$TAG.log
