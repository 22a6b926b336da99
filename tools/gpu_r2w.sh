#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2w}
timeout 900 python -m pytest tests -m gpu -x -q -k "dataflow or joint or coulomb or fuzz_pile or fuzz_bit_exact or motor or limit" > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -4 $OUT/pytest_$TAG.log | cut -c1-200
for sc in joint_grid many_pyramids_coulomb; do
  RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 300 2>&1 | head -2 | cut -c1-70,800-1000
done
RP_FLOW=1 RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py large_pyramid 300 2>&1 | head -2 | cut -c1-70,800-1000
for wg in 2; do
RP_FLOW_WG_PER_CU=$wg RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py joint_grid 300 2>&1 | head -2 | cut -c1-70,800-1000
done
