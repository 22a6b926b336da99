#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2t}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -4 $OUT/pytest_$TAG.log | cut -c1-200
timeout 200 python tools/lp_steady.py 2>&1 | tail -3 | cut -c1-300
for sc in many_pyramids; do
  RP_NO_FAST=1 RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 500 2>&1 | head -1 | cut -c1-90
done
timeout 100 python tools/sleep_awake_cost.py
timeout 900 python tools/bench_configs.py $OUT/configs_$TAG.json --quick > $OUT/configs_$TAG.log 2>&1; cat $OUT/configs_$TAG.log | cut -c1-160
