#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_r2ac.log 2>&1; echo "rc=$?" >> $OUT/pytest_r2ac.log
tail -3 $OUT/pytest_r2ac.log | cut -c1-200
timeout 200 python tools/lp_steady.py 2>&1 | tail -3 | cut -c1-300
RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py large_pyramid 300 2>&1 | head -1 | cut -c1-70
