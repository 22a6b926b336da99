#!/bin/bash
# Whole -m gpu suite + the default bench line (+ optional C4 on one GPU): outputs under gpurun_out/.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-s}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -6 $OUT/pytest_$TAG.log
timeout 600 python bench.py > $OUT/bench_$TAG.log 2>&1; echo "rc=$?" >> $OUT/bench_$TAG.log
tail -2 $OUT/bench_$TAG.log | cut -c1-1500
if [[ "${2:-}" == c4 ]]; then
  timeout 900 python bench.py --workload c4 --steps 300 --no-cpu-baseline > $OUT/bench_c4_$TAG.log 2>&1; echo "rc=$?" >> $OUT/bench_c4_$TAG.log
  tail -2 $OUT/bench_c4_$TAG.log | cut -c1-1200
fi
