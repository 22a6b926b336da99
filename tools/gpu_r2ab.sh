#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for k in 1 2 3 4 5 6 7 8 9 10 11 12; do
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "concurrently" 2>&1 | tail -1 | cut -c1-120
done
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_r2ab.log 2>&1; echo "rc=$?" >> $OUT/pytest_r2ab.log
tail -3 $OUT/pytest_r2ab.log | cut -c1-200
