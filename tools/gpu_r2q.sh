#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2q}
timeout 900 python -m pytest tests -m gpu -x -q -k "sleep or fuzz_bit_exact or wake" > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -4 $OUT/pytest_$TAG.log | cut -c1-200
timeout 100 python tools/sleep_awake_cost.py
timeout 900 python tools/bench_configs.py $OUT/configs_$TAG.json > $OUT/configs_$TAG.log 2>&1; cat $OUT/configs_$TAG.log | cut -c1-200
