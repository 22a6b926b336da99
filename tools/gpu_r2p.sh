#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2p}
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "solve_groups or sensors" > $OUT/pytest_fz_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_fz_$TAG.log
tail -30 $OUT/pytest_fz_$TAG.log | cut -c1-250
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fuzz.py::test_fuzz_solve_groups_bit_exact --deselect tests/test_gpu_fuzz.py::test_fuzz_sensors_bit_exact > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -6 $OUT/pytest_$TAG.log | cut -c1-200
timeout 100 python tools/sleep_awake_cost.py
