#!/bin/bash
# Round-2 GPU session K: substep solve-groups + sensors, then the whole suite.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2k}
timeout 400 python -m pytest tests/test_gpu_groups.py tests/test_gpu_sensors.py -m gpu -q > $OUT/pytest_grp_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_grp_$TAG.log
tail -60 $OUT/pytest_grp_$TAG.log | cut -c1-220
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_groups.py --deselect tests/test_gpu_sensors.py > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -8 $OUT/pytest_$TAG.log
