#!/bin/bash
# sleeping: step-by-step divergence report, then the whole GPU parity suite
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
TAG=${1:-s}
timeout 300 python tools/gpu_feature_diag.py > $OUT/sleep_diag_$TAG.log 2>&1; echo "diag rc=$?" >> $OUT/sleep_diag_$TAG.log
tail -40 $OUT/sleep_diag_$TAG.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee $OUT/pytest_$TAG.log
