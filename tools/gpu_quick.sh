#!/bin/bash
# quick GPU check: parity tests + un-profiled timing of many_pyramids
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
TAG=${1:-q}
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_$TAG.log
timeout 300 python tools/prof_run.py many_pyramids 400 2>&1 | tee $OUT/run_many_$TAG.log
