#!/bin/bash
# half-space shape: new GPU tests + the regressions closest to the touched code
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_halfspace.py -x -q -m gpu 2>&1 | tail -25 | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "halfspace or test_fuzz_bit_exact or sensors" 2>&1 | tail -15 | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or capsule or tumbl" 2>&1 | tail -8 | cut -c1-250
