#!/bin/bash
# PMC traffic record + bench line for the final kernel sources (a comment changed: the stamp is a hash of the files)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash tools/gpu_profile.sh many_pyramids r2end_mp > $OUT/profile_mp_r2end.log 2>&1
cp $OUT/r2end_mp_hbm_traffic.json profiles/many_pyramids_hbm_traffic.json 2>/dev/null
timeout 300 python bench.py > $OUT/bench_r2end.log 2>&1; tail -1 $OUT/bench_r2end.log | cut -c1-300
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coulomb or generic" 2>&1 | tail -2
