#!/bin/bash
# round 5: the register-lean island kernel (two islands per 768-thread workgroup) — parity, the island-count sweep, stage cycles
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
( RP_ISL_DENSE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_islands.py tests/test_gpu_arena.py -m gpu -x -q 2>&1 | tail -8 ) > $O/r05_lean_parity.txt
cat $O/r05_lean_parity.txt
( echo "== planner's choice (classic up to 240 islands, lean beyond)"; timeout 600 python tools/island_count_sweep.py short;
  echo "== lean only (RP_ISL_DENSE=1)"; RP_ISL_DENSE=1 timeout 600 python tools/island_count_sweep.py short ) > $O/r05_lean_sweep.txt 2>&1
cat $O/r05_lean_sweep.txt
( RP_HIP_LIB=$PWD/rapier_amd/librapier_hip_prof.so RP_ISL_DENSE=1 python tools/isl_profile.py 19 19 ) > $O/r05_lean_profile.txt 2>&1
cat $O/r05_lean_profile.txt
