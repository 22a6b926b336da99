#!/bin/bash
# round 5: the register-lean island kernel — parity, then the island-count sweep in three configurations
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
( RP_ISL_DENSE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_islands.py -m gpu -x -q 2>&1 | tail -8 ) > $O/r05_lean_parity.txt
cat $O/r05_lean_parity.txt
( echo "== classic"; timeout 600 python tools/island_count_sweep.py short;
  echo "== lean (RP_ISL_DENSE=1)"; RP_ISL_DENSE=1 timeout 600 python tools/island_count_sweep.py short;
  echo "== lean + HSA_SCRATCH_SINGLE_LIMIT=2G"; HSA_SCRATCH_SINGLE_LIMIT=2147483648 RP_ISL_DENSE=1 timeout 600 python tools/island_count_sweep.py short;
  echo "== dense_check lean"; RP_ISL_DENSE=1 timeout 300 python tools/dense_check.py;
  echo "== dense_check lean + limit"; HSA_SCRATCH_SINGLE_LIMIT=2147483648 RP_ISL_DENSE=1 timeout 300 python tools/dense_check.py ) > $O/r05_lean_sweep.txt 2>&1
cat $O/r05_lean_sweep.txt
