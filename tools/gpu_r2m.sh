#!/bin/bash
# joint rows prefetched ahead of the ticket wait (dataflow launch): joint parity + C5 timing
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "joint or motor or limit or flow" 2>&1 | tail -4 | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "test_fuzz_bit_exact or params" 2>&1 | tail -3 | cut -c1-250
python - <<'P'
import time
from rapier_amd import PhysicsWorld, scenes as S
w = PhysicsWorld.from_scene(S.joint_grid(100)); w.step(200); w.sync()
t=time.perf_counter(); w.step(500); w.sync(); dt=time.perf_counter()-t
print("C5 joint_grid: %.1f steps/s (%.3f ms)" % (500/dt, dt*2))
P
