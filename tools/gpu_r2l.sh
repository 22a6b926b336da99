#!/bin/bash
# rows preloaded before the first store in the global path's constraint functions: parity of every global-path form + timings
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groups.py -x -q -m gpu -k "not full_size" 2>&1 | tail -6 | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-250
timeout 200 python tools/lp_steady.py 2>&1 | cut -c1-300
python tools/coulomb_rate.py 2>&1 | head -1
python - <<'P'
import time
from rapier_amd import PhysicsWorld, scenes as S
w = PhysicsWorld.from_scene(S.joint_grid(100)); w.step(200); w.sync()
t=time.perf_counter(); w.step(500); w.sync(); dt=time.perf_counter()-t
print("C5 joint_grid: %.1f steps/s (%.3f ms)" % (500/dt, dt*2))
P
