#!/bin/bash
# generic island kernel: Coulomb worlds on islands; the twist model forced through it (RP_ISL_GENERIC=1); timing
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coulomb or golden" 2>&1 | tail -12 | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "test_fuzz_bit_exact or test_fuzz_params or halfspace" 2>&1 | tail -12 | cut -c1-250
RP_ISL_GENERIC=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pyramid or golden or stack or tumbl or sleep" 2>&1 | tail -12 | cut -c1-250
python - <<'P'
import time, numpy as np
from rapier_amd import PhysicsWorld, scenes as S
sc = S.many_pyramids(); sc.params["friction_model"] = S.FRICTION_COULOMB
w = PhysicsWorld.from_scene(sc); w.step(300); w.sync()
t=time.perf_counter(); w.step(1000); w.sync(); dt=time.perf_counter()-t
print("C3 coulomb: %.1f steps/s (%.3f ms)" % (1000/dt, dt), {k: w.counters()[k] for k in ("fast_steps","full_steps","replayed_steps","num_islands")} )
P
