#!/bin/bash
# builds rapier_amd/librapier_hip.so (plain) and rapier_amd/librapier_hip_prof.so (-DRP_ISL_PROFILE island kernels: cycle stamps per stage,
# read by tools/isl_profile.py)
set -e
cd "$(dirname "$0")/../rapier_amd/csrc"
make -s
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950"
/opt/rocm/bin/hipcc $FLAGS -DRP_ISL_PROFILE -c rp_islands.hip -o /tmp/rp_islands_prof.o
/opt/rocm/bin/hipcc $FLAGS -mllvm -disable-machine-licm -DRP_ISL_PROFILE -c rp_islands_lean.hip -o /tmp/rp_islands_lean_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librapier_hip_prof.so rp_api.o rp_broadphase.o rp_narrowphase.o rp_solver.o /tmp/rp_islands_prof.o /tmp/rp_islands_lean_prof.o rp_joints.o rp_sleep.o rp_flow.o rp_tiles.o
echo built
