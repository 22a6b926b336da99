#!/bin/bash
# builds rapier_amd/librapier_hip_tprof.so: the library with -DRP_TILE_PROFILE tile sweeps (per-phase wall-clock stamps of tile 0,
# read by tools/tile_diag.py through RP_HIP_LIB)
set -e
cd "$(dirname "$0")/../rapier_amd/csrc"
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -DRP_TILE_PROFILE -c rp_tiles.hip -o /tmp/rp_tiles_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librapier_hip_tprof.so rp_api.o rp_broadphase.o rp_narrowphase.o rp_solver.o rp_islands.o rp_islands_lean.o rp_joints.o rp_sleep.o rp_flow.o /tmp/rp_tiles_prof.o
echo built
