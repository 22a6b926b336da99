#!/bin/bash
# island-kernel iteration: parity tests + timing with the plain build, then the cycle-stamp breakdown
# with the -DRP_ISL_PROFILE build (rapier_amd/librapier_hip_prof.so, swapped in temporarily)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/prof_run.py many_pyramids 400 2>&1 | tail -2
if [[ -f rapier_amd/librapier_hip_prof.so ]]; then
  cp rapier_amd/librapier_hip.so /tmp/plain.so; cp rapier_amd/librapier_hip_prof.so rapier_amd/librapier_hip.so
  timeout 300 python tools/isl_profile.py 2>&1 | tail -12
  cp /tmp/plain.so rapier_amd/librapier_hip.so
fi
