#!/bin/bash
# Build an instrumented twin of the library (-DRP_FLOW_TRACE) next to the product and print one body's event timeline.
set -u
cd $GRAFT_REPO_ROOT
RP_HIP_LIB=$GRAFT_REPO_ROOT/rapier_amd/librapier_hip_trace.so ${2:+RP_FLOW_DRY=1} timeout 200 python tools/flow_trace.py ${1:-large_pyramid} 120
