#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for wg in 1 2; do
RP_FLOW=1 RP_FLOW_WG_PER_CU=$wg RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py large_pyramid 300 2>&1 | head -2 | cut -c1-70,800-1000
done
