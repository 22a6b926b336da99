#!/bin/bash
export TMPDIR=/tmp
cd /tmp
for v in "1 large_pyramid 100" "0 large_pyramid 30" "0 large_pyramid 60"; do
    set -- $v
    rm -rf /tmp/pr_dbg
    RP_DEBUG=1 RP_NO_GRAPH=$1 RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace -d /tmp/pr_dbg -o x -- python $GRAFT_REPO_ROOT/tools/prof_run.py $2 $3 > /tmp/dbg.log 2>&1
    echo "nograph=$1 scene=[$2 $3] rc=$? $(grep -c steps/s /tmp/dbg.log) $(grep -m1 -o 'SIGSEGV.*' /tmp/dbg.log | head -c 40)"
    grep "RPDBG" /tmp/dbg.log | tail -4
done
