#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tools/color_profile.py 2>&1 | tail -2 | cut -c1-330
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_r2aa.log 2>&1; echo "rc=$?" >> $OUT/pytest_r2aa.log
tail -3 $OUT/pytest_r2aa.log | cut -c1-200
rm -rf /tmp/pr_jg
(cd /tmp && RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_jg -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py joint_grid 50 > $OUT/kt_jg_r2aa.log 2>&1)
d=$(find /tmp/pr_jg -name '*.db' | head -1)
[[ -n "$d" ]] && python tools/rocpd_stats.py $d 2>&1 | grep -E "k_joint_color\(" | cut -c1-150
