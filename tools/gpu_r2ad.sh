#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_r2ad.log 2>&1; echo "rc=$?" >> $OUT/pytest_r2ad.log
tail -3 $OUT/pytest_r2ad.log | cut -c1-200
timeout 200 python tools/lp_steady.py 2>&1 | tail -3 | cut -c1-300
RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py joint_grid 300 2>&1 | head -1 | cut -c1-70
rm -rf /tmp/pr_lp
(cd /tmp && RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_lp -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py large_pyramid 100 > $OUT/kt_lp_r2ad.log 2>&1)
d=$(find /tmp/pr_lp -name '*.db' | head -1)
[[ -n "$d" ]] && python tools/rocpd_stats.py $d 2>&1 | grep -E "k_bp_rebuild|k_layout_rebuild|k_np_update" | cut -c1-150
