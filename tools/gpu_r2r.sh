#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2r}
timeout 600 python -m pytest tests -m gpu -x -q -k "dataflow or large_pyramid or joint_grid or fuzz_pile or coulomb_multi" > $OUT/pytest_a_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_a_$TAG.log
tail -5 $OUT/pytest_a_$TAG.log | cut -c1-200
RP_NO_FLOW=1 timeout 900 python -m pytest tests -m gpu -x -q -k "large_pyramid or joint_grid or fuzz_pile or fuzz_bit_exact or tumble or kinematic or sleep" > $OUT/pytest_b_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_b_$TAG.log
tail -5 $OUT/pytest_b_$TAG.log | cut -c1-200
for sc in large_pyramid joint_grid; do
  RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 300 2>&1 | head -1 | cut -c1-70
  RP_NO_FLOW=1 RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 300 2>&1 | head -1 | cut -c1-70
done
RP_NO_FLOW=1 timeout 200 python tools/lp_steady.py 2>&1 | tail -3 | cut -c1-300
rm -rf /tmp/pr_lp
(cd /tmp && RP_NO_FLOW=1 RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_lp -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py large_pyramid 100 > $OUT/kt_lp_$TAG.log 2>&1)
d=$(find /tmp/pr_lp -name '*.db' | head -1)
[[ -n "$d" ]] && python tools/rocpd_stats.py $d > $OUT/kstats_lp_perstage_$TAG.txt 2>&1
head -24 $OUT/kstats_lp_perstage_$TAG.txt | cut -c1-150
