#!/bin/bash
# Round-2 GPU session I: the new tests first (quarantine, event caps), then the whole -m gpu suite, then first-step colouring cost.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2i}
timeout 300 python -m pytest tests/test_gpu_quarantine.py -m gpu -q > $OUT/pytest_quar_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_quar_$TAG.log
tail -25 $OUT/pytest_quar_$TAG.log
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 --deselect tests/test_gpu_quarantine.py > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -30 $OUT/pytest_$TAG.log
for sc in large_pyramid many_pyramids joint_grid; do
  rm -rf /tmp/pr_$sc
  (cd /tmp && RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_$sc -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py $sc 100 > $OUT/kt_${sc}_$TAG.log 2>&1)
  d=$(find /tmp/pr_$sc -name '*.db' | head -1)
  [[ -n "$d" ]] && python tools/rocpd_stats.py $d > $OUT/kstats_${sc}_$TAG.txt 2>&1
  grep -E "k_color_pairs|k_joint_color\(|k_isl_union|k_island_solve" $OUT/kstats_${sc}_$TAG.txt | cut -c1-150
done
