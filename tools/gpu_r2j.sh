#!/bin/bash
# Round-2 GPU session J: sensors, the chain-following colouring (whole suite), first-step colouring cost.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2j}
timeout 300 python -m pytest tests/test_gpu_sensors.py -m gpu -q > $OUT/pytest_sens_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_sens_$TAG.log
tail -40 $OUT/pytest_sens_$TAG.log
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_sensors.py > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -12 $OUT/pytest_$TAG.log
for sc in large_pyramid many_pyramids; do
  rm -rf /tmp/pr_$sc
  (cd /tmp && RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_$sc -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py $sc 100 > $OUT/kt_${sc}_$TAG.log 2>&1)
  d=$(find /tmp/pr_$sc -name '*.db' | head -1)
  [[ -n "$d" ]] && python tools/rocpd_stats.py $d > $OUT/kstats_${sc}_$TAG.txt 2>&1
  grep -E "k_color_pairs" $OUT/kstats_${sc}_$TAG.txt | cut -c1-150
done
