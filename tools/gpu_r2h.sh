#!/bin/bash
# Round-2 GPU session H (state of the tree after the container restart): dataflow correctness subset, steps/s of the global-path
# scenes with the dataflow launch and with per-stage launches, lp_steady, bench.py, rocprofv3 kernel stats for large_pyramid.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2h}
timeout 420 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dataflow or large_pyramid_bit_exact or joint_grid_bit_exact" > $OUT/pytest_flow_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_flow_$TAG.log
tail -4 $OUT/pytest_flow_$TAG.log
for sc in large_pyramid joint_grid many_pyramids_coulomb; do
  RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 300 > $OUT/flow_${sc}_$TAG.log 2>&1; tail -2 $OUT/flow_${sc}_$TAG.log | cut -c1-120,800-1300
  RP_NO_FLOW=1 RP_PROF_TIMERS=0 timeout 200 python tools/prof_run.py $sc 300 > $OUT/noflow_${sc}_$TAG.log 2>&1; tail -1 $OUT/noflow_${sc}_$TAG.log | cut -c1-120
done
timeout 200 python tools/lp_steady.py > $OUT/lp_steady_$TAG.log 2>&1; tail -3 $OUT/lp_steady_$TAG.log | cut -c1-400
timeout 300 python bench.py > $OUT/bench_$TAG.log 2>&1; tail -1 $OUT/bench_$TAG.log
for sc in large_pyramid joint_grid; do
  rm -rf /tmp/pr_$sc
  (cd /tmp && RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_$sc -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py $sc 100 > $OUT/kt_${sc}_$TAG.log 2>&1)
  d=$(find /tmp/pr_$sc -name '*.db' | head -1)
  [[ -n "$d" ]] && python tools/rocpd_stats.py $d > $OUT/kstats_${sc}_$TAG.txt 2>&1
  head -16 $OUT/kstats_${sc}_$TAG.txt | cut -c1-150
done
