#!/bin/bash
# One GPU-box session: parity tests, bench, rocprof kernel stats. Outputs under gpurun_out/.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-a}
WHAT=${2:-all}
if [[ $WHAT == all || $WHAT == *test* ]]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_$TAG.log
  tail -5 $OUT/pytest_$TAG.log
fi
if [[ $WHAT == all || $WHAT == *bench* ]]; then
  timeout 600 python bench.py > $OUT/bench_$TAG.log 2>&1; echo "bench rc=$?" >> $OUT/bench_$TAG.log
  tail -3 $OUT/bench_$TAG.log
fi
if [[ $WHAT == all || $WHAT == *prof* ]]; then
  for sc in many_pyramids large_pyramid; do
    timeout 300 python tools/prof_run.py $sc 200 > $OUT/run_${sc}_$TAG.log 2>&1
    tail -2 $OUT/run_${sc}_$TAG.log
    rm -rf /tmp/prof_$sc
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$sc -o $sc -- python $GRAFT_REPO_ROOT/tools/prof_run.py $sc 100 > $OUT/rocprof_${sc}_$TAG.log 2>&1)
    f=$(find /tmp/prof_$sc -name '*kernel_stats.csv' | head -1)
    [[ -n "$f" ]] && cp $f $OUT/${sc}_kernel_stats_$TAG.csv
    d=$(find /tmp/prof_$sc -name '*.db' | head -1)
    [[ -n "$d" ]] && python tools/rocpd_stats.py $d > $OUT/${sc}_kernel_stats_$TAG.txt 2>&1
    ls -R /tmp/prof_$sc | head -20 >> $OUT/rocprof_${sc}_$TAG.log
    head -40 $OUT/${sc}_kernel_stats_$TAG.txt
  done
fi
