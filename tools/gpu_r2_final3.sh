#!/bin/bash
# evidence refresh on the final tree: configs table, large_pyramid over time, its kernel stats
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=r2i3
timeout 900 python tools/bench_configs.py $OUT/configs_$TAG.json > $OUT/configs_$TAG.log 2>&1; cat $OUT/configs_$TAG.log | cut -c1-200
timeout 200 python tools/lp_steady.py > $OUT/lp_steady_$TAG.txt 2>&1; cat $OUT/lp_steady_$TAG.txt | cut -c1-300
rm -rf /tmp/pr_lp
(cd /tmp && RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_lp -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py large_pyramid 100 > $OUT/kt_lp_$TAG.log 2>&1)
d=$(find /tmp/pr_lp -name '*.db' | head -1)
[[ -n "$d" ]] && python tools/rocpd_stats.py $d > $OUT/kstats_lp_$TAG.txt 2>&1
head -12 $OUT/kstats_lp_$TAG.txt | cut -c1-150
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_short_$TAG.log 2>&1; tail -1 $OUT/bench_short_$TAG.log | cut -c1-200
