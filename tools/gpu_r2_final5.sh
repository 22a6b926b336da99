#!/bin/bash
# after the last kernel-source edit: global-path parity, PMC traffic record, bench lines, C2 over time + kernel stats
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=r2zz
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groups.py tests/test_gpu_fuzz.py -x -q -m gpu -k "not full_size" > $OUT/pytest_$TAG.log 2>&1; tail -2 $OUT/pytest_$TAG.log | cut -c1-200
bash tools/gpu_profile.sh many_pyramids ${TAG}_mp > $OUT/profile_mp_$TAG.log 2>&1
cp $OUT/${TAG}_mp_hbm_traffic.json profiles/many_pyramids_hbm_traffic.json 2>/dev/null
timeout 400 python bench.py > $OUT/bench_$TAG.log 2>&1; tail -1 $OUT/bench_$TAG.log | cut -c1-300
timeout 200 python tools/lp_steady.py > $OUT/lp_steady_$TAG.txt 2>&1; cat $OUT/lp_steady_$TAG.txt | cut -c1-250
rm -rf /tmp/pr_lp
(cd /tmp && RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_lp -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py large_pyramid 100 > $OUT/kt_lp_$TAG.log 2>&1)
d=$(find /tmp/pr_lp -name '*.db' | head -1)
[[ -n "$d" ]] && python tools/rocpd_stats.py $d > $OUT/kstats_large_pyramid_$TAG.txt 2>&1
head -12 $OUT/kstats_large_pyramid_$TAG.txt | cut -c1-150
