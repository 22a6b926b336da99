#!/bin/bash
# Round-2 closing evidence: whole -m gpu suite, smoke, PMC traffic record + bench.py, C4 on one GPU, every config, large_pyramid
# over time + kernel stats, joint_grid kernel stats.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2z}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -4 $OUT/pytest_$TAG.log | cut -c1-200
timeout 300 python __graft_entry__.py smoke > $OUT/smoke_$TAG.log 2>&1; tail -1 $OUT/smoke_$TAG.log
bash tools/gpu_profile.sh many_pyramids ${TAG}_mp > $OUT/profile_mp_$TAG.log 2>&1
cp $OUT/${TAG}_mp_hbm_traffic.json profiles/many_pyramids_hbm_traffic.json 2>/dev/null
head -3 $OUT/${TAG}_mp_kernel_stats.txt | cut -c1-150
timeout 400 python bench.py > $OUT/bench_$TAG.log 2>&1; tail -1 $OUT/bench_$TAG.log | cut -c1-400
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_short_$TAG.log 2>&1; tail -1 $OUT/bench_short_$TAG.log | cut -c1-200
timeout 400 python bench.py --workload c4 --gpus 1 --steps 300 --no-cpu-baseline > $OUT/bench_c4_$TAG.log 2>&1; tail -1 $OUT/bench_c4_$TAG.log | cut -c1-200
timeout 900 python tools/bench_configs.py $OUT/configs_$TAG.json > $OUT/configs_$TAG.log 2>&1; cat $OUT/configs_$TAG.log | cut -c1-200
timeout 200 python tools/lp_steady.py > $OUT/lp_steady_$TAG.txt 2>&1; cat $OUT/lp_steady_$TAG.txt | cut -c1-250
for wl in large_pyramid joint_grid; do
  rm -rf /tmp/pr_$wl
  (cd /tmp && RP_PROF_TIMERS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_$wl -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py $wl 100 > $OUT/kt_${wl}_$TAG.log 2>&1)
  d=$(find /tmp/pr_$wl -name '*.db' | head -1)
  [[ -n "$d" ]] && python tools/rocpd_stats.py $d > $OUT/kstats_${wl}_$TAG.txt 2>&1
  head -9 $OUT/kstats_${wl}_$TAG.txt | cut -c1-150
done
