#!/bin/bash
# Round-2 evidence run (final tree): whole -m gpu suite, smoke, PMC traffic record + bench.py, C4 on one GPU, every config.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2g}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -4 $OUT/pytest_$TAG.log | cut -c1-200
timeout 300 python __graft_entry__.py smoke > $OUT/smoke_$TAG.log 2>&1; tail -1 $OUT/smoke_$TAG.log
bash tools/gpu_profile.sh many_pyramids ${TAG}_mp > $OUT/profile_mp_$TAG.log 2>&1
cp $OUT/${TAG}_mp_hbm_traffic.json profiles/many_pyramids_hbm_traffic.json 2>/dev/null
head -4 $OUT/${TAG}_mp_kernel_stats.txt | cut -c1-150
timeout 400 python bench.py > $OUT/bench_$TAG.log 2>&1; tail -1 $OUT/bench_$TAG.log | cut -c1-1300
timeout 400 python bench.py --workload c4 --gpus 1 --steps 300 --no-cpu-baseline > $OUT/bench_c4_$TAG.log 2>&1; tail -1 $OUT/bench_c4_$TAG.log | cut -c1-300
timeout 900 python tools/bench_configs.py $OUT/configs_$TAG.json > $OUT/configs_$TAG.log 2>&1; cat $OUT/configs_$TAG.log | cut -c1-200
timeout 100 python tools/sleep_awake_cost.py
