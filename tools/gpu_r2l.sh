#!/bin/bash
# Round-2 GPU session L: CCD counter tests, bench.py (headline), PMC traffic record for the stamped roofline, C4 on one GPU, configs table.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2l}
timeout 300 python -m pytest tests/test_gpu_ccd_flag.py -m gpu -q -s > $OUT/pytest_ccd_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_ccd_$TAG.log
tail -12 $OUT/pytest_ccd_$TAG.log | cut -c1-200
bash tools/gpu_profile.sh many_pyramids ${TAG}_mp > $OUT/profile_mp_$TAG.log 2>&1
cp $OUT/${TAG}_mp_hbm_traffic.json profiles/many_pyramids_hbm_traffic.json 2>/dev/null
head -14 $OUT/${TAG}_mp_kernel_stats.txt | cut -c1-150
timeout 400 python bench.py > $OUT/bench_$TAG.log 2>&1; tail -1 $OUT/bench_$TAG.log
timeout 400 python bench.py --workload c4 --gpus 1 --steps 300 --no-cpu-baseline > $OUT/bench_c4_$TAG.log 2>&1; tail -1 $OUT/bench_c4_$TAG.log
timeout 900 python tools/bench_configs.py $OUT/configs_$TAG.json > $OUT/configs_$TAG.log 2>&1; cat $OUT/configs_$TAG.log | cut -c1-200
