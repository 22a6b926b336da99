#!/bin/bash
# Round-2 evidence run: the whole -m gpu suite, smoke, bench.py (with the freshly measured PMC traffic record), C4 on one GPU, every
# config, rocprofv3 kernel stats + PMC passes for the headline scene and for b3d_large_pyramid / b3d_joint_grid (global path).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2f}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -4 $OUT/pytest_$TAG.log | cut -c1-200
timeout 300 python __graft_entry__.py smoke > $OUT/smoke_$TAG.log 2>&1; tail -1 $OUT/smoke_$TAG.log
bash tools/gpu_profile.sh many_pyramids ${TAG}_mp > $OUT/profile_mp_$TAG.log 2>&1
cp $OUT/${TAG}_mp_hbm_traffic.json profiles/many_pyramids_hbm_traffic.json 2>/dev/null
head -5 $OUT/${TAG}_mp_kernel_stats.txt | cut -c1-150
timeout 400 python bench.py > $OUT/bench_$TAG.log 2>&1; tail -1 $OUT/bench_$TAG.log | cut -c1-1500
timeout 400 python bench.py --workload c4 --gpus 1 --steps 300 --no-cpu-baseline > $OUT/bench_c4_$TAG.log 2>&1; tail -1 $OUT/bench_c4_$TAG.log | cut -c1-300
timeout 900 python tools/bench_configs.py $OUT/configs_$TAG.json > $OUT/configs_$TAG.log 2>&1; cat $OUT/configs_$TAG.log | cut -c1-200
timeout 100 python tools/sleep_awake_cost.py
for sc in large_pyramid joint_grid; do
  bash tools/gpu_profile.sh $sc ${TAG}_$sc > $OUT/profile_${sc}_$TAG.log 2>&1
  head -12 $OUT/${TAG}_${sc}_kernel_stats.txt | cut -c1-150
  python - <<PY
import json
d=json.load(open("$OUT/${TAG}_${sc}_hbm_traffic.json"))
ks=sorted(d["kernels"].items(), key=lambda kv:-kv[1]["hbm_bytes"])[:6]
for k,v in ks: print("  PMC", k[:50], "%.1f MB/launch" % (v["hbm_bytes"]/1e6))
PY
done
