#!/bin/bash
# A/B of library variants on one box: tools/ab_bench.sh <tag> "<bench args>" <lib> [<lib> ...]   ("default" = rapier_amd/librapier_hip.so)
# Every variant runs the same bench.py command, interleaved over 3 rounds; one line per run in gpurun_out/<tag>_ab.txt
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT; mkdir -p gpurun_out
TAG=$1; ARGS=$2; shift 2
: > gpurun_out/${TAG}_ab.txt
for round in 1 2 3; do
  for lib in "$@"; do
    if [[ "$lib" == default ]]; then unset RP_HIP_LIB; else export RP_HIP_LIB=$ROOT/$lib; fi
    line=$(timeout 600 python bench.py $ARGS --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$lib round $round: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; print(round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],5), "kernel_ms", r.get("kernel_ms_per_launch"))' 2>&1)" >> gpurun_out/${TAG}_ab.txt
  done
done
cat gpurun_out/${TAG}_ab.txt
