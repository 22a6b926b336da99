#!/bin/bash
# A/B of ENVIRONMENT settings on one box with one library: tools/ab_env.sh <tag> "<bench args>" "<env A or ->" "<env B>" ...   (3 interleaved rounds)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT; mkdir -p gpurun_out
TAG=$1; ARGS=$2; shift 2
: > gpurun_out/${TAG}_ab.txt
for round in 1 2 3; do
  for e in "$@"; do
    if [[ "$e" == "-" ]]; then line=$(timeout 600 python bench.py $ARGS --no-cpu-baseline 2>/dev/null | tail -1); else line=$(env $e timeout 600 python bench.py $ARGS --no-cpu-baseline 2>/dev/null | tail -1); fi
    echo "[$e] round $round: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; print(round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],5), "kernel_ms", r.get("kernel_ms_per_launch"), r.get("path"))' 2>&1)" >> gpurun_out/${TAG}_ab.txt
  done
done
cat gpurun_out/${TAG}_ab.txt
