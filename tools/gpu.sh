#!/bin/bash
# One parameterised GPU-box session (replaces the per-experiment driver scripts of rounds 1-2).
#   gpurun --timeout T -- 'bash tools/gpu.sh <tag> <job> [<job> ...]'
# jobs:  tests[=<pytest -k expr>]   pytest -m gpu (-x), log gpurun_out/<tag>_pytest.log
#        file=<tests/file.py>       one test file
#        smoke                      __graft_entry__.smoke()
#        bench[=<bench.py args>]    bench.py, JSON line in gpurun_out/<tag>_bench.json
#        forcedist                  bench.py --gpus 1 --force-dist (the nccl leg on one rank)
#        prof=<scene>               rocprofv3 kernel-trace stats + PMC FETCH/WRITE passes (tools/gpu_profile.sh)
#        configs                    tools/bench_configs.py (every single-GPU config, table in gpurun_out/<tag>_configs.json)
#        py=<script and args>       python <script ...>, log gpurun_out/<tag>_py.log
#        env:NAME=VALUE             export an environment variable for the jobs that follow
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
TAG=${1:-run}; shift || true
n=0
for job in "$@"; do
  n=$((n + 1))
  case "$job" in
    env:*) export "${job#env:}" ;;
    tests|tests=*)
      k="${job#tests}"; k="${k#=}"
      if [[ -n "$k" ]]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$k" > $OUT/${TAG}_pytest_$n.log 2>&1; else timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_$n.log 2>&1; fi
      echo "pytest rc=$?" >> $OUT/${TAG}_pytest_$n.log; tail -15 $OUT/${TAG}_pytest_$n.log ;;
    file=*)
      timeout 1500 python -m pytest "${job#file=}" -m gpu -x -q > $OUT/${TAG}_pytest_$n.log 2>&1
      echo "pytest rc=$?" >> $OUT/${TAG}_pytest_$n.log; tail -25 $OUT/${TAG}_pytest_$n.log ;;
    smoke) timeout 600 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/${TAG}_smoke.log; tail -3 $OUT/${TAG}_smoke.log ;;
    bench|bench=*)
      a="${job#bench}"; a="${a#=}"
      timeout 900 python bench.py $a > $OUT/${TAG}_bench_$n.json 2> $OUT/${TAG}_bench_$n.err; echo "bench rc=$?" >> $OUT/${TAG}_bench_$n.err
      tail -c 1800 $OUT/${TAG}_bench_$n.json; tail -3 $OUT/${TAG}_bench_$n.err ;;
    forcedist)
      timeout 900 python bench.py --gpus 1 --force-dist --steps 300 --warmup 60 --no-cpu-baseline > $OUT/${TAG}_forcedist.json 2> $OUT/${TAG}_forcedist.err; echo "forcedist rc=$?" >> $OUT/${TAG}_forcedist.err
      tail -c 1200 $OUT/${TAG}_forcedist.json; tail -5 $OUT/${TAG}_forcedist.err ;;
    prof=*) bash tools/gpu_profile.sh "${job#prof=}" ${TAG}_${job#prof=} ;;
    configs) timeout 1500 python tools/bench_configs.py > $OUT/${TAG}_configs.json 2> $OUT/${TAG}_configs.err; tail -30 $OUT/${TAG}_configs.json ;;
    py=*) timeout 1500 python ${job#py=} > $OUT/${TAG}_py_$n.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_py_$n.log; tail -40 $OUT/${TAG}_py_$n.log ;;
    *) echo "unknown job $job" ;;
  esac
done
