#!/bin/bash
# rocprofv3 evidence for one scene: kernel-trace stats + FETCH_SIZE / WRITE_SIZE PMC passes (separate runs).
# Usage: tools/gpu_profile.sh <scene> <tag> [steps]     outputs: gpurun_out/<tag>_*
# (steps: the kernel-trace run of large_pyramid is kept to 100 steps — rocprofv3 7.2 segfaults inside the HIP runtime once a
#  --stats run of that scene passes ~30 k kernel records; the PMC passes and shorter traces are unaffected)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
SC=${1:-many_pyramids}; TAG=${2:-prof}; N=${3:-300}
[[ $SC == large_pyramid && $# -lt 3 ]] && N=100
cd /tmp
rm -rf /tmp/pr_kt /tmp/pr_f /tmp/pr_w
RP_PROF_TIMERS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pr_kt -o kt -- python $GRAFT_REPO_ROOT/tools/prof_run.py $SC $N > $OUT/${TAG}_kt.log 2>&1
d=$(find /tmp/pr_kt -name '*.db' | head -1)
[[ -n "$d" ]] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $d > $OUT/${TAG}_kernel_stats.txt 2>&1
find /tmp/pr_kt -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
RP_PROF_COUNTERS=/tmp/pr_counters.json RP_PROF_TIMERS=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pr_f -o f --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_run.py $SC 100 > $OUT/${TAG}_pmc_fetch.log 2>&1
RP_PROF_TIMERS=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pr_w -o w --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_run.py $SC 100 > $OUT/${TAG}_pmc_write.log 2>&1
WL=c3; [[ $SC == large_pyramid ]] && WL=large_pyramid; [[ $SC == joint_grid ]] && WL=joint_grid
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pr_f /tmp/pr_w $WL 160 /tmp/pr_counters.json > $OUT/${TAG}_hbm_traffic.json 2> $OUT/${TAG}_pmc_summary.err   # (prof_run.py: 60 warm-up + 100 steps)
cat $OUT/${TAG}_hbm_traffic.json
head -12 $OUT/${TAG}_kernel_stats.txt
tail -2 $OUT/${TAG}_kt.log
