set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
for lib in r04j a; do
  rm -rf /tmp/pr_$lib
  RP_HIP_LIB=$GRAFT_REPO_ROOT/rapier_amd/librapier_hip_$lib.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_$lib -o kt -- python $GRAFT_REPO_ROOT/tools/jg_diag.py jg 300 > $OUT/r04s_$lib.log 2>&1
  d=$(find /tmp/pr_$lib -name '*.db' | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $d > $OUT/r04s_${lib}_stats.txt 2>&1
  head -14 $OUT/r04s_${lib}_stats.txt
done
