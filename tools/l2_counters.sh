#!/bin/bash
# What serves a tile sweep's row loads — the XCD's L2, or what lies behind it?  (first question.)
# Three rocprofv3 PMC passes over tools/prof_run.py <scene> 100 (60 warm-up + 100 steps), counters only (no trace domains besides
# --kernel-trace): L2 hits / misses, the L2's read requests to the fabric and how many of them are flagged DRAM, their sizes.
#   gpurun -- 'bash tools/l2_counters.sh large_pyramid r06'      -> gpurun_out/<tag>_<scene>_l2_counters.txt
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
SC=${1:-large_pyramid}; TAG=${2:-l2}
cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum" "TCC_EA0_RDREQ_128B_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1)); rm -rf /tmp/l2_$i
  RP_PROF_TIMERS=0 timeout 600 rocprofv3 --pmc $set --kernel-trace -d /tmp/l2_$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_run.py $SC 100 > /tmp/l2_$i.log 2>&1
done
python - "$SC" > $OUT/${TAG}_${SC}_l2_counters.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sorted(glob.glob("/tmp/l2_[0-9]*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
names = ["TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "FETCH_SIZE", "WRITE_SIZE"]
print(f"# {sys.argv[1]}: rocprofv3 PMC, average per launch (tools/l2_counters.sh; 60 warm-up + 100 steps; one pass per counter pair)")
print("kernel".ljust(44) + "launches".rjust(9) + "".join(n.replace("TCC_", "").replace("_sum", "").rjust(16) for n in names) + "   L2 hit rate   DRAM share of EA reads")
for k in sorted(acc, key=lambda k: -acc[k].get("TCC_REQ_sum", [0, 0])[0]):
    v = acc[k]
    n = max(c[1] for c in v.values())
    if n < 50: continue
    avg = {m: (v[m][0] / v[m][1] if m in v and v[m][1] else 0.0) for m in names}
    hit = avg["TCC_HIT_sum"] / max(avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"], 1.0)
    dram = avg["TCC_EA0_RDREQ_DRAM_sum"] / max(avg["TCC_EA0_RDREQ_sum"], 1.0)
    print(k[:43].ljust(44) + str(n).rjust(9) + "".join(f"{avg[m]:16.0f}" for m in names) + f"   {hit:10.3f}   {dram:10.3f}")
PY
cat $OUT/${TAG}_${SC}_l2_counters.txt | cut -c1-260 | head -14
