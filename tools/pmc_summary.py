"""Summarise the FETCH_SIZE / WRITE_SIZE rocprofv3 PMC passes into HBM bytes per launch per kernel.

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE (KiB) reports half of the bytes of wide
coalesced reads -> doubled; WRITE_SIZE (KiB) taken as is.  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                a = acc[row["Kernel_Name"].split("(")[0]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    return acc


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {"unit": "bytes per launch", "correction": "hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE under-count)", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f = fetch[k][0] / max(fetch[k][1], 1) if k in fetch else 0.0
    w = write[k][0] / max(write[k][1], 1) if k in write else 0.0
    out["kernels"][k] = {"launches_fetch_pass": fetch[k][1] if k in fetch else 0, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes": (2.0 * f + w) * 1024.0}
# the island kernel: one fused step per launch (k_island_solve) or up to 32 (k_island_solve_steps, round 6) — bytes per STEP from the
# fused steps the profiled run counted (argv[5]: the counters tools/prof_run.py wrote), bytes per launch of the 32-step form beside it
isl_total = sum(out["kernels"][k]["hbm_bytes"] * out["kernels"][k]["launches_fetch_pass"] for k in ("k_island_solve", "k_island_solve_steps") if k in out["kernels"])
counters = json.load(open(sys.argv[5])) if len(sys.argv) > 5 and os.path.exists(sys.argv[5]) else {}
fused_steps = int(counters.get("fused_steps", 0))
out["run_counters"] = {k: counters.get(k) for k in ("fused_steps", "fused_launches", "fast_steps", "full_steps", "replayed_steps")}
out["k_island_solve_hbm_bytes_per_step"] = isl_total / fused_steps if fused_steps > 0 else None
isl = out["kernels"].get("k_island_solve")
out["k_island_solve_hbm_bytes_per_launch"] = isl["hbm_bytes"] if isl else None   # (the single-step launches of the run: warm-up tails)
# stamp: bench.py refuses this record once the kernel sources change (a stale traffic figure is worse than none)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
# argv[3] = the bench workload the passes ran (c3 | large_pyramid | joint_grid), argv[4] = steps the profiled run made (warm-up included):
# the global-path scenes get the bytes of every solver-loop kernel summed per step
workload = sys.argv[3] if len(sys.argv) > 3 else "c3"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if steps > 0:
    tot = 0.0
    for k, v in out["kernels"].items():
        if k.startswith(bench.SOLVER_LOOP_KERNELS) or any(k.startswith("void " + p) for p in bench.SOLVER_LOOP_KERNELS):
            tot += v["hbm_bytes"] * v["launches_fetch_pass"]
    out["solver_loop_hbm_bytes_per_step"] = tot / steps
    out["steps_profiled"] = steps
out["workload"] = workload
out["kernel_code_sha"] = bench.kernel_code_sha(workload)
print(json.dumps(out, indent=1))
