"""The N = 8 leg of bench.py, rank by rank on ONE GPU (no process group): the whole C4 scene is built and stepped once, its proximity
groups come from the device, whole groups are bin-packed over 8 ranks, and every rank's guarded sub-world is stepped and timed in
turn.  Nothing here is a multi-GPU measurement: it shows that every rank's shard builds, steps with a quiet guard, and how long the
slowest one takes — bench.py's `value` at N = 8 is (cuboids of all ranks / 10,780) * steps / that time when the ranks run side by side.
    python tools/shard_dryrun.py [world=8] [steps=300] [warmup=120] [ranks to run: all]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from rapier_amd import PhysicsWorld, scenes as S, sharding  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 120
only = int(sys.argv[4]) if len(sys.argv) > 4 else world   # (a quick look: the first few ranks only)
rows, cols = sharding.C4_GRIDS.get(world, (54, 54))
t = time.perf_counter()
full = S.many_pyramids(rows, cols)
wf = PhysicsWorld.from_scene(full, 0)
wf.step(1)
groups = wf.proximity_groups()
wf.close()
body_rank, n_groups = sharding.shards_from_groups(groups, world)
print(f"{rows}x{cols} pyramids = {rows * cols * 55:,} cuboids: {n_groups} proximity groups from the device in {time.perf_counter() - t:.1f} s")
worst, total = 0.0, 0
for rank in range(min(world, only)):
    scene, gids = sharding.partition_scene(full, body_rank, rank)
    guard = sharding.guard_boxes(full, groups, body_rank, rank)
    w = PhysicsWorld.from_scene(scene, 0)
    w.set_shard_guard(*guard)
    w.step(warmup); w.sync()
    t0 = time.perf_counter()
    w.step(steps); w.sync()
    dt = time.perf_counter() - t0
    c = w.counters()
    pos, vel = w.read_bodies()
    assert np.isfinite(pos).all() and np.isfinite(vel).all() and c["overflow_flags"] == 0
    worst = max(worst, dt); total += c["num_dynamic_bodies"]
    print(f"rank {rank}: {c['num_dynamic_bodies'] // 55} islands, {len(guard[0])} foreign boxes guarded, {steps / dt:,.0f} steps/s ({dt / steps * 1e3:.3f} ms/step), "
          f"fast {c['fast_steps']} full {c['full_steps']} replayed {c['replayed_steps']}")
    w.close()
print(f"slowest rank {worst / steps * 1e3:.3f} ms/step -> if the {world} ranks ran side by side: {total / 10780 * steps / worst:,.0f} C3-equivalent steps/s "
      f"({steps / worst:,.0f} steps/s of the sharded world; no collective in the timed region)")
