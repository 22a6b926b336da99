"""b3d_large_world (examples3d/b3d_large_world.rs, box3d's `large_world` benchmark) at its release size on the device: a 1000 x 1000
floor of parentless fixed cuboids — ONE MILLION static shapes — onto which 100 spheres are dropped, one every 5 steps.  The oracle's
sort-and-sweep broad phase cannot follow at this size (tests/test_pipeline_unit.py compares a 40 x 40 floor bit for bit); here the
checks are the size-independent ones: every sphere ends at rest on the floor, nothing overflows, nothing is non-finite.
    python tools/large_world.py [grid=1000] [steps=700]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 700
spheres = 100
t = time.perf_counter()
sc = S.large_world(grid)
t_scene = time.perf_counter() - t
t = time.perf_counter()
w = PhysicsWorld.from_scene(sc)
w.step(1); w.sync()                      # the device world is built by the first step (a million colliders into the broad phase)
t_build = time.perf_counter() - t
print(f"{grid} x {grid} = {grid * grid:,} parentless fixed cuboids: descriptors {t_scene:.2f} s, device world + first step {t_build:.2f} s")
dropped, balls, t_insert = 0, [], 0.0
t0 = time.perf_counter()
for step in range(1, steps):
    if dropped < spheres and step % 5 == 0:
        ti = time.perf_counter()
        b = w.insert_body(S.body_desc(translation=S.large_world_drop(dropped, grid, spheres=spheres), can_sleep=1))
        w.insert_collider(S.collider_desc(shape=S.SHAPE_BALL, half_extents=(0.5, 0, 0)), b)
        t_insert += time.perf_counter() - ti
        balls.append(int(b) & 0xFFFFFFFF); dropped += 1
    w.step(1)
    if step == 5 * spheres + 1:
        w.sync(); t_drop = time.perf_counter() - t0
w.sync()
dt = time.perf_counter() - t0
p, v = w.read_bodies()
c = w.counters()
assert np.isfinite(p).all() and np.isfinite(v).all() and c["overflow_flags"] == 0, c
rest = np.abs(p[balls, 1] - 0.75).max()
print(f"{steps - 1} steps with {dropped} spheres dropped (one every 5 steps): {(steps - 1) / dt:,.0f} steps/s ({dt / (steps - 1) * 1e3:.3f} ms/step; the {5 * spheres} steps of the drop "
      f"phase {t_drop / (5 * spheres) * 1e3:.3f} ms/step, of which {t_insert / dropped * 1e3:.2f} ms per insertion x {dropped})")
print(f"spheres at rest on the floor: max |y - 0.75| = {rest:.4f}, max speed {np.abs(v[balls]).max():.2e}, sleeping {c['num_sleeping_bodies']}; pairs {c['num_pairs']}, "
      f"broad-phase rebuilds {c['bp_rebuilds']}, large list {c['bp_large_list']}, fast {c['fast_steps']} full {c['full_steps']} replayed {c['replayed_steps']}")
assert rest < 0.02
