"""The widened shape row, measured: S.shapes_rain — n bodies of all ten shape kinds landing on a slab — on the device and
on the oracle: steps/s per phase (falling: few pairs; landing: thousands of GJK / EPA manifold updates per step; resting: recycled),
the narrow-phase share from the stage counters, and the state after the run compared with the oracle's bit for bit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402
import oracle_ffi  # noqa: E402
from oracle_ffi import OracleWorld  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
sc = S.shapes_rain(n)
t = time.perf_counter(); g = PhysicsWorld.from_scene(sc); g.step(1); g.sync(); t_build = time.perf_counter() - t
oracle_ffi.set_threads(max(1, min(os.cpu_count() or 1, 32)))
o = OracleWorld(sc); o.step(1)
print(f"shapes_rain: {n} bodies ({len(sc.colliders)} colliders, {len(sc.polyhedra)} polyhedra); device world + first step {t_build:.2f} s")
print("phase (steps)        GPU steps/s   ms/step   oracle steps/s (32 thr)   manifolds   full updates/step   bit-exact")
done = 1
for name, upto in (("falling", 30), ("landing", 90), ("settling", 240), ("resting", 480)):
    k = upto - done
    g.counters()  # (drain)
    t = time.perf_counter(); g.step(k); g.sync(); dt_g = (time.perf_counter() - t) / k
    c = g.counters()
    ko = min(k, 40)
    t = time.perf_counter(); o.step(ko); dt_o = (time.perf_counter() - t) / ko
    o.step(k - ko)
    done = upto
    gp, gv = g.read_bodies(); op, ov = o.read()
    same = np.array_equal(gp, op) and np.array_equal(gv, ov)
    print(f"{name:9s} ({done - k:3d}-{done:3d})   {1 / dt_g:10.0f} {dt_g * 1e3:9.3f} {1 / dt_o:16.1f} {c['num_manifolds']:18d} {c.get('full_updates', -1):12d}      {same}", flush=True)
c = g.counters()
print("counters:", {k_: c[k_] for k_ in ("num_pairs", "num_manifolds", "num_solver_contacts", "overflow_flags", "fast_steps", "full_steps", "replayed_steps") if k_ in c})
