"""The reference's box3d ports that need convex hulls, at FULL size (examples3d/b3d_junkyard.rs: 10,584 rocks + the orbiting pusher;
b3d_washer.rs: 8,000 cubes in the spinning ring of 40 hulls): device steps/s per window, the oracle beside it, and whether the two
states are equal bit for bit at the end of every window."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402
import oracle_ffi  # noqa: E402
from oracle_ffi import OracleWorld  # noqa: E402

oracle_ffi.set_threads(max(1, min(os.cpu_count() or 1, 16)))
which = sys.argv[1:] or ["junkyard", "washer"]
for name in which:
    sc = S.junkyard() if name == "junkyard" else S.washer()
    t = time.perf_counter(); g = PhysicsWorld.from_scene(sc); g.step(0); g.sync(); tb = time.perf_counter() - t
    o = OracleWorld(sc)
    print(f"b3d_{name}: {len(sc.bodies)} bodies, {len(sc.colliders)} colliders, {len(sc.polyhedra)} polyhedra; device world {tb:.2f} s")
    print("steps            GPU steps/s   ms/step   oracle steps/s (16 thr)   manifolds   solver contacts   bit-exact")
    done = 0
    for upto in (120, 240, 360, 480, 600):
        tg = to = 0.0
        for k in range(done + 1, upto + 1):
            if name == "junkyard":
                tgt = S.junkyard_pusher_target(k)
                g.set_next_kinematic_position([sc.pusher], tgt); o.set_next_kinematic_position(sc.pusher, tgt)
            t = time.perf_counter(); g.step(1); g.sync(); tg += time.perf_counter() - t
            t = time.perf_counter(); o.step(1); to += time.perf_counter() - t
        n = upto - done; done = upto
        gp, gv = g.read_bodies(); op, ov = o.read()
        same = np.array_equal(gp, op) and np.array_equal(gv, ov)
        c = g.counters()
        print(f"{upto - n:4d}-{upto:4d} {n / tg:14.0f} {tg / n * 1e3:9.3f} {n / to:18.1f} {c['num_manifolds']:18d} {c['num_solver_contacts']:12d}        {same}", flush=True)
    c = g.counters()
    print("counters:", {k_: c[k_] for k_ in ("num_pairs", "num_manifolds", "overflow_flags", "num_tiles", "fast_steps", "full_steps", "replayed_steps", "lean_steps") if k_ in c})
