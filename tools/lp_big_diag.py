"""A pyramid taller than BASELINE's (base 300 / 400 collapses while it settles): counters every 50 steps — robustness probe.
    python tools/lp_big_diag.py <base> [steps] [timers at step]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S
base = int(sys.argv[1]); total = int(sys.argv[2]) if len(sys.argv) > 2 else 700; tim = int(sys.argv[3]) if len(sys.argv) > 3 else -1
w = PhysicsWorld.from_scene(S.large_pyramid(base))
done = 0
while done < total:
    if done == tim: w.enable_timers(True)
    w.step(50); w.sync(); done += 50
    c = w.counters()
    print(base, done, {k: c[k] for k in ("num_pairs", "num_manifolds", "num_tiles", "tile_sweeps", "overflow_flags", "num_colors", "num_parallel_stages", "num_global_bodies", "full_updates", "lean_steps", "replayed_steps")}, flush=True)
pos, vel = w.read_bodies()
import numpy as np
assert np.isfinite(pos).all() and np.isfinite(vel).all()
