"""What a short timed region costs: `bench.py --steps 20 --warmup 5` (the driver's command) times ONE rp_step(20) + rp_sync.  Prints the
wall time of regions of several lengths on b3d_many_pyramids, the launches each took, and the fixed cost per region (intercept of a fit)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapier_amd import PhysicsWorld, scenes as S

w = PhysicsWorld.from_scene(S.many_pyramids())
w.step(0); w.step(5); w.sync()
rows = []
for n in (20, 20, 20, 1, 1, 2, 4, 8, 16, 32, 64, 128, 256, 20, 20):
    c0 = w.counters()
    t0 = time.perf_counter(); w.step(n); t1 = time.perf_counter(); w.sync(); t2 = time.perf_counter()
    c1 = w.counters()
    rows.append((n, (t2 - t0) * 1e6, (t1 - t0) * 1e6))
    print(f"steps {n:4d}: region {(t2 - t0) * 1e6:9.1f} us ({(t2 - t0) * 1e6 / n:7.2f} per step), rp_step returned after {(t1 - t0) * 1e6:7.1f} us, "
          f"launches {c1['fused_launches'] - c0['fused_launches']}, fused {c1['fused_steps'] - c0['fused_steps']}, full {c1['full_steps'] - c0['full_steps']}, replayed {c1['replayed_steps'] - c0['replayed_steps']}")
x = np.array([r[0] for r in rows[3:13]], float); y = np.array([r[1] for r in rows[3:13]], float)
a, b = np.polyfit(x, y, 1)
print(f"fit: region = {b:.1f} us + {a:.2f} us per step")
