"""Does the dense form of k_island_solve really keep two workgroups on a CU?  361 islands, RP_ISL_DENSE=1: if the fused launch's
arrival barrier times out (FL_GRID_TIMEOUT: not every workgroup resident), the world drops to the two-kernel fast graph after one
~1 s stall and `replayed_steps` counts the step that was replayed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S
for r, c in ((16, 16), (19, 19), (20, 24)):
    w = PhysicsWorld.from_scene(S.many_pyramids(rows=r, cols=c))
    t = time.perf_counter(); w.step(60); w.sync(); warm = time.perf_counter() - t
    t = time.perf_counter(); w.step(600); w.sync(); dt = (time.perf_counter() - t) / 600
    k = w.counters()
    print(f"{r * c} islands, RP_ISL_DENSE={os.environ.get('RP_ISL_DENSE')}: warm-up {warm:.3f} s, {dt * 1e6:.1f} us/step, fast {k['fast_steps']} full {k['full_steps']} replayed {k['replayed_steps']}")
