"""steps/s of b3d_many_pyramids under FrictionModel::Coulomb (islands on k_island_generic) and, with RP_ISL_GENERIC=1, of the twist
model through the same kernel."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S
for model in ("coulomb", "twist"):
    sc = S.many_pyramids()
    if model == "coulomb":
        sc.params["friction_model"] = S.FRICTION_COULOMB
    w = PhysicsWorld.from_scene(sc); w.step(300); w.sync()
    t = time.perf_counter(); w.step(1000); w.sync(); dt = time.perf_counter() - t
    c = w.counters()
    print(f"C3 {model}: {1000 / dt:.1f} steps/s ({dt:.3f} ms/step)", {k: c[k] for k in ("fast_steps", "full_steps", "replayed_steps") if k in c})
