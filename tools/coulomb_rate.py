"""steps/s of b3d_many_pyramids under FrictionModel::Coulomb: islands on k_island_solve_coul (the lane pair, rp_coulomb_pair.h), then
with RP_ISL_GENERIC=1 on k_island_generic (rows in HBM, one lane per manifold), and of the twist model through that kernel."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S
for model, generic in (("coulomb", False), ("coulomb", True), ("twist", False), ("twist", True)):
    sc = S.many_pyramids()
    if model == "coulomb":
        sc.params["friction_model"] = S.FRICTION_COULOMB
    if generic: os.environ["RP_ISL_GENERIC"] = "1"
    else: os.environ.pop("RP_ISL_GENERIC", None)
    w = PhysicsWorld.from_scene(sc); w.step(300); w.sync()
    t = time.perf_counter(); w.step(1000); w.sync(); dt = time.perf_counter() - t
    c = w.counters()
    print(f"C3 {model}{' (k_island_generic)' if generic else ''}: {1000 / dt:.1f} steps/s ({dt:.3f} ms/step)", {k: c[k] for k in ("fast_steps", "fused_steps", "full_steps", "replayed_steps") if k in c}, flush=True)
