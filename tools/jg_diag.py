"""b3d_joint_grid: steps/s and path counters under the current environment switches (A/B runs on one box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S
name = sys.argv[1] if len(sys.argv) > 1 else "jg"
sc = S.joint_grid(100) if name == "jg" else S.large_pyramid(200)
w = PhysicsWorld.from_scene(sc)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
w.step(60); w.sync()
t = time.perf_counter(); w.step(N); w.sync(); dt = (time.perf_counter() - t) * 1000 / N
c = w.counters()
print(name, {k: os.environ.get(k) for k in ("RP_NO_BP_INCR", "RP_NO_LEAN", "RP_NO_TILES")}, f"{1000 / dt:.0f} steps/s {dt:.4f} ms/step",
      {k: c[k] for k in ("num_pairs", "bp_rebuilds", "lean_steps", "full_steps", "replayed_steps", "num_tiles", "full_updates")})
