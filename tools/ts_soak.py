"""Memory-model soak of k_tile_step: b3d_large_pyramid for N steps on the one-launch TGS loop (write-through publishes read back with sc1
loads, neighbour flags, no agent acquire) against the same world on the sixteen sweep / prepare launches per step (kernel boundaries): one
stale halo read anywhere in N x 12 phases x 240 tiles and the two trajectories of the settling pile part.   python tools/ts_soak.py [steps=30000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rapier_amd import PhysicsWorld, scenes as S
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
sc = S.large_pyramid(200)
os.environ.pop("RP_NO_TILE_STEP", None)
a = PhysicsWorld.from_scene(sc)
os.environ["RP_NO_TILE_STEP"] = "1"
b = PhysicsWorld.from_scene(sc)
os.environ.pop("RP_NO_TILE_STEP", None)
done = 0
for chunk in (1000, 4000, 5000, 10000, 10000, 20000, 50000):
    if done >= steps: break
    n = min(chunk, steps - done)
    t = time.perf_counter(); a.step(n); a.sync(); ta = time.perf_counter() - t
    t = time.perf_counter(); b.step(n); b.sync(); tb = time.perf_counter() - t
    done += n
    pa, va = a.read_bodies(); pb, vb = b.read_bodies()
    ca, cb = a.counters(), b.counters()
    same = np.array_equal(pa, pb) and np.array_equal(va, vb)
    print(f"step {done}: identical {same}; k_tile_step {n / ta:,.0f} steps/s ({ca['tile_step_steps']} steps on it: {ca['lean_steps']} lean + {ca['full_steps']} full graphs enqueued, disabled {ca['joint_net_disabled']}, replayed {ca['replayed_steps']}), "
          f"sweep launches {n / tb:,.0f} steps/s ({cb['tile_step_steps']}); manifolds {ca['num_manifolds']}", flush=True)
    assert same and np.isfinite(pa).all()
