"""Memory-model soak of k_joint_net_step: b3d_joint_grid for N steps on the joint-net launch (write-through publishes, neighbour flags, the
solver forked beside the collision stage) against the same world on the eight sweep launches (kernel boundaries between sweeps): one stale
halo read anywhere in N x 8 sweeps x 155 tiles and the two chaotic trajectories part.     python tools/jn_soak.py [steps=100000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rapier_amd import PhysicsWorld, scenes as S
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sc = S.joint_grid(100)
os.environ.pop("RP_NO_JOINT_NET", None)
a = PhysicsWorld.from_scene(sc)
os.environ["RP_NO_JOINT_NET"] = "1"
b = PhysicsWorld.from_scene(sc)
os.environ.pop("RP_NO_JOINT_NET", None)
done = 0
for chunk in (1000, 9000, 40000, 50000, 100000, 300000):
    if done >= steps: break
    n = min(chunk, steps - done)
    t = time.perf_counter(); a.step(n); a.sync(); ta = time.perf_counter() - t
    t = time.perf_counter(); b.step(n); b.sync(); tb = time.perf_counter() - t
    done += n
    pa, va = a.read_bodies(); pb, vb = b.read_bodies()
    ca, cb = a.counters(), b.counters()
    same = np.array_equal(pa, pb) and np.array_equal(va, vb)
    print(f"step {done}: identical {same}; joint-net launch {n / ta:,.0f} steps/s ({ca['joint_net_steps']} steps on it, disabled {ca['joint_net_disabled']}, replayed {ca['replayed_steps']}), sweep launches {n / tb:,.0f} steps/s ({cb['joint_net_steps']})", flush=True)
    assert same and np.isfinite(pa).all()
