import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_gpu_fuzz as F
from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sc, rng = F._scene(seed)
g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
for step in range(1, 4):
    g.step(1); o.step(1)
    gp, gv = g.read_bodies(); op, ov = o.read()
    bad = np.where((gp != op).any(1) | (gv != ov).any(1))[0]
    print("step", step, "bad bodies", bad.tolist())
    for b in bad[:6]:
        d = sc.bodies[b]
        cols = [i for i, p in enumerate(sc.collider_parents) if p == b]
        js = [j for j in range(len(sc.joints)) if int(sc.joints[j]['body1']) == b or int(sc.joints[j]['body2']) == b]
        print(f"  body {b}: type {int(d['body_type'])} locked {int(d['locked_axes']):#x} dom {int(d['dominance'])} addm {float(d['additional_mass'])} gyro {int(d['gyroscopic'])} grav {float(d['gravity_scale'])} damp {float(d['linear_damping'])},{float(d['angular_damping'])} ncol {len(cols)} shapes {[int(sc.colliders[c]['shape']) for c in cols]} joints {[(j, int(sc.joints[j]['locked_axes']), int(sc.joints[j]['limit_axes']), int(sc.joints[j]['motor_axes'])) for j in js]}")
        print("     gpu", gp[b], gv[b]); print("     ora", op[b], ov[b])
    if len(bad): break
print(g.counters()); print(o.stats())
gm, gn, gi = g.contacts(); om, on, oi = o.manifolds()
gk = {(a, b): (c, n, tuple(np.round(nn, 6)), tuple(i)) for (a, b, c, n), nn, i in zip(gm.tolist(), gn.tolist(), gi.tolist())}
ok = {(a, b): (c, n, tuple(np.round(nn, 6)), tuple(i)) for (a, b, c, n), nn, i in zip(om.tolist(), on.tolist(), oi.tolist())}
for key in sorted(set(gk) | set(ok)):
    if gk.get(key) != ok.get(key):
        print(f"manifold {key}: shapes {int(sc.colliders[key[0]]['shape'])},{int(sc.colliders[key[1]]['shape'])}\n   gpu {gk.get(key)}\n   ora {ok.get(key)}")
