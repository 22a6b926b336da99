import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld
sc = S.box_stack(3); sc.bodies[2]["dominance"] = 1
g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
for k in range(1, 10):
    g.step(1); o.step(1)
    gp, gv = g.read_bodies(); op, ov = o.read()
    gm, gn, gi = g.contacts(); om, on, oi = o.manifolds()
    print(k, "gpu y", gp[1:4, 1], "vy", gv[1:4, 1], "| ora y", op[1:4, 1], "vy", ov[1:4, 1])
    print("    gpu man", [(a[:4], np.round(b, 3).tolist()) for a, b in zip(gm.tolist(), gi.tolist())])
    print("    ora man", [(a[:4], np.round(b, 3).tolist()) for a, b in zip(om.tolist(), oi.tolist())])
    print("    ", g.counters()["num_solver_contacts"], o.stats()["num_solver_contacts"])
