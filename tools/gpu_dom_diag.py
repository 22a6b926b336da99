import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld

def run(name, sc, steps=60):
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for k in range(1, steps + 1):
        g.step(1); o.step(1)
        gp, gv = g.read_bodies(); op, ov = o.read()
        if (gp != op).any() or (gv != ov).any():
            bad = np.where((gp != op).any(1) | (gv != ov).any(1))[0]
            print(f"{name}: DIVERGES at step {k} bodies {bad.tolist()} maxdv {np.abs(gv-ov).max():.4g}")
            gm, gn, gi = g.contacts(); om, on, oi = o.manifolds()
            for a, b in zip(sorted(zip(gm.tolist(), gi.tolist())), sorted(zip(om.tolist(), oi.tolist()))):
                print("   gpu", a, "\n   ora", b)
            return
    print(f"{name}: ok")

for dom_idx in (None, 1, 2, 3):
    sc = S.box_stack(3)
    if dom_idx is not None:
        sc.bodies[dom_idx]["dominance"] = 1
    run(f"stack3 dom body {dom_idx}", sc)
sc = S.box_stack(3, gap=0.3)
sc.bodies[3]["dominance"] = 1
run("stack3 gap, top dom", sc)
sc = S.box_stack(4)
sc.bodies[2]["dominance"] = 1; sc.bodies[4]["dominance"] = -1
run("stack4 mixed", sc)
sc = S.tumble(24, seed=3)
for i in range(1, 25, 3):
    sc.bodies[i]["dominance"] = 1
run("tumble24 dom", sc, 120)
print("--- global path (Coulomb model) on the same stacks")
for dom_idx in (2, 3):
    sc = S.box_stack(3); sc.bodies[dom_idx]["dominance"] = 1; sc.params["friction_model"] = S.FRICTION_COULOMB
    run(f"coulomb stack3 dom body {dom_idx}", sc)
# a joint poisons the island -> global path with the twist model
sc = S.box_stack(3); sc.bodies[2]["dominance"] = 1
sc.add_joint(0, 3, (5.0, 9.0, 0.0), (5.0, 6.5, 0.0), locked_axes=0, limits={0: (-100.0, 100.0)})
run("twist, global path (free joint on the top box) dom body 2", sc)
