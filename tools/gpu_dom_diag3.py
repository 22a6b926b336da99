import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld
def run(name, mod):
    sc = S.box_stack(3); sc.bodies[2]["dominance"] = 1
    mod(sc)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(7); o.step(7)
    gp, gv = g.read_bodies(); op, ov = o.read()
    gm, gn, gi = g.contacts(); om, on, oi = o.manifolds()
    print(name, "vy1 gpu", gv[1, 1], "ora", ov[1, 1], "imp gpu", np.round(gi[:, 0], 3).tolist(), "ora", np.round(oi[:, 0], 3).tolist())
run("as is", lambda sc: None)
run("static freq = dyn freq", lambda sc: sc.params.__setitem__("static_contact_natural_frequency", 30.0))
run("warmstart 0", lambda sc: sc.params.__setitem__("warmstart_coefficient", 0.0))
run("1 substep", lambda sc: sc.params.__setitem__("num_solver_iterations", 1))
run("friction 0", lambda sc: [c.__setitem__("friction", 0.0) for c in sc.colliders])
run("top box removed", lambda sc: (sc.bodies.pop(), sc.colliders.pop(), sc.collider_parents.pop()))
