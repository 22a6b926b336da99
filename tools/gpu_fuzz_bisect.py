"""Bisect a fuzz divergence: keep the ground + a chosen set of bodies of a seed scene, neutralise attributes one at a time."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import copy
import numpy as np
import test_gpu_fuzz as F
from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld

seed = int(sys.argv[1]); keep = [int(x) for x in sys.argv[2].split(',')]
full, _ = F._scene(seed)


def sub(mod=None):
    sc = S.Scene(name="sub", gravity=full.gravity)
    sc.params = full.params.copy()
    remap = {}
    for b in [0] + keep:
        remap[b] = len(sc.bodies)
        sc.bodies.append(full.bodies[b].copy())
    for c, p in enumerate(full.collider_parents):
        if p in remap:
            sc.colliders.append(full.colliders[c].copy()); sc.collider_parents.append(remap[p])
    if mod:
        mod(sc)
    return sc


def run(name, sc, steps=3):
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for k in range(1, steps + 1):
        g.step(1); o.step(1)
        gp, gv = g.read_bodies(); op, ov = o.read()
        if (gp != op).any() or (gv != ov).any():
            print(f"{name}: DIVERGES at step {k}", np.abs(gv - ov).max()); return
    print(f"{name}: ok")


def setall(field, value, what="colliders"):
    def m(sc):
        for x in getattr(sc, what):
            x[field] = value
    return m


run("as is", sub())
run("restitution 0", sub(setall("restitution", 0.0)))
run("friction 0.5", sub(setall("friction", 0.5)))
run("rules avg", sub(lambda sc: (setall("friction_rule", 0)(sc), setall("restitution_rule", 0)(sc))))
run("no events", sub(setall("active_events", 0)))
run("identity body rot", sub(setall("rotation", (0, 0, 0, 1), "bodies")))
run("zero vel", sub(lambda sc: (setall("linvel", (0, 0, 0), "bodies")(sc), setall("angvel", (0, 0, 0), "bodies")(sc))))
run("no damping/gscale", sub(lambda sc: (setall("linear_damping", 0.0, "bodies")(sc), setall("angular_damping", 0.0, "bodies")(sc), setall("gravity_scale", 1.0, "bodies")(sc))))
run("no sleep", sub(setall("can_sleep", 0, "bodies")))
run("gyro off", sub(setall("gyroscopic", 0, "bodies")))
run("groups all", sub(lambda sc: (setall("collision_memberships", 0xFFFFFFFF)(sc), setall("collision_filter", 0xFFFFFFFF)(sc))))

print("--- full scene variants")
def fullmod(mod=None, joints=True):
    sc = copy.deepcopy(full)
    if not joints:
        sc.joints = []
    if mod:
        mod(sc)
    return sc
run("full", fullmod())
run("full, no joints", fullmod(joints=False))
run("full, dominance 0", fullmod(setall("dominance", 0, "bodies")))
run("full, no locked axes", fullmod(setall("locked_axes", 0, "bodies")))
run("full, no additional mass", fullmod(setall("additional_mass", 0.0, "bodies")))
run("full, platform fixed", fullmod(lambda sc: sc.bodies[4].__setitem__("body_type", S.BODY_FIXED)))
run("full, no sleep", fullmod(setall("can_sleep", 0, "bodies")))
run("full, no events", fullmod(setall("active_events", 0)))
run("full, restitution 0", fullmod(setall("restitution", 0.0)))
run("full, groups all", fullmod(lambda sc: (setall("collision_memberships", 0xFFFFFFFF)(sc), setall("collision_filter", 0xFFFFFFFF)(sc))))
run("full, warmstart_joints 0", fullmod(lambda sc: sc.params.__setitem__("warmstart_joints", 0)))
