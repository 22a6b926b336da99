"""What a feature costs on the headline scene (b3d_many_pyramids, 10,780 cuboids): settled steps/s with the feature on, and which way
the steps went (VERDICT r4 weak #5 "feature cliffs off the fused step")."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402


def compound():
    """every cuboid's collider sits 1 mm off its body origin: the same pile, but a world of 'compound bodies' to the planner"""
    sc = S.many_pyramids()
    for c, p in zip(sc.colliders, sc.collider_parents):
        if p > 0:
            c["translation"] = (0.001, 0.0, 0.0)
    return sc


def sensor():
    sc = S.many_pyramids()
    t = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, 40.0, 0.0))
    sc.add_collider(t, half_extents=(200.0, 0.5, 200.0), sensor=1, active_events=S.ACTIVE_EVENTS_COLLISION)
    return sc


def coulomb():
    sc = S.many_pyramids(); sc.params["friction_model"] = S.FRICTION_COULOMB
    return sc


cases = [("plain", S.many_pyramids, 60, 1000), ("compound bodies", compound, 60, 1000), ("a sensor overhead", sensor, 60, 1000),
         ("contact-force events", lambda: S.many_pyramids().enable_events(3, 100.0), 60, 1000), ("Coulomb friction", coulomb, 60, 300)]
print(f"{'world':28s} {'steps/s':>9s} {'us/step':>9s}  fused / fast / full / replayed")
for name, make, warm, steps in cases:
    w = PhysicsWorld.from_scene(make())
    w.step(warm); w.sync(); w.step(10); w.sync()
    c0 = w.counters()
    t = time.perf_counter(); w.step(steps); w.sync(); dt = time.perf_counter() - t
    c1 = w.counters()
    d = {k: c1[k] - c0[k] for k in ("fused_steps", "fast_steps", "full_steps", "replayed_steps")}
    if name.startswith("contact"):
        w.contact_force_events()
    print(f"{name:28s} {steps / dt:9.0f} {dt / steps * 1e6:9.1f}  {d['fused_steps']} / {d['fast_steps']} / {d['full_steps']} / {d['replayed_steps']}")
