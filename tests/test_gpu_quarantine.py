"""Containment of non-finite state through the C ABI: the reference's own quarantine tests
(/root/reference/src/pipeline/physics_pipeline/quarantine.rs:210-443) restated on the device path — NaN / infinity injected with
rp_bodies_write, rp_bodies_add_force and rp_bodies_set_next_kinematic_position, observed with rp_quarantine_read and rp_bodies_read.
The oracle holds no quarantine (it is a restatement of the healthy path), so these are outcome tests like the reference's."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S

pytestmark = pytest.mark.gpu


def _world_with_ground(extra=()):
    s = S.Scene(name="quarantine")
    s.gravity = (0.0, -9.81, 0.0)
    g = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(50.0, 0.5, 50.0))
    ids = []
    for kw in extra:
        kw = dict(kw)
        shape = kw.pop("shape", dict(shape=S.SHAPE_BALL, half_extents=(0.5, 0, 0)))
        b = s.add_body(**kw)
        s.add_collider(b, **shape)
        ids.append(b)
    return s, ids


def _assert_live_bodies_finite(w, skip=()):
    pos, vel = w.read_bodies()
    keep = np.ones(len(pos), bool)
    keep[list(skip)] = False
    assert np.isfinite(pos[keep]).all() and np.isfinite(vel[keep]).all()


def test_healthy_sim_never_quarantines():
    s, (b,) = _world_with_ground([dict(translation=(0, 3, 0))])
    w = PhysicsWorld.from_scene(s)
    for _ in range(60):
        w.step(1)
        assert w.quarantined().size == 0
    assert np.isfinite(w.read_bodies()[0][b]).all()


@pytest.mark.parametrize("bad", [np.nan, np.inf])
def test_user_set_non_finite_position_is_quarantined(bad):
    s, (poisoned, healthy) = _world_with_ground([dict(translation=(0, 3, 0)), dict(translation=(5, 3, 0))])
    w = PhysicsWorld.from_scene(s)
    w.step(1)
    before = w.read_bodies()[0][poisoned].copy()
    pose = before.copy(); pose[1] = bad
    w.write_bodies([poisoned], pos7=pose[None])
    w.step(1)
    assert list(w.quarantined()) == [poisoned]
    pos, vel = w.read_bodies()
    assert np.array_equal(pos[poisoned], before) and not vel[poisoned].any()  # last valid pose, stopped
    y = pos[healthy, 1]
    for _ in range(10):
        w.step(1)
        _assert_live_bodies_finite(w)
    pos2, _ = w.read_bodies()
    assert pos2[healthy, 1] < y                       # the simulation keeps running for everyone else
    assert np.array_equal(pos2[poisoned], before)     # disabled: no gravity, no motion
    assert w.counters()["overflow_flags"] == 0


def test_user_set_nan_linvel_is_quarantined():
    s, (poisoned,) = _world_with_ground([dict(translation=(0, 3, 0))])
    w = PhysicsWorld.from_scene(s)
    w.step(1)
    before = w.read_bodies()[0][poisoned].copy()
    w.write_bodies([poisoned], vel6=np.full((1, 6), np.nan, np.float32))
    w.step(1)
    assert list(w.quarantined()) == [poisoned]
    pos, vel = w.read_bodies()
    assert not vel[poisoned].any() and np.array_equal(pos[poisoned], before)  # neutralised before it could corrupt the pose


def test_nan_force_is_quarantined_at_end_of_step():
    s, (poisoned, other) = _world_with_ground([dict(translation=(0, 3, 0)), dict(translation=(6, 3, 0))])
    w = PhysicsWorld.from_scene(s)
    w.step(1)
    before = w.read_bodies()[0][poisoned].copy()
    # a NaN force is invisible at the start of the step; integration turns it into a NaN velocity and pose mid-step
    w.add_force([poisoned], force=np.full((1, 3), np.nan, np.float32))
    w.step(1)
    assert list(w.quarantined()) == [poisoned]
    pos, vel = w.read_bodies()
    assert not vel[poisoned].any() and np.array_equal(pos[poisoned], before)  # rolled back to the pose before the poisoned step
    for _ in range(10):
        w.step(1)
        _assert_live_bodies_finite(w)
    assert np.array_equal(w.read_bodies()[0][poisoned], before)


def test_nan_spread_through_contacts_is_contained():
    s, (bottom, top, far) = _world_with_ground([dict(translation=(0, 0.5, 0)), dict(translation=(0, 1.5, 0)), dict(translation=(8, 0.5, 0))])
    w = PhysicsWorld.from_scene(s)
    for _ in range(30):
        w.step(1)
        assert w.quarantined().size == 0
    w.add_force([bottom], force=np.full((1, 3), np.nan, np.float32))
    w.step(1)
    q = set(int(x) for x in w.quarantined())
    assert bottom in q and far not in q  # the island may be infected through the solver; nothing outside it is
    _assert_live_bodies_finite(w)
    for _ in range(10):
        w.step(1)
        assert set(int(x) for x in w.quarantined()) == q
        _assert_live_bodies_finite(w)


def test_joint_partner_survives_user_set_nan_pose():
    s, (poisoned, partner) = _world_with_ground([dict(translation=(0, 3, 0)), dict(translation=(0, 4, 0))])
    s.add_joint(poisoned, partner, (0, 0.5, 0), (0, -0.5, 0), locked_axes=S.LOCK_ALL)
    w = PhysicsWorld.from_scene(s)
    w.step(1)
    pose = w.read_bodies()[0][poisoned].copy(); pose[0] = np.nan
    w.write_bodies([poisoned], pos7=pose[None])
    w.step(1)
    assert list(w.quarantined()) == [poisoned]
    y = w.read_bodies()[0][partner, 1]
    for _ in range(10):
        w.step(1)
        _assert_live_bodies_finite(w)
    assert w.read_bodies()[0][partner, 1] < y  # released from the joint with the disabled body, it falls


def test_nan_kinematic_target_is_quarantined():
    s, (poisoned,) = _world_with_ground([dict(body_type=S.BODY_KINEMATIC_POSITION, translation=(0, 3, 0))])
    w = PhysicsWorld.from_scene(s)
    w.step(1)
    before = w.read_bodies()[0][poisoned].copy()
    target = before.copy(); target[:3] = np.nan
    w.set_next_kinematic_position([poisoned], target[None])
    w.step(1)
    assert list(w.quarantined()) == [poisoned]
    assert np.array_equal(w.read_bodies()[0][poisoned], before)  # repaired from the valid half of the pose


def test_quarantine_inside_a_resting_pyramid_keeps_the_fast_path_sound():
    """A poisoned cube in the middle of an LDS-resident island: the island kernel's write-back rolls it back, the host disables
    it at the next observation, and the pyramid above it re-settles with every live body finite."""
    w = PhysicsWorld.from_scene(S.pyramid10())
    w.step(40)
    victim = 12
    w.add_force([victim], force=np.array([[np.nan, 0, 0]], np.float32))
    w.step(5)
    q = set(int(x) for x in w.quarantined())
    assert victim in q
    w.step(100)
    _assert_live_bodies_finite(w)
    assert set(int(x) for x in w.quarantined()) == q
    assert w.counters()["overflow_flags"] == 0


def test_event_reads_with_a_small_cap_lose_nothing():
    """rp_collision_events_read writes min(queued, cap) events and leaves the rest queued: reading a burst of
    Started events three at a time yields exactly the list one large read yields."""
    def burst():
        s, ids = _world_with_ground([dict(translation=(2.0 * k, 0.45, 0)) for k in range(8)])
        s.enable_events(S.ACTIVE_EVENTS_COLLISION, 0.0)
        w = PhysicsWorld.from_scene(s)
        w.step(3)
        return w
    ref = burst().collision_events()
    assert len(ref) == 8
    w = burst()
    got = []
    while True:
        part = w.collision_events(cap=3)
        if len(part) == 0:
            break
        assert len(part) <= 3
        got.append(part)
    assert np.array_equal(np.concatenate(got), ref)
