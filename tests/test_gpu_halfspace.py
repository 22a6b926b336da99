"""Half-spaces on the device path (RP_SHAPE_HALFSPACE = ColliderBuilder::halfspace) against the oracle, bit for bit: the plane as
first and as second collider of its pairs, on no body / a fixed body / a moving kinematic body, under cuboids, balls and capsules,
with events, sensors, sleeping and deep penetration (the non-solid point projection of contact_manifold_convex_ball)."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld

pytestmark = pytest.mark.gpu


def _lockstep(sc, steps, every=1, sleeping=False):
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for k in range(0, steps, every):
        g.step(every); o.step(every)
        ge = sorted(tuple(int(x) for x in e) for e in g.collision_events())
        oe = sorted(tuple(int(x) for x in e) for e in o.collision_events())
        assert ge == oe, (k, ge, oe)
        gp, gv = g.read_bodies(); op, ov = o.read()
        np.testing.assert_array_equal(gp, op, err_msg=f"poses at step {k + every}")
        np.testing.assert_array_equal(gv, ov, err_msg=f"velocities at step {k + every}")
        if sleeping:
            np.testing.assert_array_equal(g.sleeping(), o.sleeping())
        yield k + every, g, o, ge
    assert g.counters()["overflow_flags"] == 0


def test_halfspace_scene_bit_exact():
    sc = S.halfspace_scene()
    n_events = 0
    for step, g, o, ev in _lockstep(sc, 400, every=5):
        n_events += len(ev)
    assert n_events >= 16                                           # every dynamic collider touched the ground plane (events on)
    c = g.counters()
    assert c["num_manifolds"] >= 16
    pos, _ = g.read_bodies()
    assert pos[1:17, 1].min() > 0.15                                # nothing fell through a plane


def test_halfspace_scene_larger_and_with_sleeping():
    sc = S.halfspace_scene(n_side=7)
    sc.enable_sleep()                                              # the riders of the rising plane stay awake, the rest falls asleep
    for step, g, o, ev in _lockstep(sc, 600, every=20, sleeping=True):
        pass
    asleep = g.sleeping()
    assert asleep.any() and not asleep.all()


def test_deep_penetration_of_balls_is_resolved_like_the_oracle():
    sc = S.Scene(name="deep", gravity=(0.0, -2.0, 0.0))
    sc.add_collider(-1, shape=S.SHAPE_HALFSPACE, half_extents=(0.0, 1.0, 0.0))
    g0 = sc.add_body(body_type=S.BODY_FIXED, translation=(4.0, 1.0, 0.0)); sc.add_collider(g0, half_extents=(1.0, 1.0, 1.0))
    g1 = sc.add_body(body_type=S.BODY_FIXED, translation=(8.0, 1.0, 0.0)); sc.add_collider(g1, shape=S.SHAPE_CAPSULE, half_extents=(1.0, 0.6, 0.0))
    for p in [(0.0, -0.2, 0.0), (4.2, 1.8, 0.1), (8.3, 1.2, 0.05), (8.0, 1.0, 0.0)]:   # below the plane, inside the box, inside the capsule, ON its segment
        b = sc.add_body(translation=p); sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.25, 0.0, 0.0))
    for step, g, o, ev in _lockstep(sc, 240, every=4):
        pass
    pos, _ = g.read_bodies()
    assert pos[2, 1] > 0.2 and pos[3, 1] > 2.2 and pos[2:, 1].min() > 0.2   # out of the plane, on top of the box, nobody left inside a shape or below the plane


def test_halfspace_sensor_and_removal():
    sc = S.Scene(name="hs_sensor", gravity=(0.0, -9.81, 0.0))
    hs = sc.add_collider(-1, shape=S.SHAPE_HALFSPACE, half_extents=(0.0, 1.0, 0.0), sensor=1, active_events=S.ACTIVE_EVENTS_COLLISION)
    cols = []
    for k, (shape, he) in enumerate([(S.SHAPE_BALL, (0.3, 0, 0)), (S.SHAPE_CUBOID, (0.3, 0.2, 0.1)), (S.SHAPE_CAPSULE, (0.4, 0.2, 0.0))]):
        b = sc.add_body(translation=(2.0 * k, 1.0 + 0.5 * k, 0.0), rotation=(0.2, 0.1, 0.3, 0.9273618))
        cols.append(sc.add_collider(b, shape=shape, half_extents=he))
    started = []
    for step, g, o, ev in _lockstep(sc, 90):
        started += [e[1] for e in ev if e[2] == 1 and e[3] & 1]
        for c in cols:
            assert g.intersection_pair(hs, c) == o.intersection_pair(hs, c)
    assert started == cols
    g.remove_collider(hs); o.remove_collider(hs)                    # Stopped | SENSOR | REMOVED for the three intersecting pairs
    g.step(1); o.step(1)
    ge = sorted(tuple(int(x) for x in e) for e in g.collision_events()); oe = sorted(tuple(int(x) for x in e) for e in o.collision_events())
    assert ge == oe and len(ge) == 3 and all(e[2] == 0 for e in ge)


def test_halfspace_inserted_into_a_running_world_and_validation():
    sc = S.Scene(name="late_plane", gravity=(0.0, -9.81, 0.0))
    for k in range(6):
        b = sc.add_body(translation=(1.5 * k, 3.0 + 0.2 * k, 0.0), rotation=(0.1 * k, 0.0, 0.1, 0.99))
        sc.add_collider(b, shape=[S.SHAPE_CUBOID, S.SHAPE_BALL, S.SHAPE_CAPSULE][k % 3], half_extents=(0.3, 0.25, 1.0))
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(20); o.step(20)
    plane = S.collider_desc(shape=S.SHAPE_HALFSPACE, half_extents=(0.0, 1.0, 0.0), friction=0.9)
    g.insert_collider(plane); o.add_collider(-1, shape=S.SHAPE_HALFSPACE, half_extents=(0.0, 1.0, 0.0), friction=0.9)
    for _ in range(12):
        g.step(15); o.step(15)
        gp, gv = g.read_bodies(); op, ov = o.read()
        np.testing.assert_array_equal(gp, op); np.testing.assert_array_equal(gv, ov)
    assert gp[:, 1].min() > 0.2
    with pytest.raises(Exception, match="unit outward normal"):
        g.insert_collider(S.collider_desc(shape=S.SHAPE_HALFSPACE, half_extents=(0.0, 2.0, 0.0)))
    with pytest.raises(Exception, match="fixed or kinematic parent"):
        g.insert_collider(S.collider_desc(shape=S.SHAPE_HALFSPACE, half_extents=(0.0, 1.0, 0.0)), parent=0)
    g.step(5); o.step(5)                                            # the refused insertions left the world untouched
    np.testing.assert_array_equal(g.read_bodies()[0], o.read()[0])
