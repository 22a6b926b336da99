"""Sensors on the device path (narrow_phase/intersections.rs): the intersection graph of ColliderBuilder::sensor(true) colliders
against the oracle — same Started / Stopped | SENSOR event stream, same NarrowPhase::intersection_pair answers, and bodies that
fall through the sensors bit for bit as in the oracle.  Scenes: miri_scenes.rs:196-229 (sensor_overlap) and a trigger volume
crossed by every shape kind."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld
from test_reference_kats import ground, quat_from_scaled_axis, world

pytestmark = pytest.mark.gpu


def _lockstep(sc, steps, every=1):
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for k in range(0, steps, every):
        g.step(every); o.step(every)
        ge = [tuple(int(x) for x in e) for e in g.collision_events()]
        oe = [tuple(int(x) for x in e) for e in o.collision_events()]
        assert ge == oe, (k, ge, oe)
        gp, gv = g.read_bodies(); op, ov = o.read()
        np.testing.assert_array_equal(gp, op); np.testing.assert_array_equal(gv, ov)
        yield k + every, g, o, ge


def test_sensor_overlap_matches_the_oracle():
    sc = world()
    sb = sc.add_body(body_type=S.BODY_FIXED)
    sensor = sc.add_collider(sb, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0), active_events=S.ACTIVE_EVENTS_COLLISION, sensor=1)
    ball = sc.add_body(translation=(0.0, 0.4, 0.0))
    ball_co = sc.add_collider(ball, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0))
    seen = []
    for step, g, o, ev in _lockstep(sc, 121):
        seen += ev
        assert g.intersection_pair(sensor, ball_co) == o.intersection_pair(sensor, ball_co)
        if step == 1:
            assert g.intersection_pair(sensor, ball_co) is True and len(ev) == 1 and ev[0][2] == 1 and ev[0][3] & 1  # Started | SENSOR
    assert [e[2] for e in seen] == [1, 0] and all(e[3] & 1 for e in seen)
    assert g.read_bodies()[0][ball, 1] < -5.0          # it fell straight through: sensors exert no force
    assert g.counters()["num_manifolds"] == 0


def test_trigger_volume_crossed_by_every_shape():
    sc = world()
    ground(sc)
    trig = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, 3.0, 0.0))
    tc = sc.add_collider(trig, half_extents=(4.0, 0.5, 4.0), active_events=S.ACTIVE_EVENTS_COLLISION, sensor=1)
    cap = sc.add_body(translation=(-2.0, 6.0, 0.0), rotation=quat_from_scaled_axis((0.0, 0.0, 0.7)))
    sc.add_collider(cap, shape=S.SHAPE_CAPSULE, half_extents=(0.5, 0.25, 1.0))
    box = sc.add_body(translation=(0.0, 6.5, 0.0), rotation=quat_from_scaled_axis((0.3, 0.2, 0.1)))
    sc.add_collider(box, half_extents=(0.3, 0.3, 0.3))
    ball = sc.add_body(translation=(2.0, 7.0, 0.0))
    sc.add_collider(ball, shape=S.SHAPE_BALL, half_extents=(0.3, 0.0, 0.0))
    started, stopped = set(), set()
    for step, g, o, ev in _lockstep(sc, 240, every=4):
        for c1, c2, st, fl, _ in ev:
            assert fl & 1 and tc in (c1, c2)
            (started if st else stopped).add(c1 + c2 - tc)
    assert started == stopped == {tc + 1, tc + 2, tc + 3}
    assert g.read_bodies()[0][[cap, box, ball], 1].max() < 1.0  # all three rest on the ground, below the trigger
    assert len(g.intersection_pairs()) == 0 and all(o.intersection_pair(tc, c) is None for c in (tc + 1, tc + 2, tc + 3))  # out of the trigger's fat AABB: no pair left


def test_moving_sensors_of_every_shape_and_removal():
    """Dynamic sensor colliders (a capsule and a ball sensor riding on falling bodies, a cuboid sensor on a kinematic platform)
    sweeping over resting shapes; then a collider inside a sensor is removed: Stopped | SENSOR | REMOVED."""
    sc = world()
    ground(sc)
    rest = []
    for k, (shape, he) in enumerate([(S.SHAPE_BALL, (0.4, 0, 0)), (S.SHAPE_CUBOID, (0.4, 0.4, 0.4)), (S.SHAPE_CAPSULE, (0.4, 0.3, 0.0))]):
        b = sc.add_body(translation=(3.0 * k - 3.0, 0.45, 0.0))
        rest.append(sc.add_collider(b, shape=shape, half_extents=he, active_events=S.ACTIVE_EVENTS_COLLISION))
    for k, (shape, he) in enumerate([(S.SHAPE_CAPSULE, (0.6, 0.3, 1.0)), (S.SHAPE_BALL, (0.7, 0, 0)), (S.SHAPE_CUBOID, (0.5, 0.2, 0.5))]):
        b = sc.add_body(translation=(3.0 * k - 3.0 + 0.2, 4.0 + k, 0.1), rotation=quat_from_scaled_axis((0.2 * k, 0.1, 0.4)), gravity_scale=0.3)
        sc.add_collider(b, shape=shape, half_extents=he, density=0.5, sensor=1)   # falls THROUGH the resting shape and the floor
    plat = sc.add_body(body_type=S.BODY_KINEMATIC_VELOCITY, translation=(-6.0, 0.6, 0.0), linvel=(2.0, 0.0, 0.0))
    sc.add_collider(plat, half_extents=(0.5, 0.5, 0.5), sensor=1, active_events=S.ACTIVE_EVENTS_COLLISION)
    n_events = 0
    for step, g, o, ev in _lockstep(sc, 300, every=3):
        n_events += len(ev)
        assert sorted(map(tuple, g.intersection_pairs().tolist())) == sorted(
            (a, b, int(bool(o.intersection_pair(a, b)))) for a, b, _ in g.intersection_pairs().tolist())
    assert n_events >= 10
    # removal of a collider that currently intersects a sensor
    sc2 = world()
    sb = sc2.add_body(body_type=S.BODY_FIXED)
    sc2.add_collider(sb, half_extents=(2.0, 2.0, 2.0), sensor=1, active_events=S.ACTIVE_EVENTS_COLLISION)
    inner = sc2.add_body(translation=(0.0, 0.0, 0.0), gravity_scale=0.0)
    ic = sc2.add_collider(inner, shape=S.SHAPE_BALL, half_extents=(0.3, 0, 0))
    g, o = PhysicsWorld.from_scene(sc2), OracleWorld(sc2)
    g.step(2); o.step(2)
    assert [tuple(e)[:4] for e in g.collision_events().tolist()] == [tuple(e)[:4] for e in o.collision_events()] != []
    g.remove_collider([ic]); o.remove_collider(ic)
    g.step(2); o.step(2)
    ge, oe = g.collision_events().tolist(), [list(map(int, e)) for e in o.collision_events()]
    assert ge == oe and len(ge) == 1 and ge[0][2] == 0 and ge[0][3] == 3   # Stopped, SENSOR | REMOVED
