"""The single-kernel fused step (k_island_solve validates the step itself, rp_api.hip plan_fused) in the worlds that used to fall off
it: bodies with several / offset colliders, sleep-enabled worlds whose bodies are awake, worlds with sensors,
worlds that raise contact-force events.  Each world runs in lockstep with the oracle — poses, velocities, events, sleep states bit for
bit — and must have taken fused steps (rp_counters::fused_steps)."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld

pytestmark = pytest.mark.gpu


def _same_state(g, o, msg):
    gp, gv = g.read_bodies(); op, ov = o.read()
    assert np.isfinite(gp).all()
    np.testing.assert_array_equal(gp, op, err_msg=msg); np.testing.assert_array_equal(gv, ov, err_msg=msg)


def _dumbbells(n=6):
    """compound bodies: dumbbells (a capsule + two balls, colliders away from the body origin) dropped criss-cross on a slab, a box
    with an offset collider on top"""
    s = S.Scene(name=f"dumbbells_{n}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(30.0, 0.5, 30.0))
    for i in range(n):
        d = s.add_body(translation=(0.1 * i, 0.6 + 0.9 * i, 0.05 * i), rotation=(0.0, 0.38268343, 0.0, 0.92387953) if i % 2 else (0.0, 0.0, 0.0, 1.0))
        s.add_collider(d, shape=S.SHAPE_CAPSULE, half_extents=(0.8, 0.15, 0.0))
        s.add_collider(d, shape=S.SHAPE_BALL, half_extents=(0.35, 0.0, 0.0), translation=(0.95, 0.0, 0.0))
        s.add_collider(d, shape=S.SHAPE_BALL, half_extents=(0.35, 0.0, 0.0), translation=(-0.95, 0.0, 0.0))
    b = s.add_body(translation=(4.0, 0.5, 0.0))
    s.add_collider(b, half_extents=(0.5, 0.5, 0.5), translation=(0.2, 0.0, 0.1))
    return s


def test_compound_bodies_keep_the_fused_step():
    sc = _dumbbells()
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for k in range(10):
        g.step(40); o.step(40)
        _same_state(g, o, f"dumbbells +{40 * (k + 1)}")
    c = g.counters()
    assert c["fused_steps"] > 50 and c["overflow_flags"] == 0, c
    assert c["num_manifolds"] == o.stats()["num_active_manifolds"]
    # a kick: the dumbbell's far ball leaves its fat AABB first (the body origin barely moves) — the validators walk every collider
    v = np.array([[0.0, 0.0, 0.0, 0.0, 6.0, 0.0]], np.float32)
    g.write_bodies([1], vel6=v); o.set_vel(1, v[0, :3], v[0, 3:])
    for k in range(5):
        g.step(20); o.step(20)
        _same_state(g, o, f"dumbbells after the spin +{20 * (k + 1)}")


def test_sleep_enabled_world_keeps_the_fused_step_while_awake_and_falls_asleep_on_time():
    sc = S.many_pyramids(rows=2, cols=3).enable_sleep()
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    fused_while_awake = 0
    for k in range(30):
        g.step(8); o.step(8)
        _same_state(g, o, f"sleep-enabled pyramids +{8 * (k + 1)}")
        np.testing.assert_array_equal(g.sleeping(), o.sleeping(), err_msg=f"sleep states +{8 * (k + 1)}")
        if not g.sleeping().any():
            fused_while_awake = g.counters()["fused_steps"]
    assert fused_while_awake > 10, fused_while_awake
    assert g.sleeping()[1:].all()                                    # every pyramid asleep in the end
    # wake one pyramid up: the step goes through the full graph, then back to fused steps
    v = np.array([[1.0, 2.0, 0.0, 0.0, 0.0, 0.0]], np.float32)
    g.write_bodies([55], vel6=v); o.set_vel(55, v[0, :3], v[0, 3:])
    g.wake_up([55]); o.wake_up(55)
    before = g.counters()["fused_steps"]
    for k in range(10):
        g.step(6); o.step(6)
        _same_state(g, o, f"after the wake-up +{6 * (k + 1)}")
        np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    assert g.counters()["fused_steps"] > before


def test_world_with_a_sensor_keeps_the_fused_step_and_raises_the_same_events():
    sc = S.many_pyramids(rows=1, cols=2)
    trig = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, 30.0, 0.0))
    tc = sc.add_collider(trig, half_extents=(40.0, 0.5, 40.0), active_events=S.ACTIVE_EVENTS_COLLISION, sensor=1)
    ball = sc.add_body(translation=(0.0, 34.0, 0.0))
    bc = sc.add_collider(ball, shape=S.SHAPE_BALL, half_extents=(0.3, 0.0, 0.0))
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    seen = []
    for k in range(90):
        g.step(2); o.step(2)
        ge = [tuple(int(x) for x in e) for e in g.collision_events()]
        oe = [tuple(int(x) for x in e) for e in o.collision_events()]
        assert ge == oe, (k, ge, oe)
        seen += ge
        _same_state(g, o, f"sensor world +{2 * (k + 1)}")
        assert g.intersection_pair(tc, bc) == o.intersection_pair(tc, bc)
    assert [e[2] for e in seen if tc in e[:2]] == [1, 0]            # the ball entered and left the trigger
    c = g.counters()
    assert c["fused_steps"] > 20 and c["overflow_flags"] == 0, c


def test_contact_force_events_on_the_fused_step():
    sc = S.many_pyramids(rows=2, cols=2).enable_events(S.ACTIVE_EVENTS_COLLISION | S.ACTIVE_EVENTS_CONTACT_FORCE, 20.0)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for k in range(8):
        g.step(25); o.step(25)
        _same_state(g, o, f"force events +{25 * (k + 1)}")
        gm, gv = g.contact_force_events(); om, ov = o.force_events()
        go = np.lexsort((gm[:, 1], gm[:, 0], gm[:, 2])) if len(gm) else np.zeros(0, int)
        oo = np.lexsort((om[:, 1], om[:, 0], om[:, 2])) if len(om) else np.zeros(0, int)
        np.testing.assert_array_equal(gm[go], om[oo]); np.testing.assert_array_equal(gv[go], ov[oo])
        assert len(gm) > 0
    assert g.counters()["fused_steps"] > 60



def test_several_fused_steps_per_launch_commit_and_abort_step_by_step():
    """Round 6: a world whose islands fit one per workgroup takes up to 32 fused steps in ONE launch (k_island_solve_steps) — every step
    still validates itself, arrives, and commits only when every workgroup did; the first aborted step ends the launch for all of them
    and the host replays the rest.  (a) a settled field of pyramids: launches of many steps, bit for bit the oracle; (b) a kick in the
    middle of a long batch: the launch stops at the step the kicked cube leaves its fat AABB, the steps behind it are replayed."""
    sc = S.many_pyramids(3, 3)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(90); o.step(90)
    c0 = g.counters()
    g.step(200); o.step(200)
    gp, gv = g.read_bodies(); op, ov = o.read()
    np.testing.assert_array_equal(gp, op); np.testing.assert_array_equal(gv, ov)
    c1 = g.counters()
    steps, launches = c1["fused_steps"] - c0["fused_steps"], c1["fused_launches"] - c0["fused_launches"]
    assert steps >= 150 and launches * 8 <= steps, (c0, c1)          # (many steps per launch)
    top = 55                                                           # the top cube of the first pyramid
    vel = np.zeros((1, 6), np.float32); vel[0, :3] = (3.0, 4.0, 0.5)
    g.write_bodies([top], vel6=vel); o.set_vel(top, vel[0, :3], vel[0, 3:])
    for n in (64, 1, 200):
        g.step(n); o.step(n)
        gp, gv = g.read_bodies(); op, ov = o.read()
        np.testing.assert_array_equal(gp, op, err_msg=f"kick +{n}"); np.testing.assert_array_equal(gv, ov, err_msg=f"kick +{n}")
    assert g.counters()["replayed_steps"] > c1["replayed_steps"]


def test_an_island_with_a_pair_into_another_island_takes_one_step_per_launch():
    """A pair without solver contacts is listed with the island of its first dynamic body; when its other body lives in ANOTHER island the
    validating workgroup reads a pose another workgroup writes — sound across a kernel boundary only.  Two stacks 0.05 apart (inside each
    other's fat AABBs, outside the contact prediction): the device finds the pair, the world goes back to one fused step per launch."""
    sc = S.Scene(name="near_stacks", gravity=(0.0, -9.81, 0.0))
    gb = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -1.0, 0.0)); sc.add_collider(gb, half_extents=(20.0, 1.0, 20.0))
    for x in (-0.525, 0.525):
        for k in range(3):
            b = sc.add_body(translation=(x, 0.5 + k, 0.0)); sc.add_collider(b, half_extents=(0.5, 0.5, 0.5), density=100.0)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for n in (80, 100, 150):
        g.step(n); o.step(n)
        gp, gv = g.read_bodies(); op, ov = o.read()
        np.testing.assert_array_equal(gp, op, err_msg=f"+{n}"); np.testing.assert_array_equal(gv, ov, err_msg=f"+{n}")
    c = g.counters()
    assert c["num_islands"] == 2 and c["num_pairs"] > c["num_manifolds"], c         # two islands, pairs without solver contacts between them
    assert c["fused_steps"] > 100 and c["fused_launches"] * 2 > c["fused_steps"], c  # fused, but (after the first long launch was cut short) one step per launch
