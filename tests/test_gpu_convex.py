"""Cylinders and cones on the device path (RP_SHAPE_CYLINDER / RP_SHAPE_CONE = ColliderBuilder::cylinder / cone, collider.rs:770, :789)
against the oracle, bit for bit: GJK / EPA + polygonal feature maps (contact_manifold_pfm_pfm), the ball and half-space arms of the
dispatcher, sensors, continuous collision detection, sleeping, compound bodies and insertion into a running world — the CONVEX
instantiations of the narrow-phase, sensor and CCD kernels (rp_convex.h).  What the oracle itself is pinned on: tests/test_convex_oracle.py."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld, lib

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


@pytest.mark.parametrize("ground", ["cuboid", "cylinder", "halfspace"])
def test_clutter_of_every_shape_bit_exact(ground):
    sc = S.convex_clutter(40, 3, ground)
    for step, g, o, ev in _lockstep(sc, 400, every=4):
        pass
    c = g.counters()
    assert c["num_manifolds"] == o.stats()["num_active_manifolds"] and c["num_manifolds"] > 40
    pos, _ = g.read_bodies()
    dyn = [i for i, b in enumerate(sc.bodies) if int(b["body_type"]) == S.BODY_DYNAMIC]
    assert pos[dyn, 1].min() > 0.1                                       # nothing fell through


@pytest.mark.parametrize("seed", [11, 12])
def test_clutter_other_seeds_with_contact_impulses(seed):
    sc = S.convex_clutter(30, seed, "cuboid")
    for step, g, o, ev in _lockstep(sc, 240, every=8):
        pass
    gm, gn, gi = g.contacts()
    om, on, oi = o.manifolds()
    gk = {(a, b): (c, n, tuple(i), tuple(nn)) for (a, b, c, n), i, nn in zip(gm.tolist(), gi.tolist(), gn.tolist())}
    ok = {(a, b): (c, n, tuple(i), tuple(nn)) for (a, b, c, n), i, nn in zip(om.tolist(), oi.tolist(), on.tolist())}
    assert gk == ok and len(gk) > 20          # same manifolds, colours, normals, point counts and impulses


def test_issue_810_on_the_device():
    """crates/rapier3d/tests/issue_810_cubes_thin_cylinder_tunnel.rs: the cubes stay on the disc — and take the oracle's path bit for bit
    (fast bodies: the continuous-collision pass against the disc runs for every cube on its way down)"""
    sc = S.issue_810_disc()
    for step, g, o, ev in _lockstep(sc, 600, every=10):
        pos, _ = g.read_bodies()
        below = pos[1:, 1] < -1.95 - 0.5
        assert not (below & (np.hypot(pos[1:, 0], pos[1:, 2]) <= 9.8)).any(), step
    assert (np.abs(pos[1:, 1] + 1.95) < 0.2).sum() >= 15
    c = g.counters()
    assert c["ccd_active_count"] > 0 and c["num_solver_contacts"] == o.stats()["num_solver_contacts"]


def test_sleeping_clutter_bit_exact():
    sc = S.convex_clutter(24, 5, "cuboid").enable_sleep()
    for step, g, o, ev in _lockstep(sc, 900, every=30, sleeping=True):
        pass
    assert g.sleeping().any()


def test_sensor_cylinder_and_cone_regions():
    """a sensor cylinder and a sensor cone (GJK intersection tests) crossed by a ball, a box, a capsule and a cylinder"""
    sc = S.Scene(name="convex_sensors", gravity=(0.0, -9.81, 0.0))
    z0 = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, 2.0, 0.0))
    s0 = sc.add_collider(z0, shape=S.SHAPE_CYLINDER, half_extents=(0.5, 3.0, 0.0), sensor=1, active_events=S.ACTIVE_EVENTS_COLLISION)
    z1 = sc.add_body(body_type=S.BODY_FIXED, translation=(8.0, 2.0, 0.0))
    s1 = sc.add_collider(z1, shape=S.SHAPE_CONE, half_extents=(1.0, 2.0, 0.0), sensor=1, active_events=S.ACTIVE_EVENTS_COLLISION)
    gr = sc.add_body(body_type=S.BODY_FIXED, translation=(4.0, -0.5, 0.0)); sc.add_collider(gr, half_extents=(12.0, 0.5, 6.0))
    shapes = [(S.SHAPE_BALL, (0.3, 0, 0)), (S.SHAPE_CUBOID, (0.3, 0.2, 0.25)), (S.SHAPE_CAPSULE, (0.3, 0.15, 0.0)), (S.SHAPE_CYLINDER, (0.25, 0.3, 0.0))]
    for k, (shape, he) in enumerate(shapes):
        for x0 in (0.0, 8.0):
            b = sc.add_body(translation=(x0 - 1.5 + 1.0 * k, 5.0 + 0.7 * k, 0.3 * k), rotation=(0.2, 0.1, 0.3, 0.9273618), angvel=(1.0, 0.0, 2.0))
            sc.add_collider(b, shape=shape, half_extents=he, active_events=S.ACTIVE_EVENTS_COLLISION)
    n_sensor = 0
    for step, g, o, ev in _lockstep(sc, 200):
        n_sensor += sum(1 for e in ev if e[3] & 1)
    assert n_sensor >= 12                                               # every faller entered and left its region


def test_ccd_fast_cylinder_and_cone_bullets():
    """fast support-mapped bodies against a thin floor and against a cone: the conservative advancement runs on GJK distances"""
    sc = S.Scene(name="convex_ccd", gravity=(0.0, -9.81, 0.0))
    fl = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.02, 0.0)); sc.add_collider(fl, half_extents=(100.0, 0.02, 100.0))
    pil = sc.add_body(body_type=S.BODY_FIXED, translation=(6.0, 1.0, 0.0)); sc.add_collider(pil, shape=S.SHAPE_CONE, half_extents=(1.0, 0.8, 0.0))
    disc = sc.add_body(body_type=S.BODY_FIXED, translation=(-6.0, 1.0, 0.0)); sc.add_collider(disc, shape=S.SHAPE_CYLINDER, half_extents=(0.03, 2.0, 0.0))
    b0 = sc.add_body(translation=(0.0, 6.0, 0.0), linvel=(0.0, -150.0, 0.0), rotation=(0.3, 0.0, 0.2, 0.9327379), angvel=(3.0, 0.0, 1.0))
    sc.add_collider(b0, shape=S.SHAPE_CYLINDER, half_extents=(0.08, 0.1, 0.0))
    b1 = sc.add_body(translation=(6.0, 9.0, 0.1), linvel=(0.0, -200.0, 0.0))
    sc.add_collider(b1, shape=S.SHAPE_CONE, half_extents=(0.1, 0.08, 0.0))
    b2 = sc.add_body(translation=(-6.3, 8.0, 0.2), linvel=(0.0, -180.0, 0.0), ccd_enabled=1)
    sc.add_collider(b2, half_extents=(0.05, 0.05, 0.05))
    b3 = sc.add_body(translation=(3.0, 5.0, 3.0), linvel=(0.0, -120.0, 0.0), ccd_enabled=1)
    sc.add_collider(b3, shape=S.SHAPE_BALL, half_extents=(0.06, 0.0, 0.0))
    tgt = sc.add_body(translation=(3.0, 0.3, 3.0)); sc.add_collider(tgt, shape=S.SHAPE_CYLINDER, half_extents=(0.3, 0.5, 0.0))
    for step, g, o, ev in _lockstep(sc, 120):
        pass
    pos, _ = g.read_bodies()
    assert pos[[b0, b1, b2, b3], 1].min() > -0.01                       # nobody went through the 4 cm floor
    assert g.counters()["ccd_clamp_count"] >= 3


def test_a_cylinder_inserted_into_a_running_cuboid_world():
    """the world starts without any support-mapped shape (the plain kernels), then receives a cylinder and a cone: has_convex switches
    on in place, the step graphs are captured again with the CONVEX kernels"""
    sc = S.box_stack(4)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(30); o.step(30)
    for shape, he, pos in ((S.SHAPE_CYLINDER, (0.4, 0.3, 0.0), (0.1, 6.0, 0.05)), (S.SHAPE_CONE, (0.4, 0.35, 0.0), (1.5, 3.0, 0.0))):
        body = S.body_desc(translation=pos, rotation=(0.1, 0.2, 0.3, 0.9273618))
        col = S.collider_desc(shape=shape, half_extents=he, density=2.0)
        hb = g.insert_body(body); g.insert_collider(col, hb)
        ob = lib().ro_add_body(o._w, np.array([body], S.BODY_DTYPE).ctypes.data)
        lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
        for n in (1, 20, 100):
            g.step(n); o.step(n)
            gp, gv = g.read_bodies(); op, ov = o.read()
            np.testing.assert_array_equal(gp, op); np.testing.assert_array_equal(gv, ov)
    assert g.counters()["num_manifolds"] == o.stats()["num_active_manifolds"]


def test_rest_heights_on_the_device():
    s2 = float(np.sin(np.pi / 4))
    sc = S.Scene(name="rest", gravity=(0.0, -9.81, 0.0))
    gr = sc.add_body(body_type=S.BODY_FIXED, translation=(0, -0.5, 0)); sc.add_collider(gr, half_extents=(20, 0.5, 5))
    want = []
    for k, (shape, rot, rest) in enumerate([(S.SHAPE_CYLINDER, (0, 0, 0, 1), 0.5), (S.SHAPE_CYLINDER, (0, 0, s2, s2), 0.3), (S.SHAPE_CONE, (0, 0, 0, 1), 0.5)]):
        b = sc.add_body(translation=(3.0 * k - 3.0, 1.0, 0.0), rotation=rot)
        sc.add_collider(b, shape=shape, half_extents=(0.5, 0.3, 0))
        want.append((b, rest))
    for step, g, o, ev in _lockstep(sc, 300, every=50):
        pass
    pos, vel = g.read_bodies()
    for b, rest in want:
        assert abs(pos[b][1] - rest) < 2.5e-3 and abs(vel[b][1]) < 1e-2


def test_unknown_shape_and_bad_dimensions_are_refused():
    from rapier_amd.world import RapierHipError
    sc = S.box_stack(1)
    g = PhysicsWorld.from_scene(sc)
    hb = g.insert_body(S.body_desc(translation=(0, 5, 0)))
    with pytest.raises(RapierHipError):
        g.insert_collider(S.collider_desc(shape=6, half_extents=(0.5, 0.5, 0.5)), hb)
    with pytest.raises(RapierHipError):
        g.insert_collider(S.collider_desc(shape=S.SHAPE_CONE, half_extents=(0.5, 0.0, 0.0)), hb)


def test_round_clutter_bit_exact():
    """ColliderBuilder::round_cuboid / round_cylinder / round_cone / round_convex_hull (parry RoundShape<S>): the inner shapes' GJK / EPA
    manifolds with border radii, against every other shape, a round-cuboid ground and a half-space ramp"""
    sc = S.round_clutter(30, 6)
    for step, g, o, ev in _lockstep(sc, 400, every=4):
        pass
    c = g.counters()
    assert c["num_manifolds"] == o.stats()["num_active_manifolds"] and c["num_manifolds"] > 30
    pos, _ = g.read_bodies()
    dyn = [i for i, b in enumerate(sc.bodies) if int(b["body_type"]) == S.BODY_DYNAMIC]
    assert pos[dyn, 1].min() > 0.1


def test_round_shapes_sleeping_sensors_and_ccd():
    sc = S.Scene(name="round_misc", gravity=(0.0, -9.81, 0.0))
    fl = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.25, 0.0)); sc.add_collider(fl, half_extents=(60.0, 0.25, 60.0))
    z = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, 2.0, 0.0))
    sc.add_collider(z, shape=S.SHAPE_ROUND_CUBOID, half_extents=(1.5, 0.5, 1.5), border_radius=0.3, sensor=1, active_events=S.ACTIVE_EVENTS_COLLISION)
    for k, (shape, he) in enumerate([(S.SHAPE_BALL, (0.3, 0, 0)), (S.SHAPE_ROUND_CUBOID, (0.2, 0.15, 0.2)), (S.SHAPE_ROUND_CONE, (0.3, 0.25, 0.0)), (S.SHAPE_CAPSULE, (0.3, 0.15, 0.0))]):
        b = sc.add_body(translation=(-1.2 + 0.8 * k, 5.0 + 0.8 * k, 0.2 * k), rotation=(0.2, 0.1, 0.3, 0.9273618), angvel=(1.0, 0.0, 2.0), can_sleep=1)
        sc.add_collider(b, shape=shape, half_extents=he, border_radius=0.06 if shape >= S.SHAPE_ROUND_CUBOID else 0.0, active_events=S.ACTIVE_EVENTS_COLLISION)
    fast = []
    for k in range(2):
        b = sc.add_body(translation=(6.0 + 1.5 * k, 7.0 + k, 0.1 * k), linvel=(0.0, -60.0 - 10.0 * k, 0.0), rotation=(0.3, 0.0, 0.2, 0.9327379), ccd_enabled=k)
        sc.add_collider(b, shape=S.SHAPE_ROUND_CYLINDER, half_extents=(0.06, 0.08, 0.0), border_radius=0.02, density=4.0)
        fast.append(b)
    n_sensor = 0
    for step, g, o, ev in _lockstep(sc, 600, every=2, sleeping=True):
        n_sensor += sum(1 for e in ev if e[3] & 1)
    assert n_sensor >= 6 and g.sleeping().any()
    pos, _ = g.read_bodies()
    assert pos[fast, 1].min() > 0.0 and g.counters()["ccd_clamp_count"] >= 1


def test_fountain_churn_with_the_references_shapes_in_lockstep():
    """solver_graph_stale_refs.rs:24-79 on the device: a round cylinder, a cone or a cuboid spawned every step, the outermost bodies
    removed beyond 120 (arena slots reused) — 400 steps, every state compared with the oracle's"""
    sc = S.Scene(name="fountain", gravity=(0.0, -9.81, 0.0))
    gnd = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -2.1, 0.0)); sc.add_collider(gnd, half_extents=(40.0, 2.1, 40.0))
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    alive, rad = [], 0.5          # (device handle, oracle index)
    for step_id in range(1, 400):
        g.step(1); o.step(1)
        body = S.body_desc(translation=(0.0, 10.0, 0.0), can_sleep=1)
        if step_id % 3 == 0:
            col = S.collider_desc(shape=S.SHAPE_ROUND_CYLINDER, half_extents=(rad, rad, 0.0), border_radius=rad / 10.0)
        elif step_id % 3 == 1:
            col = S.collider_desc(shape=S.SHAPE_CONE, half_extents=(rad, rad, 0.0))
        else:
            col = S.collider_desc(half_extents=(rad, rad, rad))
        hb = g.insert_body(body); g.insert_collider(col, hb)
        ob = lib().ro_add_body(o._w, np.array([body], S.BODY_DTYPE).ctypes.data)
        lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
        assert int(hb) & 0xFFFFFFFF == ob
        alive.append((hb, ob))
        if len(alive) + 1 > 120:
            pos, _ = o.read()
            order = sorted(alive, key=lambda h: -(abs(pos[h[1], 0]) + abs(pos[h[1], 2])))
            for h in order[:len(alive) + 1 - 120]:
                g.remove_body([h[0]]); o.remove_body(h[1]); alive.remove(h)
        if step_id % 10 == 0 or step_id > 380:
            gp, gv = g.read_bodies([h for h, _ in alive]); op, ov = o.read()
            idx = [b for _, b in alive]
            np.testing.assert_array_equal(gp, op[idx], err_msg=f"step {step_id}"); np.testing.assert_array_equal(gv, ov[idx], err_msg=f"step {step_id}")
    g.step(1); o.step(1)          # (the pairs of the bodies removed last leave the oracle's pair set with its next broad-phase pass)
    assert g.counters()["overflow_flags"] == 0 and g.counters()["num_pairs"] == o.stats()["num_pairs"]


def test_thousands_of_tiny_islands_take_the_global_path_bit_exact(monkeypatch):
    """S.shapes_rain: 3,000 bodies of all ten shape kinds landing on a slab = far more than 960 islands of a few bodies each, which the
    layout rebuild then leaves on the global path (rp_islands.hip, lay_isl_number); the oracle, and the same world with
    RP_NO_TINY_ROUTING=1 (every island its own workgroup), must agree bit for bit"""
    sc = S.shapes_rain(3000)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    monkeypatch.setenv("RP_NO_TINY_ROUTING", "1")
    h = PhysicsWorld.from_scene(sc); h.read_bodies()            # (the device world — and with it the switch — is built by the first call that needs it)
    monkeypatch.delenv("RP_NO_TINY_ROUTING")
    for cp in (20, 60, 120, 200):
        n = cp - (0 if cp == 20 else {60: 20, 120: 60, 200: 120}[cp])
        g.step(n); o.step(n); h.step(n)
        gp, gv = g.read_bodies(); op, ov = o.read(); hp, hv = h.read_bodies()
        np.testing.assert_array_equal(gp, op, err_msg=f"step {cp}"); np.testing.assert_array_equal(gv, ov, err_msg=f"step {cp}")
        np.testing.assert_array_equal(gp, hp, err_msg=f"step {cp} (routing)"); np.testing.assert_array_equal(gv, hv, err_msg=f"step {cp} (routing)")
    c = g.counters()
    assert c["overflow_flags"] == 0 and c["num_manifolds"] == o.stats()["num_active_manifolds"] > 2000
