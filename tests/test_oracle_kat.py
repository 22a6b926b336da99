"""CPU tests that PIN THE ORACLE against the reference's own known-answer tests.

The Rust reference cannot be built here, so its bitwise goldens are out of reach; what the
reference's tests pin without bit-exactness is restated below, one test per reference test.
"""
import numpy as np
import pytest

from rapier_amd import scenes as S
from oracle_ffi import OracleWorld, lib


# /root/reference/src/dynamics/coefficient_combine_rule.rs:60-96 + rule priority (:58-86)
def test_combine_rule_vectors():
    L = lib()
    assert L.ro_combine_coefficient(0.25, 1.0, S.RULE_GEOMETRIC_MEAN, S.RULE_GEOMETRIC_MEAN) == 0.5
    assert L.ro_combine_coefficient(0.0, 5.0, S.RULE_GEOMETRIC_MEAN, S.RULE_GEOMETRIC_MEAN) == 0.0
    assert L.ro_combine_coefficient(-1.0, 4.0, S.RULE_GEOMETRIC_MEAN, S.RULE_GEOMETRIC_MEAN) == 0.0  # clamped, no NaN
    assert L.ro_combine_coefficient(0.5, 0.7, S.RULE_AVERAGE, S.RULE_AVERAGE) == pytest.approx(0.6)
    assert L.ro_combine_coefficient(0.5, -0.7, S.RULE_MIN, S.RULE_AVERAGE) == pytest.approx(0.7)  # |min|
    assert L.ro_combine_coefficient(0.5, 0.7, S.RULE_MULTIPLY, S.RULE_MIN) == pytest.approx(0.35)  # stronger rule wins
    assert L.ro_combine_coefficient(0.5, 0.7, S.RULE_MAX, S.RULE_AVERAGE) == pytest.approx(0.7)
    assert L.ro_combine_coefficient(0.8, 0.7, S.RULE_CLAMPED_SUM, S.RULE_MAX) == 1.0


# /root/reference/crates/rapier3d/tests/total_contact_impulse.rs:13-75
@pytest.mark.parametrize("cuboid", [True, False])
@pytest.mark.parametrize("coeff", [1.0, 0.5, 0.0])
def test_resting_impulse_matches_gravity_for_any_warmstart_coefficient(cuboid, coeff):
    sc = S.Scene(name="kat1", gravity=(0.0, -9.81, 0.0))
    sc.params["warmstart_coefficient"] = coeff
    sc.add_collider(-1, half_extents=(10.0, 0.5, 10.0), translation=(0.0, -0.5, 0.0))
    b = sc.add_body(translation=(0.0, 0.5, 0.0), additional_mass=1.0)
    if cuboid:
        sc.add_collider(b, half_extents=(0.5, 0.5, 0.5), density=0.0)
    else:
        sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0, 0), density=0.0)
    w = OracleWorld(sc)
    w.step(300)
    expected = 1.0 * 9.81 / 60.0
    assert abs(w.total_contact_impulse() - expected) <= expected * 1.0e-2


# /root/reference/src/geometry/broad_phase_bvh/mod.rs:281-329
def test_ball_rests_on_floor():
    sc = S.Scene(name="kat2", gravity=(0.0, -9.81, 0.0))
    sc.add_collider(-1, half_extents=(10.0, 0.5, 10.0))
    b = sc.add_body(translation=(0.0, 4.0, 0.0))
    sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0, 0))
    w = OracleWorld(sc)
    w.step(200)
    y = w.read()[0][b, 1]
    assert abs(y - 1.0) < 0.02


# The default-cadence stack of /root/reference/src/pipeline/physics_pipeline/test_staged.rs:86-148:
# three cubes at y = 1.001 i on a fixed unit cube; they must come to rest stacked and finite.
def test_three_cube_stack_rests():
    sc = S.Scene(name="kat3", gravity=(0.0, -9.81, 0.0))
    g = sc.add_body(body_type=S.BODY_FIXED)
    sc.add_collider(g, half_extents=(0.5, 0.5, 0.5))
    ids = []
    for i in (1, 2, 3):
        b = sc.add_body(translation=(0.0, 1.001 * i, 0.0))
        sc.add_collider(b, half_extents=(0.5, 0.5, 0.5))
        ids.append(b)
    w = OracleWorld(sc)
    w.step(60)
    pos, vel = w.read()
    assert np.isfinite(pos).all()
    for i, b in zip((1, 2, 3), ids):
        assert abs(pos[b, 1] - float(i)) < 0.05
        assert abs(pos[b, 0]) < 0.05 and abs(pos[b, 2]) < 0.05
    assert np.abs(vel).max() < 0.05


# Scene-size facts the survey derives from the reference examples: N and M.
def test_scene_sizes_match_reference_formulas():
    s = S.many_pyramids()
    assert s.num_dynamic == 10780 and len(s.bodies) == 10781
    assert S.large_pyramid().num_dynamic == 20100
    jg = S.joint_grid()
    assert jg.num_dynamic == 9900 and len(jg.joints) == 19800
    w = OracleWorld(S.pyramid10())
    w.step(1)
    st = w.stats()
    assert st["num_active_manifolds"] == 145 and st["num_solver_contacts"] == 580


# A settled pyramid carries its whole weight on the ground contacts: sum = N m g dt.
def test_pyramid_ground_impulse_equals_weight():
    w = OracleWorld(S.pyramid10())
    w.step(300)
    meta, nrm, imp = w.manifolds()
    ground = meta[:, 2] == 127  # dynamic-fixed pairs take the top colour (narrow_phase/mod.rs:122-131)
    assert ground.sum() == 10
    expected = 55 * 100.0 * 10.0 / 60.0
    assert abs(imp[ground].sum() - expected) <= expected * 1e-3
    np.testing.assert_allclose(nrm[ground], np.tile([0.0, 1.0, 0.0], (10, 1)), atol=1e-5)


# Colour contract: same-colour manifolds never share a dynamic body.
def test_colours_are_body_disjoint():
    sc = S.tumble(48, seed=3)
    w = OracleWorld(sc)
    parents = sc.parent_array()
    dyn = np.array([int(b["body_type"]) == S.BODY_DYNAMIC for b in sc.bodies])
    for _ in range(6):
        w.step(20)
        meta, _, _ = w.manifolds()
        for color in np.unique(meta[:, 2]):
            if color >= 128:
                continue
            seen = set()
            for c1, c2, _, _ in meta[meta[:, 2] == color]:
                for b in (parents[c1], parents[c2]):
                    if b >= 0 and dyn[b]:
                        assert b not in seen, f"colour {color} reuses body {b}"
                        seen.add(b)


# Restitution: a bouncy ball must rebound (issue_974_restitution.rs outcome), an inelastic one must not.
@pytest.mark.parametrize("restitution,should_bounce", [(0.8, True), (0.0, False)])
def test_restitution_bounce(restitution, should_bounce):
    sc = S.Scene(name="bounce", gravity=(0.0, -9.81, 0.0))
    sc.add_collider(-1, half_extents=(10.0, 0.5, 10.0), restitution=restitution)
    b = sc.add_body(translation=(0.0, 3.0, 0.0))
    sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0, 0), restitution=restitution)
    w = OracleWorld(sc)
    max_up = 0.0
    for _ in range(120):
        w.step(1)
        max_up = max(max_up, float(w.read()[1][b, 1]))
    assert (max_up > 2.0) == should_bounce


# Speed caps (speed_cap.rs:66-180): linear speed is clamped to max_linear_velocity each substep.
def test_linear_speed_cap():
    sc = S.Scene(name="cap", gravity=(0.0, 0.0, 0.0))
    b = sc.add_body(translation=(0.0, 0.0, 0.0), linvel=(1000.0, 0.0, 0.0))
    sc.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    w = OracleWorld(sc)
    w.step(1)
    assert abs(np.linalg.norm(w.read()[1][b, :3]) - 400.0) < 1e-2


# /root/reference/crates/rapier3d/tests/joint_stability.rs:105-175 (joint_net_remains_stable)
def test_spherical_joint_net_remains_stable():
    w = OracleWorld(S.joint_net(32))
    for _ in range(4):
        w.step(1000)
        pos, vel = w.read()
        assert np.linalg.norm(pos[:, :3], axis=1).max() < 500.0
        assert np.linalg.norm(vel[:, :3], axis=1).max() < 100.0


def test_spherical_joint_holds_its_anchors():
    """A swinging chain keeps every pair of joint anchors together (bilateral lock rows)."""
    sc = S.joint_chain(8)
    w = OracleWorld(sc)

    def rot(q, x):
        b, ww = q[:3], q[3]
        return x * (ww * ww - b @ b) + b * (x @ b) * 2 + np.cross(b, x) * ww * 2

    for _ in range(6):
        w.step(50)
        pos, _ = w.read()
        for j in sc.joints:
            b1, b2 = int(j["body1"]), int(j["body2"])
            a1 = pos[b1, :3] + rot(pos[b1, 3:], np.array(j["local_anchor1"]))
            a2 = pos[b2, :3] + rot(pos[b2, 3:], np.array(j["local_anchor2"]))
            assert np.linalg.norm(a1 - a2) < 0.05


def test_joint_grid_scene_matches_reference_formula():
    sc = S.joint_grid(100)
    assert len(sc.bodies) == 10000 and len(sc.joints) == 19800 and sc.num_dynamic == 9900


def test_oracle_threads_do_not_change_results():
    """The OpenMP loops only cover body-disjoint work, so any thread count gives the same bits."""
    import oracle_ffi
    res = []
    for threads in (1, 4):
        oracle_ffi.set_threads(threads)
        try:
            out = []
            import test_gpu_fuzz as F       # the randomised feature-mix scenes (sleeping, joints, events, both friction models, piles)
            fuzz = [F._scene(seed)[0] for seed in (1, 3, 6)] + [F._scene(1000, n=300, spread=2.2, per_layer=49, calm=True)[0]]
            for sc in [S.tumble(40, seed=3), S.joint_grid(12), S.many_pyramids(rows=2, cols=2), S.capsules(6), S.motorised_joints()] + fuzz:
                w = OracleWorld(sc)
                w.step(60)
                out.append(w.read() + (w.sleeping(),))
            res.append(out)
        finally:
            oracle_ffi.set_threads(1)
    for (p1, v1, s1), (p4, v4, s4) in zip(*res):
        np.testing.assert_array_equal(p1, p4)
        np.testing.assert_array_equal(v1, v4)
        np.testing.assert_array_equal(s1, s4)


# FrictionModel::Coulomb (contact_with_coulomb_friction.rs): a sliding box decelerates at mu * g under either
# friction model, and a resting pyramid carries its weight (total_contact_impulse.rs restated for the twin).
@pytest.mark.parametrize("model", [S.FRICTION_SIMPLIFIED, S.FRICTION_COULOMB])
def test_sliding_box_decelerates_at_mu_g(model):
    sc = S.box_stack(1)
    sc.params["friction_model"] = model
    sc.bodies[1]["linvel"] = (3.0, 0.0, 0.0)
    w = OracleWorld(sc)
    w.step(20)
    _, v = w.read()
    mu, g, t = 0.5, 9.81, 20.0 / 60.0
    assert v[1, 0] == pytest.approx(3.0 - mu * g * t, abs=0.15)
    w.step(60)
    _, v = w.read()
    assert abs(v[1, 0]) < 1e-3                # came to rest: static friction holds


def test_coulomb_pyramid_rests_and_carries_its_weight():
    sc = S.pyramid10()
    sc.params["friction_model"] = S.FRICTION_COULOMB
    w = OracleWorld(sc)
    w.step(300)
    pos, vel = w.read()
    assert np.abs(vel).max() < 1e-3 and pos[1:, 1].max() == pytest.approx(9.5, abs=0.02)
    meta, nrm, imp = w.manifolds()
    ground = meta[:, 0] == 0
    weight = 55 * 100.0 * 10.0 / 60.0         # 55 unit cubes of density 100, g = 10, dt = 1/60
    assert imp[ground].sum() == pytest.approx(weight, rel=0.01)


# LockedAxes (rigid_body_components.rs:533-571): a fully locked body never moves, rotation locks keep the orientation,
# a translation lock keeps that coordinate.
def test_locked_axes():
    sc = S.locked_axes_scene()
    w = OracleWorld(sc)
    p0, _ = w.read()
    w.step(240)
    p, v = w.read()
    np.testing.assert_array_equal(p[4], p0[4])                                  # 0x3F: fully locked, even when hit
    np.testing.assert_array_equal(p[2, 3:], p0[2, 3:])                          # 0x38: rotations locked
    assert p[3, 2] == p0[3, 2]                                                  # 0x1c: z translation locked
    np.testing.assert_array_equal(p[5, :3], p0[5, :3])                          # 0x07: no translation
    assert p[1, 1] < p0[1, 1] and np.isfinite(p).all()                          # the free one fell


# Compound bodies: the summed MassProperties of several colliders (parallel-axis theorem + diagonalisation) against a
# float64 reference, and a hammer that comes to rest head-down-ish without gaining energy.
def test_compound_mass_properties_and_rest():
    sc = S.Scene(name="cmp")
    b = sc.add_body(translation=(0, 5, 0))
    q = np.array([0.1, 0.2, 0.3, 0.9]); q /= np.linalg.norm(q)
    sc.add_collider(b, half_extents=(0.5, 0.25, 0.75), translation=(1.0, 0.2, -0.3), rotation=tuple(q), density=2.0)
    sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.4, 0, 0), translation=(-0.7, 0.0, 0.5), density=3.0)
    sc.add_collider(b, half_extents=(0.2, 0.9, 0.1), translation=(0.0, -0.8, 0.0), density=1.0)
    mp = OracleWorld(sc).mass_props(b).astype(np.float64)

    def rot(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    def box(he, rho):
        he = np.array(he); m = 8 * he.prod() * rho
        return m, np.diag([m / 3 * (he[1] ** 2 + he[2] ** 2), m / 3 * (he[0] ** 2 + he[2] ** 2), m / 3 * (he[0] ** 2 + he[1] ** 2)])
    m1, i1 = box((0.5, 0.25, 0.75), 2.0); r1 = rot(q)
    m2 = 4 / 3 * np.pi * 0.4 ** 3 * 3.0; i2 = np.eye(3) * 0.4 * m2 * 0.16
    m3, i3 = box((0.2, 0.9, 0.1), 1.0)
    parts = [(m1, np.array([1.0, 0.2, -0.3]), r1 @ i1 @ r1.T), (m2, np.array([-0.7, 0, 0.5]), i2), (m3, np.array([0, -0.8, 0.0]), i3)]
    M = sum(p[0] for p in parts); com = sum(p[0] * p[1] for p in parts) / M
    tensor = sum(I + m * ((com - c) @ (com - c) * np.eye(3) - np.outer(com - c, com - c)) for m, c, I in parts)
    assert mp[0] == pytest.approx(1 / M, rel=1e-6)
    np.testing.assert_allclose(mp[1:4], com, atol=1e-6)
    R = rot(mp[7:11])
    np.testing.assert_allclose(R @ np.diag(1 / mp[4:7]) @ R.T, tensor, atol=5e-6)
    w = OracleWorld(S.compound_bodies(4))
    w.step(400)
    pos, vel = w.read()
    assert np.isfinite(pos).all() and np.abs(vel).max() < 0.05 and pos[1:, 1].min() > 0.05


# test_staged.rs:86-148 scene: the 3-cube stack AND the elevated pair joined by a revolute joint about Z rest after 60
# steps; a hinge keeps its axis, a fixed joint keeps the relative pose (lock_angular, joint_constraint_helper.rs:628-673).
def test_staged_scene_with_revolute_pair_and_locked_angular_axes():
    sc = S.jointed_pairs(1)
    w = OracleWorld(sc)
    w.step(60)
    pos, vel = w.read()
    np.testing.assert_allclose(pos[1:4, 1], [0.5, 1.5, 2.5], atol=0.05)        # the stack rests (same bar as the reference test)
    a, b = 4, 5
    assert np.linalg.norm(pos[b, :3] - pos[a, :3]) == pytest.approx(1.5, abs=2e-3)   # revolute anchors coincide
    assert abs(pos[a, 2]) < 0.01 and abs(pos[b, 2]) < 0.01                     # the pair stays in its plane
    w.step(140)
    pos, vel = w.read()
    door, w1, w2 = 7, 8, 9
    assert pos[door, 1] == pytest.approx(1.5, abs=1e-3) and abs(vel[door, 3]) < 1e-3 and abs(vel[door, 5]) < 1e-3
    assert np.linalg.norm(pos[door, :3] - np.array([-5.0, 1.5, 0.0])) == pytest.approx(1.0, abs=2e-3)
    assert np.linalg.norm(pos[w2, :3] - pos[w1, :3]) == pytest.approx(1.2, abs=2e-3)
    np.testing.assert_allclose(pos[w1, 3:], pos[w2, 3:], atol=2e-3)            # welded: same orientation


# Joint limits (limit_linear / limit_angular, joint_constraint_helper.rs:166-208, 468-564): a slider drops to its lower stop,
# a door pushed into its stop stays at the limit angle, a pendulum cannot swing past its range.
def test_joint_limits_stop_at_their_bounds():
    w = OracleWorld(S.limited_joints())
    w.step(200)
    pos, vel = w.read()
    assert pos[2, 1] == pytest.approx(5.0 - 1.5, abs=5e-3)                       # prismatic lower limit -1.5 along the rail
    assert abs(pos[2, 0]) < 1e-3 and abs(pos[2, 2]) < 1e-3
    yaw = 2.0 * np.arctan2(pos[4, 4], pos[4, 6])
    assert yaw == pytest.approx(0.6, abs=5e-3) and abs(vel[4, 4]) < 1e-2          # revolute upper limit +0.6 rad
    roll = 2.0 * np.arctan2(pos[6, 5], pos[6, 6])
    assert roll == pytest.approx(-0.8, abs=5e-3)                                  # pendulum lower limit -0.8 rad


# Joint motors (motor_linear / motor_angular, joint_constraint_helper.rs:285-331, 566-625; MotorModel::combine_coefficients,
# motor_model.rs:39-58): a velocity motor reaches its target speed, a force-based position motor holds its target within what
# its force cap allows, a capped wheel motor drives a cart at a steady speed, motor impulses stay inside +-max_force * dt.
def test_joint_motors_reach_their_targets():
    sc = S.motorised_joints()
    w = OracleWorld(sc)
    w.step(300)
    pos, vel = w.read()
    assert vel[2, 5] == pytest.approx(3.0, abs=1e-3) and np.abs(vel[2, :5]).max() < 1e-4   # wheel: 3 rad/s about its axle
    # lift: spring 400 N/m towards -0.5 against its weight m g (m = 0.768 kg): rests at -0.5 - m g / k
    m = 8 * 0.4 * 0.2 * 0.4 * 3.0
    assert pos[4, 1] - 4.0 == pytest.approx(-0.5 - m * 9.81 / 400.0, abs=2e-3) and abs(vel[4, 1]) < 1e-3
    assert 1.0 < vel[7, 0] < 2.0 and abs(vel[7, 2]) < 1e-3                               # the cart rolls along +x
    imp = w.joint_motor_impulses()
    dt_sub = float(sc.params["dt"]) / int(sc.params["num_solver_iterations"])
    assert abs(imp[1, 0]) <= 60.0 * dt_sub * 1.0001 and np.abs(imp[3:5, 3]).max() <= 5.0 * dt_sub * 1.0001
    # a motor with a tiny force cap cannot hold the lift: it sags onto its lower stop
    w.set_joint_motor(1, 0, target_pos=-0.5, stiffness=400.0, damping=40.0, max_force=1.0, model=S.MOTOR_FORCE_BASED)
    w.step(200)
    assert w.read()[0][4, 1] - 4.0 == pytest.approx(-2.0, abs=5e-3)


# Capsules (ColliderBuilder::capsule_x/y/z; parry MassProperties::from_capsule, Capsule::aabb, contact_manifold_capsule_capsule /
# cuboid_capsule / convex_ball): analytic mass properties, rest heights, and the weight carried by the contacts.
def test_capsule_mass_properties_and_rest_poses():
    sc = S.Scene(name="caps", gravity=(0.0, -9.81, 0.0))
    g = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
    sc.add_collider(g, half_extents=(20.0, 0.5, 20.0))
    hh, r = 0.9, 0.35
    lying = sc.add_body(translation=(0.0, 1.0, 0.0)); sc.add_collider(lying, shape=S.SHAPE_CAPSULE, half_extents=(hh, r, 0.0))
    standing = sc.add_body(translation=(4.0, 2.0, 0.0)); sc.add_collider(standing, shape=S.SHAPE_CAPSULE, half_extents=(hh, r, 1.0))
    cross = sc.add_body(translation=(0.0, 2.0, 0.0)); sc.add_collider(cross, shape=S.SHAPE_CAPSULE, half_extents=(hh, r, 2.0))  # lands across `lying`
    ball = sc.add_body(translation=(8.0, 0.3, 0.0)); sc.add_collider(ball, shape=S.SHAPE_BALL, half_extents=(0.3, 0.0, 0.0))
    over = sc.add_body(translation=(8.0, 1.5, 0.0)); sc.add_collider(over, shape=S.SHAPE_CAPSULE, half_extents=(0.5, 0.2, 2.0))  # balances on the ball
    w = OracleWorld(sc)
    cyl, sph = np.pi * r * r * 2 * hh, 4.0 / 3.0 * np.pi * r ** 3
    i_axis = cyl * r * r / 2 + sph * 0.4 * r * r
    i_off = cyl * (3 * r * r + 4 * hh * hh) / 12 + sph * 0.4 * r * r + sph * (hh * hh + 0.75 * hh * r)
    for b, frame in ((lying, (0, 0, -np.sqrt(0.5), np.sqrt(0.5))), (standing, (0, 0, 0, 1)), (cross, (np.sqrt(0.5), 0, 0, np.sqrt(0.5)))):
        mp = w.mass_props(b)
        assert 1.0 / mp[0] == pytest.approx(cyl + sph, rel=1e-5)
        np.testing.assert_allclose(1.0 / mp[4:7], (i_off, i_axis, i_off), rtol=1e-5)
        np.testing.assert_allclose(mp[7:], frame, atol=1e-6)
    w.step(300)
    pos, vel = w.read()
    assert pos[lying, 1] == pytest.approx(r, abs=5e-3) and pos[standing, 1] == pytest.approx(hh + r, abs=5e-3)
    assert pos[cross, 1] == pytest.approx(3 * r, abs=1e-2)                       # resting across the lying capsule
    assert pos[over, 1] == pytest.approx(0.6 + 0.2, abs=1e-2)                    # on top of the ball (radius 0.3)
    assert np.abs(vel[[lying, standing]]).max() < 1e-2 and np.abs(vel[cross]).max() < 0.1   # the crossed pair balances on one contact point
    total = w.total_contact_impulse()                                            # every body's weight reaches the ground (+ stacked ones twice)
    m_big, m_small, m_ball = cyl + sph, np.pi * 0.04 * 1.0 + 4.0 / 3.0 * np.pi * 0.008, 4.0 / 3.0 * np.pi * 0.027
    expected = (m_big * 2 + m_big * 2 + (m_ball + m_small) + m_small) * 9.81 / 60.0
    assert total == pytest.approx(expected, rel=0.03)


# Events (pipeline/event_handler.rs:94-160): Started / Stopped on touching transitions, contact force events above the
# threshold with `started` on the first step above it (geometry/mod.rs:223-258).
def test_collision_and_contact_force_events():
    sc = S.box_stack(2, gap=0.5).enable_events(S.ACTIVE_EVENTS_COLLISION | S.ACTIVE_EVENTS_CONTACT_FORCE, 15.0)
    w = OracleWorld(sc)
    w.step(80)
    ev = w.collision_events()
    assert [tuple(e[:4]) for e in ev] == [(0, 1, 1, 0), (1, 2, 1, 0)]          # Started(ground, box1), Started(box1, box2)
    assert ev[0, 4] == 1 and ev[1, 4] > 10                                      # the upper box lands later
    meta, vals = w.force_events()
    ground = meta[(meta[:, 0] == 0) & (meta[:, 1] == 1)]
    assert ground[0, 3] == 1 and (ground[-20:, 3] == 0).all()                   # `started` only on the first step above the threshold
    last = vals[(meta[:, 0] == 0) & (meta[:, 1] == 1)][-1]
    assert last[3] == pytest.approx(2 * 9.81, rel=0.02) and last[1] == pytest.approx(last[3])   # carries both boxes, along +y
    assert not ((meta[:, 0] == 1) & (meta[:, 1] == 2) & (meta[:, 2] > 60)).any()   # 9.81 N between the boxes stays below 15 N
    w.remove_body(1)
    w.step(40)
    ev = w.collision_events()
    assert (0, 1, 0, 2) in [tuple(e[:4]) for e in ev] and (1, 2, 0, 2) in [tuple(e[:4]) for e in ev]   # Stopped(.., REMOVED)
    assert (0, 2, 1, 0) in [tuple(e[:4]) for e in ev]                           # the upper box lands on the ground


# Kinematic bodies (solver bodies with zero inverse mass; interpolate_kinematic_velocities, substep.rs:242-264):
# a platform carries the boxes standing on it and is never pushed back.
@pytest.mark.parametrize("position_based", [False, True])
def test_kinematic_platform_carries_boxes(position_based):
    sc = S.kinematic_platform(position_based)
    w = OracleWorld(sc)
    for k in range(120):
        if position_based:
            t = (k + 1) / 60.0
            w.set_next_kinematic_position(1, [0.6 * t, 1.0 + 0.15 * t, 0.0, 0.0, np.sin(0.1 * t), 0.0, np.cos(0.1 * t)])
        w.step(1)
    pos, vel = w.read()
    assert pos[1, 0] == pytest.approx(1.2, abs=1e-3) and pos[1, 1] == pytest.approx(1.3, abs=1e-3)   # the platform follows its program exactly
    assert vel[1, 0] == pytest.approx(0.6, abs=1e-3) and vel[1, 4] == pytest.approx(0.2, abs=1e-3)  # interpolated / prescribed velocity
    assert pos[2, 1] == pytest.approx(pos[1, 1] + 0.75, abs=0.01)     # the bottom box still stands on it
    assert abs(pos[2, 0] - pos[1, 0]) < 0.3 and vel[2, 0] == pytest.approx(0.6, abs=0.1)   # and rides along
    assert pos[5, 0] == pytest.approx(5.0, abs=1e-3)                  # the free cube on the floor is untouched


# Sleeping: RigidBodyActivation::update_energy (rigid_body_components.rs:1412-1478), whole-island sleep
# (island_manager/manager.rs:335-388), wake rules (contacts.rs:333-351, sleep.rs:31-79).  The reference's own
# sleep tests (src/pipeline/physics_pipeline/test.rs:340-372: a resting body falls asleep, a woken one is awake)
# are restated as outcomes.
def test_resting_stack_falls_asleep_as_one_island_and_wakes_as_one():
    sc = S.box_stack(3).enable_sleep()
    w = OracleWorld(sc)
    w.step(20)
    assert not w.sleeping().any()             # time_until_sleep = 0.5 s = 30 steps
    w.step(40)
    assert w.sleeping()[1:].all() and not w.sleeping()[0]   # fixed bodies never report asleep
    p0, v0 = w.read()
    assert np.all(v0 == 0.0)                  # RigidBody::sleep zeroes the velocities
    assert w.stats()["num_active_manifolds"] == 0
    w.step(50)
    p1, _ = w.read()
    np.testing.assert_array_equal(p0, p1)     # nothing moves while asleep
    w.set_vel(3, (1.0, 0.0, 0.0))             # set_linvel(.., wake_up = true) on the top box
    w.step(1)
    assert not w.sleeping().any()             # the whole island woke up
    assert w.stats()["num_active_manifolds"] == 3
    w.step(120)
    assert w.sleeping()[1:].all()


def test_islands_sleep_independently_and_impact_wakes_only_the_touched_island():
    sc = S.sleep_impact()
    w = OracleWorld(sc)
    w.step(60)
    sl = w.sleeping()
    assert sl[1:7].all() and not sl[7]        # both stacks asleep, the dropped cube still falling
    woke = None
    for k in range(200):
        w.step(1)
        sl = w.sleeping()
        if not sl[1]:
            woke = k
            break
    assert woke is not None
    assert not sl[1:4].any() and sl[4:7].all()  # the struck stack woke as one island, the other stack sleeps on
    w.step(400)
    assert w.sleeping()[1:].all()
    pos, _ = w.read()
    assert np.isfinite(pos).all() and pos[7, 1] < 4.0


def test_can_sleep_false_never_sleeps_and_blocks_its_island():
    sc = S.box_stack(3).enable_sleep()
    sc.bodies[2]["can_sleep"] = 0             # the middle box cannot sleep => the island never does
    w = OracleWorld(sc)
    w.step(200)
    assert not w.sleeping().any()


def test_removing_a_collider_wakes_its_contact_partners():
    sc = S.box_stack(3).enable_sleep()
    w = OracleWorld(sc)
    w.step(60)
    assert w.sleeping()[1:].all()
    w.remove_body(1)                          # the bottom box disappears
    w.step(1)
    assert not w.sleeping()[2:].any()
    w.step(30)
    pos, _ = w.read()
    assert pos[2, 1] < 1.4                    # the boxes above came down


def test_golden_fixtures_match_oracle():
    """tests/golden/*.npz were produced by tests/golden/make_golden.py from this oracle; they pin it
    (and, in the GPU tests, the HIP path) against silent drift."""
    import glob
    import os
    from golden.make_golden import CASES
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
    assert files, "golden fixtures missing"
    for f in files:
        d = np.load(f)
        name = os.path.basename(f)[:-4]
        scene, steps = CASES[name]()
        w = OracleWorld(scene)
        w.step(steps)
        pos, vel = w.read()
        np.testing.assert_array_equal(pos, d["pos"])
        np.testing.assert_array_equal(vel, d["vel"])
