"""More of the reference's own outcome-level tests, restated against the CPU oracle (CPU only).

Each test names the reference test it restates (/root/reference/crates/rapier3d/tests/*.rs) and keeps its scene, step
counts and acceptance thresholds.  Together with tests/test_oracle_kat.py these pin the oracle — and, through the bit-exact
GPU parity tests, the HIP path — to the behaviour the reference's maintainers assert.
"""
import numpy as np
import pytest

from rapier_amd import scenes as S
from oracle_ffi import OracleWorld


def quat_from_scaled_axis(v):
    v = np.asarray(v, np.float64)
    ang = np.linalg.norm(v)
    if ang == 0.0:
        return (0.0, 0.0, 0.0, 1.0)
    ax = v / ang
    s = np.sin(ang / 2)
    return (float(ax[0] * s), float(ax[1] * s), float(ax[2] * s), float(np.cos(ang / 2)))


def rot_matrix(q):
    x, y, z, w = [float(c) for c in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def world(gravity=(0.0, -9.81, 0.0), dt=None):
    sc = S.Scene(name="kat", gravity=gravity)
    if dt is not None:
        sc.params["dt"] = dt
    return sc


def ground(sc, he=(100.0, 0.5, 100.0), y=-0.5):
    g = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, y, 0.0))
    sc.add_collider(g, half_extents=he)
    return g


def stack(sc, x, num, can_sleep=1):
    out = []
    for i in range(num):
        b = sc.add_body(translation=(x, 0.5 + i * 1.0, 0.0), can_sleep=can_sleep)
        sc.add_collider(b, half_extents=(0.5, 0.5, 0.5))
        out.append(b)
    return out


# ---- whole_island_sleep.rs ------------------------------------------------------------------------------------
def test_whole_island_blocks_partial_sleep():
    """whole_island_sleep.rs:39-77 (and :79-103): no body of an island holding a can_sleep(false) body may sleep; once that
    body is removed the whole stack sleeps; waking one body wakes the island as a unit."""
    sc = world()
    ground(sc)
    st = stack(sc, 0.0, 6)
    restless = sc.add_body(translation=(0.0, 6.5, 0.0), can_sleep=0)
    sc.add_collider(restless, half_extents=(0.5, 0.5, 0.5))
    w = OracleWorld(sc)
    w.step(240)
    assert w.sleeping().sum() == 0
    w.remove_body(restless)
    w.step(240)
    assert w.sleeping()[st].all()
    w.wake_up(st[0], True)
    w.step(1)                                   # a wake-up request takes effect at the next step here
    assert w.sleeping().sum() == 0


def test_whole_island_sleeps_after_mover_departs():
    """whole_island_sleep.rs:105-140: a sliding kinematic body in contact keeps the island awake; the stack sleeps once the
    mover slid out of contact; the mover itself (non-zero velocity) never sleeps."""
    sc = world()
    ground(sc)
    st = stack(sc, 0.0, 4)
    mover = sc.add_body(body_type=S.BODY_KINEMATIC_VELOCITY, translation=(0.999, 0.5, -0.9), linvel=(0.0, 0.0, 0.8))
    sc.add_collider(mover, half_extents=(0.5, 0.5, 0.5))
    w = OracleWorld(sc)
    w.step(90)
    assert w.sleeping().sum() == 0
    w.step(240)
    assert w.sleeping()[st].all() and not w.sleeping()[mover]


# ---- gyroscopic.rs ----------------------------------------------------------------------------------------------
def _spinning_box(gyroscopic):
    sc = world(gravity=(0.0, 0.0, 0.0))
    b = sc.add_body(angvel=(6.0, 6.0, 0.0), gyroscopic=1 if gyroscopic else 0)
    sc.add_collider(b, half_extents=(1.0, 2.0, 3.0), density=1.0)
    return OracleWorld(sc), b


def test_angular_velocity_precesses_with_gyroscopic():
    """gyroscopic.rs:39-55"""
    w, b = _spinning_box(True)
    w0 = np.array([6.0, 6.0, 0.0])
    min_cos = np.inf
    for _ in range(300):
        w.step(1)
        av = w.read()[1][b, 3:].astype(np.float64)
        min_cos = min(min_cos, av @ w0 / (np.linalg.norm(av) * np.linalg.norm(w0)))
    assert min_cos < 0.9


def test_angular_velocity_fixed_without_gyroscopic():
    """gyroscopic.rs:57-82"""
    w, b = _spinning_box(False)
    w0 = np.array([6.0, 6.0, 0.0])
    min_cos, max_err = np.inf, 0.0
    for _ in range(300):
        w.step(1)
        av = w.read()[1][b, 3:].astype(np.float64)
        min_cos = min(min_cos, av @ w0 / (np.linalg.norm(av) * np.linalg.norm(w0)))
        max_err = max(max_err, abs(np.linalg.norm(av) - np.linalg.norm(w0)))
    assert min_cos > 0.9999 and max_err < 1.0e-3


def test_angular_momentum_conserved_with_tilted_principal_frame():
    """gyroscopic.rs:84-140: a cuboid collider attached with a rotation gives a tilted principal frame; the world angular
    momentum keeps its magnitude (2 %) and direction (cos > 0.999) over 600 steps."""
    sc = world(gravity=(0.0, 0.0, 0.0))
    b = sc.add_body(angvel=(3.0, 7.0, 2.0), gyroscopic=1)
    tilt = quat_from_scaled_axis(np.array([1.0, 1.0, 1.0]) / np.sqrt(3.0) * 0.7)
    sc.add_collider(b, half_extents=(1.0, 2.0, 3.0), density=1.0, rotation=tilt)
    w = OracleWorld(sc)
    mp = w.mass_props(b).astype(np.float64)
    frame = mp[7:11]
    assert 2.0 * np.arccos(min(1.0, abs(frame[3]))) > 0.1       # tilted principal frame
    inertia_local = rot_matrix(frame) @ np.diag(1.0 / mp[4:7]) @ rot_matrix(frame).T

    def momentum():
        pos, vel = w.read()
        r = rot_matrix(pos[b, 3:])
        return r @ (inertia_local @ (r.T @ vel[b, 3:].astype(np.float64)))
    l0 = momentum(); n0 = np.linalg.norm(l0)
    max_mag_err, min_cos = 0.0, np.inf
    for _ in range(600):
        w.step(1)
        l = momentum()
        max_mag_err = max(max_mag_err, abs(np.linalg.norm(l) - n0) / n0)
        min_cos = min(min_cos, l @ l0 / (np.linalg.norm(l) * n0))
    assert max_mag_err < 0.02 and min_cos > 0.999


# ---- issue_287_kinematic_wakes_jointed_dynamic.rs -------------------------------------------------------------------
def test_moving_kinematic_wakes_jointed_dynamic():
    sc = world()
    kin = sc.add_body(body_type=S.BODY_KINEMATIC_POSITION, can_sleep=1)
    dyn = sc.add_body(translation=(0.0, -2.0, 0.0), can_sleep=1)
    sc.add_collider(dyn, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0))
    sc.add_joint(kin, dyn, (0.0, 0.0, 0.0), (0.0, 2.0, 0.0), locked_axes=S.LOCK_REVOLUTE)   # RevoluteJointBuilder::new(Vector::X)
    w = OracleWorld(sc)
    steps = 0
    while not w.sleeping()[dyn]:
        w.step(1); steps += 1
        assert steps < 2000, "dynamic body never fell asleep"
    woke, x = False, 0.0
    for _ in range(200):
        x += 0.05
        w.set_next_kinematic_position(kin, [x, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
        w.step(1)
        woke = woke or not w.sleeping()[dyn]
    assert woke
    assert abs(w.read()[0][dyn, 0] - x) < 2.0


# ---- issue_309_locked_rotations_offset_com.rs --------------------------------------------------------------------------
def _offset_com_body():
    sc = world(gravity=(0.0, 0.0, 0.0))
    b = sc.add_body(locked_axes=0x38)           # lock_rotations()
    sc.add_collider(b, half_extents=(0.5, 0.5, 0.5), translation=(1.0, 0.0, 0.0))
    return OracleWorld(sc), b


def test_angvel_rotation_pivots_about_center_of_mass():
    w, b = _offset_com_body()
    w.step(1)

    def com():
        pos, _ = w.read()
        return pos[b, :3].astype(np.float64) + rot_matrix(pos[b, 3:]) @ np.array([1.0, 0.0, 0.0])
    c0 = com()
    for i in range(100):
        w.set_vel(b, (0.0, 0.0, 0.0), (0.0, 0.0, 3.0))
        w.step(1)
        assert np.linalg.norm(com() - c0) < 1.0e-3, i


def test_set_rotation_does_not_inject_motion():
    w, b = _offset_com_body()
    for i in range(100):
        a = 0.05 * i
        w.set_pose(b, [0.0, 0.0, 0.0, 0.0, 0.0, np.sin(a / 2), np.cos(a / 2)])
        w.step(1)
        pos, vel = w.read()
        assert np.linalg.norm(pos[b, :3]) < 1.0e-4 and np.linalg.norm(vel[b, :3]) < 1.0e-4, i


# ---- issue_746_prismatic_axis_frames.rs -----------------------------------------------------------------------------------
def test_prismatic_joint_stays_bounded_for_all_axis_rotations():
    for i in range(8):
        angle = np.pi / 2.0 * i
        sc = world(dt=0.016)
        b1 = sc.add_body(gravity_scale=0.0)
        b2 = sc.add_body(translation=(1.0, 0.0, 0.0), rotation=quat_from_scaled_axis((0.0, angle, 0.0)), gravity_scale=0.0)
        sc.add_collider(b1, half_extents=(1.0, 1.0, 1.0))
        sc.add_collider(b2, half_extents=(1.0, 1.0, 1.0))
        # local_axis2 = R^-1 X: the frame of body 2 is rotated back so that both frames' X axes agree in world space
        sc.add_joint(b1, b2, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), locked_axes=S.LOCK_PRISMATIC, contacts_enabled=0,
                     basis2=quat_from_scaled_axis((0.0, -angle, 0.0)))
        w = OracleWorld(sc)
        w.step(60)
        pos, _ = w.read()
        assert np.linalg.norm(pos[b1, :3]) < 5.0 and np.linalg.norm(pos[b2, :3]) < 5.0, i


# ---- issue_666_additional_mass_inertia.rs / issue_78_additional_mass_rest.rs ------------------------------------------------
def _topple_world(use_additional_mass):
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.1, 0.0))
    sc.add_collider(g, half_extents=(100.1, 0.1, 100.1), friction=0.5)
    vol = 8 * 0.2 * 5.0 * 1.5
    b = sc.add_body(translation=(-10.0, 6.0, 0.0), linvel=(20.0, 0.0, 0.0), additional_mass=0.5 if use_additional_mass else 0.0)
    sc.add_collider(b, half_extents=(0.2, 5.0, 1.5), friction=0.5, density=0.0 if use_additional_mass else 0.5 / vol)
    return OracleWorld(sc), b


def test_additional_mass_body_topples_like_density_twin():
    """issue_666: a tall plate sliding at 20 m/s topples (angle > 0.5 rad) whether its mass comes from the density or from
    additional_mass with a massless collider (the inertia is then derived from the shape at unit density)."""
    for use_add in (False, True):
        w, b = _topple_world(use_add)
        if use_add:
            w.step(1)
            assert (w.mass_props(b)[4:7] > 0).all()         # non-zero angular inertia
        max_angle = 0.0
        for _ in range(200):
            w.step(1)
            q = w.read()[0][b, 3:]
            max_angle = max(max_angle, 2.0 * np.arccos(min(1.0, abs(float(q[3])))))
        assert max_angle > 0.5, use_add


def test_additional_mass_body_rests_like_density_twin():
    """issue_78: a tilted 100 kg cube comes to rest (falls asleep) at the same height either way, in comparable time."""
    res = []
    for use_add in (False, True):
        sc = world()
        g = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
        sc.add_collider(g, half_extents=(5.0, 0.5, 5.0), friction=0.5)
        b = sc.add_body(translation=(0.0, 1.0, 0.0), rotation=quat_from_scaled_axis((0.0, 0.0, np.pi / 4 * 0.9)),
                        additional_mass=100.0 if use_add else 0.0, can_sleep=1)
        sc.add_collider(b, half_extents=(0.5, 0.5, 0.5), friction=0.5, restitution=0.0, density=0.0 if use_add else 100.0)
        w = OracleWorld(sc)
        for step in range(1000):
            w.step(1)
            assert w.read()[0][b, 1] > -0.5
            if w.sleeping()[b]:
                break
        else:
            pytest.fail("body never came to rest")
        res.append((step, float(w.read()[0][b, 1])))
    assert abs(res[0][1] - res[1][1]) < 0.1
    assert res[1][0] < max(res[0][0], 1) * 4


# ---- sleep_wide_bodies.rs ----------------------------------------------------------------------------------------------
def test_wide_bodies_at_rest_fall_asleep():
    """sleep_wide_bodies.rs:46-62: 64 wide U-shaped compound bodies resting on the ground all sleep within 300 steps (the
    pose-drift sleep metric scales the rotation chord with the body's extent)."""
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.1, 0.0))
    sc.add_collider(g, half_extents=(50.0, 0.1, 50.0))
    rad = 0.2
    handles = []
    for i in range(8):
        for k in range(8):
            b = sc.add_body(translation=(i * 6.0, rad + 0.01, k * 6.0), rotation=quat_from_scaled_axis((0.0, i * 0.11 + k * 0.037, 0.0)), can_sleep=1)
            sc.add_collider(b, half_extents=(rad * 10.0, rad, rad))
            sc.add_collider(b, half_extents=(rad, rad * 10.0, rad), translation=(rad * 10.0, rad * 10.0, 0.0))
            sc.add_collider(b, half_extents=(rad, rad * 10.0, rad), translation=(-rad * 10.0, rad * 10.0, 0.0))
            handles.append(b)
    w = OracleWorld(sc)
    w.step(300)
    assert w.sleeping()[handles].all()


def test_still_wide_body_reports_no_drift():
    """sleep_wide_bodies.rs:64-90: a force-free body does not move at all and sleeps despite its far-reaching shape."""
    sc = world(gravity=(0.0, 0.0, 0.0))
    b = sc.add_body(rotation=quat_from_scaled_axis((0.3, -0.7, 0.15)), can_sleep=1)
    sc.add_collider(b, half_extents=(0.2, 8.0, 0.2))
    w = OracleWorld(sc)
    p0 = w.read()[0].copy()
    w.step(200)
    np.testing.assert_array_equal(w.read()[0], p0)
    assert w.sleeping()[b]


# ---- issue_499_angular_limits.rs (the impulse-joint + torque drive) ---------------------------------------------------------
def _settled_angle_deg(limits_deg, direction, motor=False):
    sc = world(gravity=(0.0, 0.0, 0.0), dt=1.0 / 60.0)
    b1 = sc.add_body(body_type=S.BODY_FIXED)
    b2 = sc.add_body(translation=(1.0, 0.0, 0.0), angular_damping=3.0)
    sc.add_collider(b2, half_extents=(0.5, 0.1, 0.1))
    sc.add_joint(b1, b2, (0.0, 0.0, 0.0), (-1.0, 0.0, 0.0), locked_axes=S.LOCK_REVOLUTE, basis1=S.AXIS_Z_BASIS, basis2=S.AXIS_Z_BASIS,
                 limits={3: (np.radians(limits_deg[0]), np.radians(limits_deg[1]))},
                 motors={3: dict(target_vel=direction * 5.0, damping=20.0)} if motor else None)   # motor_velocity(dir * 5, 20)
    w = OracleWorld(sc)
    unwrapped, prev = 0.0, 0.0
    for _ in range(600):
        if not motor:
            w.add_force(b2, torque=(0.0, 0.0, direction * 0.1))  # add_torque accumulates (it is never reset in the reference test)
        w.step(1)
        q = w.read()[0][b2, 3:].astype(np.float64)
        ang = 2.0 * np.arctan2(q[2], q[3])
        delta = ang - prev
        if delta > np.pi:
            delta -= 2 * np.pi
        elif delta < -np.pi:
            delta += 2 * np.pi
        unwrapped += delta
        prev = ang
    return np.degrees(unwrapped)


@pytest.mark.parametrize("limits", [(-45.0, 45.0), (-135.0, 135.0), (0.0, 90.0), (-170.0, -10.0),        # within half a turn
                                    (0.0, 270.0), (-270.0, 0.0), (-90.0, 200.0), (-350.0, 0.0),          # past half a turn
                                    (45.0, 315.0), (-315.0, -45.0), (135.0, 225.0)])                     # straddling half a turn
@pytest.mark.parametrize("motor", [False, True])
def test_angular_limits_are_reached(limits, motor):
    """issue_499_angular_limits.rs:93-121 (Drive::Torque and Drive::Motor on the impulse joint): driving + / - settles within
    2 degrees of the upper / lower limit."""
    assert abs(_settled_angle_deg(limits, 1.0, motor) - limits[1]) < 2.0
    assert abs(_settled_angle_deg(limits, -1.0, motor) - limits[0]) < 2.0


@pytest.mark.parametrize("limits", [(-180.0, 180.0), (-200.0, 200.0), (-350.0, 350.0)])
@pytest.mark.parametrize("motor", [False, True])
def test_angular_limits_wider_than_a_turn_leave_the_joint_free(limits, motor):
    """issue_499_angular_limits.rs:123-133 (the reference drives this one with the velocity motor)."""
    assert _settled_angle_deg(limits, 1.0, motor) > 360.0


def test_a_joint_shoved_past_its_limit_comes_back():
    """issue_499_angular_limits.rs:135-170"""
    sc = world(gravity=(0.0, 0.0, 0.0), dt=1.0 / 60.0)
    b1 = sc.add_body(body_type=S.BODY_FIXED)
    b2 = sc.add_body(translation=(-0.996, -0.087, 0.0), rotation=quat_from_scaled_axis((0.0, 0.0, np.radians(185.0))))
    sc.add_collider(b2, half_extents=(0.5, 0.1, 0.1))
    sc.add_joint(b1, b2, (0.0, 0.0, 0.0), (-1.0, 0.0, 0.0), locked_axes=S.LOCK_REVOLUTE, basis1=S.AXIS_Z_BASIS, basis2=S.AXIS_Z_BASIS,
                 limits={3: (0.0, np.radians(170.0))})
    w = OracleWorld(sc)
    w.step(300)
    q = w.read()[0][b2, 3:].astype(np.float64)
    angle = np.degrees(2.0 * np.arctan2(q[2], q[3]))
    assert -2.0 <= angle < 172.0


# ---- issue_692_joint_get_mut_wakes_bodies.rs / issue_856_motor_position_rotating_base.rs -------------------------------------
def test_joint_get_mut_wakes_sleeping_bodies():
    """issue_692: changing a joint's motor through ImpulseJointSet::get_mut(handle, true) wakes both bodies and the motor acts."""
    sc = world()
    kin = sc.add_body(body_type=S.BODY_KINEMATIC_POSITION, can_sleep=1)
    dyn = sc.add_body(translation=(0.0, -2.0, 0.0), can_sleep=1)
    sc.add_collider(dyn, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0))
    j = sc.add_joint(kin, dyn, (0.0, 0.0, 0.0), (0.0, 2.0, 0.0), locked_axes=S.LOCK_REVOLUTE)
    w = OracleWorld(sc)
    steps = 0
    while not w.sleeping()[dyn]:
        w.step(1); steps += 1
        assert steps < 2000, "dynamic body never fell asleep"
    w.set_joint_motor(j, 3, target_vel=2.0, damping=100.0)      # set_motor_velocity(JointAxis::AngX, 2.0, 100.0)
    w.step(1)
    assert not w.sleeping()[dyn]
    moved = False
    for _ in range(50):
        w.step(1)
        moved = moved or np.linalg.norm(w.read()[1][dyn, 3:]) > 0.1
    assert moved


def test_motor_position_with_rotating_base_stays_finite():
    """issue_856: a stiff position motor whose base body — ColliderBuilder::cylinder(0.2, 1.0) — is re-oriented by the user every frame
    keeps every body finite."""
    sc = world()
    base = sc.add_body(translation=(0.0, 3.0, 0.0))
    sc.add_collider(base, shape=S.SHAPE_CYLINDER, half_extents=(0.2, 1.0, 0.0))
    hammer = sc.add_body(translation=(2.0, 3.0, 0.0))
    sc.add_collider(hammer, half_extents=(0.5, 0.1, 0.1))
    sc.add_joint(base, hammer, (1.0, 0.0, 0.0), (-1.0, 0.0, 0.0), locked_axes=S.LOCK_REVOLUTE, basis1=S.AXIS_Z_BASIS, basis2=S.AXIS_Z_BASIS,
                 motors={3: dict(target_pos=np.pi, stiffness=1.0e4, damping=100.0)})     # motor_position(PI, 1e4, 100)
    w = OracleWorld(sc)
    for i in range(300):
        a = i * 0.05
        t = w.read()[0][base, :3]
        w.set_pose(base, [t[0], t[1], t[2], 0.0, np.sin(a / 2), 0.0, np.cos(a / 2)])
        w.step(1)
        pos, vel = w.read()
        assert np.isfinite(pos).all() and np.isfinite(vel).all(), i


# ---- sleep_wake.rs -------------------------------------------------------------------------------------------------------
def _sw_world():
    sc = world()
    ground(sc, he=(50.0, 0.5, 50.0))
    return sc


def _cube(sc, t, **kw):
    kw.setdefault("can_sleep", 1)
    b = sc.add_body(translation=t, **kw)
    sc.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    return b


def test_woken_body_is_supported_by_recycled_contacts():
    """sleep_wake.rs:3-97: a cube woken by an impulse (colliders unmoved -> contact recycling path) is still supported."""
    sc = world()
    ground(sc, he=(10.0, 0.5, 10.0))
    cube = _cube(sc, (0.0, 0.6, 0.0))
    w = OracleWorld(sc)
    for _ in range(400):
        w.step(1)
        if w.sleeping()[cube]:
            break
    assert w.sleeping()[cube]
    w.apply_impulse(cube, impulse=(0.5, 0.0, 0.0))
    for _ in range(120):
        w.step(1)
        assert w.read()[0][cube, 1] > 0.4


def test_non_sleeping_neighbor_keeps_touching_row_awake():
    """sleep_wake.rs:171-210"""
    sc = _sw_world()
    row = [_cube(sc, (float(i), 0.5, 0.0), can_sleep=0 if i == 0 else 1) for i in range(8)]
    lone = _cube(sc, (30.0, 0.5, 0.0))
    w = OracleWorld(sc)
    w.step(400)
    slp = w.sleeping()
    assert not slp[row].any() and slp[lone]
    pos = w.read()[0]
    for i, h in enumerate(row):
        assert abs(pos[h, 1] - 0.5) < 0.1 and abs(pos[h, 0] - i) < 0.1


def test_impact_wakes_sleeping_region():
    """sleep_wake.rs:212-255: a fast cube thrown at a sleeping row wakes the cube it hits, which responds physically."""
    sc = _sw_world()
    row = [_cube(sc, (float(i), 0.5, 0.0)) for i in range(6)]
    w = OracleWorld(sc)
    w.step(400)
    assert w.sleeping()[row].all()
    bullet = w.add_body(translation=(-3.0, 0.5, 0.0), linvel=(20.0, 0.0, 0.0), can_sleep=1)
    w.add_collider(bullet, half_extents=(0.5, 0.5, 0.5))
    w.step(30)
    pos, vel = w.read()
    assert not w.sleeping()[row[0]]
    assert np.linalg.norm(vel[row[0], :3]) > 0.05 or pos[row[0], 0] > 0.05


def test_joint_keeps_both_sides_awake():
    """sleep_wake.rs:257-310"""
    sc = _sw_world()
    mover = _cube(sc, (0.0, 0.5, 0.0), can_sleep=0)
    b = _cube(sc, (1.0, 0.5, 0.0))
    a = _cube(sc, (2.5, 0.5, 0.0))
    control = _cube(sc, (10.0, 0.5, 0.0))
    sc.add_joint(b, a, (1.5, 0.0, 0.0), (0.0, 0.0, 0.0), locked_axes=S.LOCK_ALL)
    w = OracleWorld(sc)
    w.step(400)
    slp = w.sleeping()
    assert not slp[mover] and not slp[b] and not slp[a] and slp[control]
    pos = w.read()[0]
    assert abs(pos[a, 0] - 2.5) < 0.1 and abs(pos[b, 0] - 1.0) < 0.1


def test_corner_velocity_sleep_metric():
    """sleep_wake.rs:312-356: a long beam pivoting at 0.3 rad/s (tips at 3 m/s) stays awake; a 5 cm pebble spinning at
    0.55 rad/s (surface at < 0.05 m/s) sleeps."""
    sc = _sw_world()
    beam = sc.add_body(translation=(0.0, 30.0, 0.0), angvel=(0.0, 0.0, 0.3), gravity_scale=0.0, can_sleep=1)
    sc.add_collider(beam, half_extents=(10.0, 0.1, 0.1))
    pebble = sc.add_body(translation=(0.0, 30.0, 20.0), angvel=(0.0, 0.0, 0.55), gravity_scale=0.0, can_sleep=1)
    sc.add_collider(pebble, half_extents=(0.05, 0.05, 0.05))
    w = OracleWorld(sc)
    w.step(200)
    assert not w.sleeping()[beam] and w.sleeping()[pebble]


def test_sliding_support_wakes_sleeping_rider():
    """sleep_wake.rs:358-418"""
    sc = _sw_world()
    pusher = sc.add_body(body_type=S.BODY_KINEMATIC_VELOCITY, translation=(-1.55, 0.5, 0.0), linvel=(0.15, 0.0, 0.0), can_sleep=1)
    sc.add_collider(pusher, half_extents=(0.5, 0.5, 0.5))
    support = _cube(sc, (0.0, 0.5, 0.0))
    rider = _cube(sc, (0.0, 1.5, 0.0))
    w = OracleWorld(sc)
    slept = woke = False
    for _ in range(400):
        w.step(1)
        if w.sleeping()[rider]:
            slept = True
        elif slept:
            woke = True
    assert slept and woke
    pos = w.read()[0]
    assert pos[rider, 0] > 0.3
    assert abs(pos[rider, 1] - 1.5) < 0.2 and abs(pos[rider, 0] - pos[support, 0]) < 0.75


def test_slow_kinematic_wakes_sleeping_body_on_contact():
    """sleep_wake.rs:420-462"""
    sc = _sw_world()
    cube = _cube(sc, (0.0, 0.5, 0.0))
    wall = sc.add_body(body_type=S.BODY_KINEMATIC_VELOCITY, translation=(-2.5, 0.5, 0.0), linvel=(0.3, 0.0, 0.0), can_sleep=1)
    sc.add_collider(wall, half_extents=(0.5, 0.5, 0.5))
    w = OracleWorld(sc)
    w.step(120)
    assert w.sleeping()[cube]
    w.step(320)
    assert not w.sleeping()[cube] and w.read()[0][cube, 0] > 0.2


def test_huge_slow_platform_wakes_small_sleeping_body():
    """sleep_wake.rs:464-508"""
    sc = _sw_world()
    cube = _cube(sc, (0.0, 0.5, 0.0))
    plat = sc.add_body(body_type=S.BODY_KINEMATIC_VELOCITY, translation=(-11.0, 0.5, 0.0), linvel=(0.3, 0.0, 0.0), can_sleep=1)
    sc.add_collider(plat, half_extents=(10.0, 0.4, 10.0))
    w = OracleWorld(sc)
    w.step(90)
    assert w.sleeping()[cube]
    w.step(60)
    assert not w.sleeping()[cube] and w.read()[0][cube, 0] > 0.02


def _tilt(q):
    return float(np.arccos(np.clip((rot_matrix(q) @ np.array([0.0, 1.0, 0.0]))[1], -1.0, 1.0)))


def test_slow_dynamic_intruder_wakes_sleeping_body():
    """sleep_wake.rs:510-556: a slowly toppling domino chain falls completely (the wave does not stall against sleeping
    dominoes) and the fallen chain goes back to sleep."""
    sc = _sw_world()
    dom = []
    for i in range(10):
        b = sc.add_body(translation=(i * 0.4, 2.0, 0.0), rotation=quat_from_scaled_axis((0.0, 0.0, -0.2)) if i == 0 else (0, 0, 0, 1), can_sleep=1)
        sc.add_collider(b, half_extents=(0.1, 2.0, 1.0))
        dom.append(b)
    w = OracleWorld(sc)
    w.step(900)
    pos = w.read()[0]
    for i, h in enumerate(dom):
        assert _tilt(pos[h, 3:]) > 0.5, i
    w.step(600)
    assert w.sleeping()[dom].all()


def test_grazing_wedge_wakes_sleeping_chain():
    """sleep_wake.rs:558-618: dominoes 150..186 of the domino-spiral demo, the pre-tilted one leaning backward onto the segment."""
    sc = _sw_world()
    f32 = np.float32
    two_pi = f32(2.0) * f32(np.pi)
    curr_angle, curr_rad = f32(0.0), f32(10.0)
    dom = []
    for i in range(187):
        perimeter = two_pi * curr_rad
        prev_angle = curr_angle
        curr_angle = f32(curr_angle + two_pi * f32(0.4) / perimeter)
        x, z = np.sin(curr_angle), np.cos(curr_angle)
        nudged = np.fmod(curr_angle, two_pi) < np.fmod(prev_angle, two_pi)
        tilt = 0.2 if nudged else 0.0
        if i >= 150:
            rot = quat_from_scaled_axis((0.0, float(curr_angle), 0.0))
            tilt_axis = rot_matrix(rot) @ np.array([0.0, 0.0, 1.0])
            tq = quat_from_scaled_axis(tilt_axis * tilt)
            # tilt_rot * rot
            ax, ay, az, aw = tq; bx, by, bz, bw = rot
            q = (aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                 aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz)
            b = sc.add_body(translation=(float(x * curr_rad), 2.1, float(z * curr_rad)), rotation=q, can_sleep=1)
            sc.add_collider(b, half_extents=(0.1, 2.0, 1.0))
            dom.append(b)
        curr_rad = f32(curr_rad + f32(1.5) / perimeter)
    w = OracleWorld(sc)
    w.step(1800)
    pos = w.read()[0]
    for i, h in enumerate(dom):
        assert _tilt(pos[h, 3:]) > 0.5, i


def test_sub_gate_impact_wakes_and_transfers_momentum():
    """sleep_wake.rs:620-676: a 0.6 m/s frictionless impact on a sleeping cube wakes it in the contact step and hands over
    its momentum (target vx ~ 0.3, > 0.25)."""
    sc = _sw_world()
    g0 = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 70.0))
    sc.add_collider(g0, half_extents=(20.0, 0.5, 10.0), friction=0.0)
    target = sc.add_body(translation=(0.0, 0.5, 70.0), can_sleep=1)
    sc.add_collider(target, half_extents=(0.5, 0.5, 0.5), friction=0.0)
    w = OracleWorld(sc)
    w.step(120)
    assert w.sleeping()[target]
    mover = w.add_body(translation=(-3.0, 0.5, 70.0), linvel=(0.6, 0.0, 0.0), can_sleep=1)
    w.add_collider(mover, half_extents=(0.5, 0.5, 0.5), friction=0.0)
    w.step(240)
    assert not w.sleeping()[target]
    assert w.read()[1][target, 0] > 0.25


def test_floating_region_does_not_sleep_partially():
    """sleep_wake.rs:678-740"""
    sc = _sw_world()
    mover = _cube(sc, (0.0, 0.5, 0.0), can_sleep=0)
    col = [_cube(sc, (0.0, 1.5 + i, 0.0)) for i in range(3)]
    control = [_cube(sc, (10.0, 0.5 + i, 0.0)) for i in range(3)]
    w = OracleWorld(sc)
    w.step(600)
    slp = w.sleeping()
    assert not slp[mover] and not slp[col].any() and slp[control].all()
    pos = w.read()[0]
    for i, h in enumerate([mover] + col):
        assert abs(pos[h, 1] - (0.5 + i)) < 0.1 and abs(pos[h, 0]) < 0.1


def test_whole_floating_island_sleeps():
    """sleep_wake.rs:742-772"""
    sc = _sw_world()
    a = sc.add_body(translation=(0.0, 30.0, 0.0), gravity_scale=0.0, can_sleep=1)
    sc.add_collider(a, half_extents=(0.5, 0.5, 0.5))
    b = sc.add_body(translation=(0.999, 30.0, 0.0), gravity_scale=0.0, can_sleep=1)
    sc.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    w = OracleWorld(sc)
    w.step(400)
    assert w.sleeping()[a] and w.sleeping()[b]


def test_slow_drift_does_not_sleep_mid_motion():
    """sleep_wake.rs:774-810"""
    sc = _sw_world()
    drifter = sc.add_body(translation=(0.0, 30.0, 0.0), linvel=(0.3, 0.0, 0.0), gravity_scale=0.0, can_sleep=1)
    sc.add_collider(drifter, half_extents=(0.5, 0.5, 0.5))
    creeper = sc.add_body(translation=(0.0, 30.0, 20.0), linvel=(0.03, 0.0, 0.0), gravity_scale=0.0, can_sleep=1)
    sc.add_collider(creeper, half_extents=(0.5, 0.5, 0.5))
    w = OracleWorld(sc)
    w.step(400)
    assert not w.sleeping()[drifter] and w.sleeping()[creeper]


# ---- joint_stability.rs / joint_assembly_persistence.rs / issue_952_simd_joint_offset_com.rs -------------------------------
def _axis_basis_z(angle):
    """local basis whose X axis is X rotated by `angle` about Z (PrismaticJointBuilder::new(axis) completes the frame with
    the minimal rotation taking +X to the axis)"""
    return (0.0, 0.0, float(np.sin(angle / 2)), float(np.cos(angle / 2)))


def test_prismatic_limit_chains_remain_stable():
    """joint_stability.rs:38-102: ten chains of ten boxes hanging from limited prismatic joints on alternating diagonal rails
    stay bounded (|pos| < 200, |v| < 100) over 10,000 steps."""
    sc = world()
    rad, shift = 0.4, 1.0
    for chain in range(10):
        x = chain * 4.0
        parent = sc.add_body(body_type=S.BODY_FIXED, translation=(x, 0.0, 0.0))
        sc.add_collider(parent, half_extents=(rad, rad, rad))
        for i in range(10):
            child = sc.add_body(translation=(x, -(i + 1) * shift, 0.0))
            sc.add_collider(child, half_extents=(rad, rad, rad))
            basis = _axis_basis_z(np.pi / 4 if i % 2 == 0 else 3 * np.pi / 4)          # (+-1, 1, 0) / sqrt(2)
            sc.add_joint(parent, child, (0.0, 0.0, 0.0), (0.0, shift, 0.0), locked_axes=S.LOCK_PRISMATIC, basis1=basis, basis2=basis,
                         limits={0: (-1.5, 1.5)})
            parent = child
    w = OracleWorld(sc)
    for _ in range(10):
        w.step(1000)
        pos, vel = w.read()
        assert np.linalg.norm(pos[:, :3], axis=1).max() < 200.0
        assert np.linalg.norm(vel[:, :3], axis=1).max() < 100.0


def _pendulum(anchor_pos=(0.0, 0.0, 0.0)):
    sc = world()
    anchor = sc.add_body(body_type=S.BODY_FIXED, translation=anchor_pos)
    bob = sc.add_body(translation=(anchor_pos[0], anchor_pos[1] - 2.0, anchor_pos[2]), can_sleep=1)
    sc.add_collider(bob, shape=S.SHAPE_BALL, half_extents=(0.2, 0.0, 0.0))
    j = sc.add_joint(anchor, bob, (0.0, 0.0, 0.0), (0.0, 2.0, 0.0), locked_axes=S.LOCK_LIN)
    return OracleWorld(sc), anchor, bob, j


def test_removed_joint_stops_constraining():
    """joint_assembly_persistence.rs:32-54"""
    w, _, bob, j = _pendulum()
    w.step(30)
    assert w.read()[0][bob, 1] > -2.5
    w.remove_joint(j)
    w.step(60)
    assert w.read()[0][bob, 1] < -4.0


def test_moved_fixed_anchor_takes_its_joint_along():
    """joint_assembly_persistence.rs:84-107: after the fixed anchor is moved the bob is pinned 2.0 from the NEW anchor."""
    w, anchor, bob, _ = _pendulum()
    w.step(30)
    w.set_pose(anchor, [5.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
    w.step(300)
    dist = np.linalg.norm(w.read()[0][bob, :3] - np.array([5.0, 0.0, 0.0]))
    assert abs(dist - 2.0) < 0.3


@pytest.mark.parametrize("n", [8, 128])
def test_joints_respect_offset_center_of_mass(n):
    """issue_952: pendulums whose capsule_x(1.0, 0.2) collider — hence centre of mass — sits 1.0 off the body
    origin the revolute joint is anchored at; 128 of them fill a parallel joint colour (>= 64 joints).  The anchor never
    strays more than 1e-2 from its base."""
    sc = world()
    bodies = []
    for i in range(n):
        base = sc.add_body(body_type=S.BODY_FIXED, translation=(i * 20.0, 0.0, 0.0))
        b = sc.add_body(translation=(i * 20.0, 0.0, 0.0))
        sc.add_collider(b, shape=S.SHAPE_CAPSULE, half_extents=(1.0, 0.2, 0.0), translation=(1.0, 0.0, 0.0))
        sc.add_joint(base, b, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), locked_axes=S.LOCK_REVOLUTE, basis1=S.AXIS_Z_BASIS, basis2=S.AXIS_Z_BASIS, contacts_enabled=0)
        bodies.append(b)
    w = OracleWorld(sc)
    max_err = 0.0
    for _ in range(120):
        w.step(1)
        pos = w.read()[0]
        err = np.abs(pos[bodies, :3] - np.array([[i * 20.0, 0.0, 0.0] for i in range(n)])).max()
        max_err = max(max_err, float(err))
    assert max_err < 1.0e-2


# ---- contact_force_event_first_tick.rs -----------------------------------------------------------------------------------
def test_force_event_started_marks_threshold_crossings_not_contact_newness():
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED)
    sc.add_collider(g, half_extents=(10.0, 0.5, 10.0), translation=(0.0, -0.5, 0.0))
    ball = sc.add_body(translation=(0.0, 0.5, 0.0), additional_mass=1.0)
    sc.add_collider(ball, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0), density=0.0,
                    active_events=S.ACTIVE_EVENTS_COLLISION | S.ACTIVE_EVENTS_CONTACT_FORCE, contact_force_event_threshold=30.0)
    w = OracleWorld(sc)
    started_steps, force_events = [], []

    def step_range(lo, hi, press):
        for i in range(lo, hi):
            if press:
                w.add_force(ball, force=(0.0, -100.0, 0.0))
            w.step(1)
            w.add_force(ball, reset=True)
            started_steps.extend(i for e in w.collision_events() if e[2])
            meta, _ = w.force_events()
            force_events.extend((i, bool(m[3])) for m in meta)

    step_range(0, 100, False)                                   # A: settle; contact starts, no force events
    assert len(started_steps) == 1 and not force_events
    step_range(100, 160, True)                                  # B: press; the crossing comes long after Started
    assert force_events and force_events[0][1]
    assert force_events[0][0] >= 100 and force_events[0][0] > started_steps[0] + 50
    assert not any(first for _, first in force_events[1:]) and len(force_events) > 10
    step_range(160, 165, False)                                 # C: release; relaxation events are continuations
    assert not any(first for _, first in force_events[1:])
    after_press = len(force_events)
    step_range(165, 220, False)
    assert len(force_events) == after_press
    step_range(220, 280, True)                                  # D: press again: a fresh episode
    ep2 = force_events[after_press:]
    assert ep2 and ep2[0][1] and not any(first for _, first in ep2[1:])
    before = len(force_events)                                  # E: separate entirely, land hard
    w.set_vel(ball, (0.0, 8.0, 0.0))
    step_range(280, 500, False)
    ep3 = force_events[before:]
    assert ep3 and ep3[0][1]


# ---- issue_974_restitution.rs / speed_cap.rs ------------------------------------------------------------------------------
def _rebound(e):
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED)
    sc.add_collider(g, half_extents=(30.0, 0.1, 30.0), restitution=e)
    ball = sc.add_body(translation=(0.0, 2.3, 0.0))
    sc.add_collider(ball, shape=S.SHAPE_BALL, half_extents=(0.2, 0.0, 0.0), density=1.0, restitution=e)
    w = OracleWorld(sc)
    touched, apex = False, 0.0
    for _ in range(400):
        w.step(1)
        y = float(w.read()[0][ball, 1])
        touched = touched or y < 0.35
        if touched:
            apex = max(apex, y)
    return (apex - 0.3) / 2.0


@pytest.mark.parametrize("e", [0.3, 0.5, 0.8, 0.95])
def test_restitution_rebound_matches_e_squared(e):
    """issue_974_restitution.rs:71-81: a ball dropped from 2 m recovers e^2 of the drop height (+-0.05)."""
    assert abs(_rebound(e) - e * e) < 0.05


def test_zero_restitution_does_not_bounce():
    """issue_974_restitution.rs:83-90"""
    assert _rebound(0.0) < 0.02


def _free_ball(params=None, **body):
    sc = world(gravity=(0.0, 0.0, 0.0))
    for k, v in (params or {}).items():
        sc.params[k] = v
    b = sc.add_body(**body)
    sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.2, 0.0, 0.0))
    w = OracleWorld(sc)
    w.step(1)
    return w.read()[1][b]


def test_speed_caps():
    """speed_cap.rs:61-142: linear cap (400 by default), cap disabled, angular cap pi/4 per step unless allow_fast_rotation."""
    assert abs(np.linalg.norm(_free_ball(linvel=(10000.0, 0.0, 0.0))[:3]) - 400.0) < 1.0
    assert np.linalg.norm(_free_ball({"normalized_max_linear_velocity": S.F32_MAX}, linvel=(10000.0, 0.0, 0.0))[:3]) > 9000.0
    assert abs(np.linalg.norm(_free_ball(angvel=(0.0, 500.0, 0.0))[3:]) - np.pi / 4 * 60.0) < 2.0
    assert np.linalg.norm(_free_ball(angvel=(0.0, 500.0, 0.0), allow_fast_rotation=1)[3:]) > 400.0


# ---- miri_scenes.rs (the behavioural assertions of the native run) ------------------------------------------------------------
def test_miri_scenes():
    # resting_ball_with_force_events (:98-124)
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
    sc.add_collider(g, half_extents=(10.0, 0.5, 10.0), active_events=S.ACTIVE_EVENTS_CONTACT_FORCE, contact_force_event_threshold=0.0)
    ball = sc.add_body(translation=(0.0, 0.5, 0.0), can_sleep=1)
    sc.add_collider(ball, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0))
    w = OracleWorld(sc); w.step(120)
    assert abs(w.read()[0][ball, 1] - 0.5) < 0.05 and len(w.force_events()[0]) > 0
    # small_box_stack (:126-148)
    sc = world(); ground(sc, he=(10.0, 0.5, 10.0))
    tops = stack(sc, 0.0, 3)
    w = OracleWorld(sc); w.step(120)
    assert abs(w.read()[0][tops[2], 1] - 2.5) < 0.1
    # revolute_pendulum (:150-171)
    sc = world()
    anchor = sc.add_body(body_type=S.BODY_FIXED)
    bob = sc.add_body(translation=(1.0, 0.0, 0.0), can_sleep=1)
    sc.add_collider(bob, shape=S.SHAPE_BALL, half_extents=(0.1, 0.0, 0.0))
    sc.add_joint(anchor, bob, (0.0, 0.0, 0.0), (-1.0, 0.0, 0.0), locked_axes=S.LOCK_REVOLUTE, basis1=S.AXIS_Z_BASIS, basis2=S.AXIS_Z_BASIS)
    w = OracleWorld(sc); w.step(120)
    assert abs(np.linalg.norm(w.read()[0][bob, :3]) - 1.0) < 0.05
    # body_removal_midrun (:256-290)
    sc = world(); ground(sc, he=(10.0, 0.5, 10.0))
    a = _cube(sc, (0.0, 0.5, 0.0)); b = _cube(sc, (1.0, 0.5, 0.0))
    w = OracleWorld(sc); w.step(2); w.remove_body(a); w.step(60)
    assert abs(w.read()[0][b, 1] - 0.5) < 0.05
    # kinematic_platform_carries_box (:292-320)
    sc = world()
    plat = sc.add_body(body_type=S.BODY_KINEMATIC_VELOCITY, linvel=(0.0, 0.5, 0.0), can_sleep=1)
    sc.add_collider(plat, half_extents=(1.0, 0.1, 1.0))
    rider = sc.add_body(translation=(0.0, 0.3, 0.0), can_sleep=1)
    sc.add_collider(rider, half_extents=(0.2, 0.2, 0.2))
    w = OracleWorld(sc); w.step(60)
    assert w.read()[0][rider, 1] > 0.6


# ---- solver_graph_stale_refs.rs ------------------------------------------------------------------------------------------
def test_body_churn_with_the_references_own_shapes():
    """solver_graph_stale_refs.rs:24-79 as written: ColliderBuilder::round_cylinder(rad, rad, rad / 10), cone(rad, rad), cuboid(rad, rad,
    rad) in turn, one body per step, the outermost dynamic ones removed once more than 120 exist (800 of the reference's 1,500 steps)"""
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -2.1, 0.0))
    sc.add_collider(g, half_extents=(40.0, 2.1, 40.0))
    w = OracleWorld(sc)
    alive, rad = [], 0.5
    for step_id in range(1, 800):
        w.step(1)
        b = w.add_body(translation=(0.0, 10.0, 0.0), can_sleep=1)
        if step_id % 3 == 0:
            w.add_collider(b, shape=S.SHAPE_ROUND_CYLINDER, half_extents=(rad, rad, 0.0), border_radius=rad / 10.0)
        elif step_id % 3 == 1:
            w.add_collider(b, shape=S.SHAPE_CONE, half_extents=(rad, rad, 0.0))
        else:
            w.add_collider(b, half_extents=(rad, rad, rad))
        alive.append(b)
        if len(alive) + 1 > 120:
            pos = w.read()[0]
            order = sorted(alive, key=lambda h: -(abs(pos[h, 0]) + abs(pos[h, 2])))
            for h in order[:len(alive) + 1 - 120]:
                w.remove_body(h); alive.remove(h)
        if step_id % 50 == 0:
            pos, vel = w.read()
            assert np.isfinite(pos[alive]).all() and np.isfinite(vel[alive]).all()
            assert pos[alive, 1].min() > -0.5
            meta, _, _ = w.manifolds()
            assert len(meta) > 0


def test_body_churn_keeps_the_world_sane():
    """solver_graph_stale_refs.rs:24-79 with cuboids / balls in place of the cylinder / cone: a body spawned every step, the
    outermost dynamic ones removed once more than 120 exist; collider removal and pair removal in the same step, sleep / wake
    transitions.  The solver-graph invariants the reference asserts internally show up here as: finite state, every manifold
    references live bodies, no body tunnels through the floor."""
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -2.1, 0.0))
    sc.add_collider(g, half_extents=(40.0, 2.1, 40.0))
    w = OracleWorld(sc)
    alive = []
    for step_id in range(1, 600):
        w.step(1)
        b = w.add_body(translation=(0.0, 10.0, 0.0), can_sleep=1)
        if step_id % 3 == 0:
            w.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0))
        else:
            w.add_collider(b, half_extents=(0.5, 0.5, 0.5) if step_id % 3 == 2 else (0.5, 0.25, 0.5))
        alive.append(b)
        if len(alive) + 1 > 120:
            pos = w.read()[0]
            order = sorted(alive, key=lambda h: -(abs(pos[h, 0]) + abs(pos[h, 2])))
            for h in order[:len(alive) + 1 - 120]:
                w.remove_body(h); alive.remove(h)
        if step_id % 50 == 0:
            pos, vel = w.read()
            assert np.isfinite(pos[alive]).all() and np.isfinite(vel[alive]).all()
            assert pos[alive, 1].min() > -0.5
            meta, _, _ = w.manifolds()
            assert len(meta) > 0


# ---- parallel_path_parity.rs / simd_backend_determinism.rs (scene definitions; their bitwise goldens need the Rust build) ------
def test_reference_stress_scenes_settle_sanely():
    """The piles settle into resting cubes (two / three layers at their rest heights), fall asleep, and the kicked clusters dropped
    onto the sleeping pile wake it and come to rest on top."""
    w = OracleWorld(S.reference_pile(12, 3, 12, chain=True))
    w.step(200)
    pos, vel = w.read()
    cubes = pos[1:433]
    assert np.abs(vel[1:433]).max() < 0.5 and cubes[:, 1].min() > 0.45 and cubes[:, 1].max() < 3.0
    assert abs(np.linalg.norm(pos[434, :3] - np.array([0.3, 8.0, 0.0])) - 0.3) < 0.02    # the first link stays pinned to the anchor's joint point
    sc = S.reference_pile(14, 2, 14, chain=False)
    w = OracleWorld(sc)
    w.step(220)
    assert w.sleeping()[1:].mean() > 0.9
    for rnd in range(10):
        for body, col in S.reference_cluster(rnd):
            b = w.add_body(**{k: (tuple(body[k]) if body[k].shape else body[k].item()) for k in ("translation", "linvel", "can_sleep")})
            w.add_collider(b, half_extents=(0.5, 0.5, 0.5))
        w.step(40)
        pos, vel = w.read()
        assert np.isfinite(pos).all() and pos[1:, 1].min() > 0.4
    w.step(300)
    pos, vel = w.read()
    assert np.abs(vel[1:, :3]).max() < 0.5 and pos[1:, 1].max() < 8.0               # the drops came to rest on the pile


# ---- issue_970_multi_collider_body_perf.rs / issue_730_many_separate_colliders_perf.rs (their behavioural halves) -------------
def multi_collider_slab(nx=50, nz=40):
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED)
    sc.add_collider(g, half_extents=(100.0, 0.5, 100.0))
    body = sc.add_body(translation=(0.0, 1.05, 0.0), can_sleep=1)
    for i in range(nx):
        for j in range(nz):
            sc.add_collider(body, half_extents=(0.5, 0.5, 0.5), translation=(i - nx / 2.0, 0.0, j - nz / 2.0))
    return sc, body


def test_multi_collider_body_rests_and_sleeps():
    """issue_970: ONE dynamic body carrying a 50 x 40 slab of 2,000 unit boxes (same-parent colliders never pair; every collider
    pairs with the ground, far more manifolds between the same two bodies than there are colours) comes to rest and falls asleep."""
    sc, body = multi_collider_slab()
    w = OracleWorld(sc)
    w.step(400)
    assert w.sleeping()[body]
    assert w.read()[0][body, 1] == pytest.approx(1.0, abs=0.02)
    assert w.stats()["num_pairs"] == 2000                       # one pair per collider, none between siblings


def separate_colliders_scene():
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED)
    n = 0
    for i in range(20):
        for j in range(20):
            pos = (i - 10.0, 0.0, j - 10.0)
            if n % 3 == 2:
                sc.add_collider(g, shape=S.SHAPE_CAPSULE, half_extents=(0.3, 0.4, 1.0), translation=pos)
            elif n % 3 == 1:
                sc.add_collider(g, shape=S.SHAPE_CYLINDER, half_extents=(0.5, 0.4, 0.0), translation=pos)      # ColliderBuilder::cylinder(0.5, 0.4)
            else:
                sc.add_collider(g, half_extents=(0.5, 0.5, 0.5), translation=pos)
            n += 1
    balls = []
    for k in range(500):
        b = sc.add_body(translation=((k % 10) - 5.0, 3.0 + (k // 100) * 1.5, ((k // 10) % 10) - 5.0), can_sleep=1)
        sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.4, 0.0, 0.0))
        balls.append(b)
    return sc, balls


def test_balls_rain_on_many_separate_fixed_colliders():
    """issue_730: 400 primitives attached to one fixed body, 500 balls raining on them: nothing tunnels, the fixed siblings never
    pair with each other."""
    sc, balls = separate_colliders_scene()
    w = OracleWorld(sc)
    w.step(100)
    pos, vel = w.read()
    assert np.isfinite(pos).all() and pos[balls, 1].min() > 0.0
    meta, _, _ = w.manifolds()
    parents = np.array(sc.collider_parents)
    assert (parents[meta[:, 0]] != parents[meta[:, 1]]).all()


# ---- persistent_islands.rs (island equality through the oracle's label accessor) ---------------------------------------------
def _islands_world(xs):
    sc = world()
    ground(sc)
    hs = [_cube(sc, (float(x), float(y), 0.0)) for x, y in xs]
    return OracleWorld(sc), hs


def _same(w, a, b):
    lab = w.island_labels()
    return lab[a] >= 0 and lab[a] == lab[b]


def test_islands_merge_on_touch_and_split_on_separation():
    """persistent_islands.rs:45-76, :136-157: stacked boxes share an island, distant ones do not; once the top box is teleported
    away the two are in distinct islands — in the very step the contact stops touching."""
    w, (bottom, top, lone) = _islands_world([(0.0, 0.5), (0.0, 1.5), (20.0, 0.5)])
    w.step(240)
    assert _same(w, bottom, top) and not _same(w, bottom, lone)
    w.set_pose(top, [40.0, 0.5, 0.0, 0.0, 0.0, 0.0, 1.0])
    w.step(1)
    assert not _same(w, bottom, top)
    w.step(240)
    assert not _same(w, bottom, top)


def test_islands_follow_joints():
    """persistent_islands.rs:78-109 (a spherical joint satisfied at rest instead of the rope joint): a joint between two distant
    resting boxes merges their islands; removing it separates them again."""
    w, (a, b) = _islands_world([(0.0, 0.5), (20.0, 0.5)])
    w.step(240)
    assert not _same(w, a, b)
    jd = np.zeros((), S.JOINT_DTYPE)
    jd["body1"], jd["body2"] = a, b
    jd["local_anchor1"], jd["local_anchor2"] = (10.0, 0.0, 0.0), (-10.0, 0.0, 0.0)
    jd["local_basis1"] = jd["local_basis2"] = (0, 0, 0, 1)
    jd["locked_axes"], jd["contacts_enabled"] = S.LOCK_LIN, 1
    for k in range(6):
        jd["motors"][k] = S.motor_desc()
    from oracle_ffi import lib
    j = lib().ro_add_joint(w._w, np.array([jd], S.JOINT_DTYPE).ctypes.data)
    w.step(1)
    assert _same(w, a, b)
    w.remove_joint(j)
    w.step(240)
    assert not _same(w, a, b)


def test_islands_split_when_the_bridge_goes():
    """persistent_islands.rs:111-134, :159-198: removing the middle box of a touching row splits the sides; lifting half of a
    six-box row away as a block leaves two multi-body islands."""
    w, (left, middle, right) = _islands_world([(0.0, 0.5), (1.0, 0.5), (2.0, 0.5)])
    w.step(240)
    assert _same(w, left, right)
    w.remove_body(middle)
    w.step(240)
    assert not _same(w, left, right)
    w, row = _islands_world([(float(i), 0.5) for i in range(6)])
    w.step(240)
    assert _same(w, row[0], row[5])
    for i in range(3, 6):
        w.set_pose(row[i], [30.0 + (i - 3), 0.5, 0.0, 0.0, 0.0, 0.0, 1.0])
    w.step(240)
    assert not _same(w, row[0], row[3]) and _same(w, row[0], row[2]) and _same(w, row[3], row[5])


# ---- issue_868_same_machine_determinism.rs -----------------------------------------------------------------------------------
def eight_ball_drop():
    sc = world()
    sc.add_collider(-1, half_extents=(100.0, 0.1, 100.0))
    hs = []
    for x in range(2):
        for y in range(2):
            for z in range(2):
                b = sc.add_body(translation=(x * 0.51 + 3.0, y * 0.5 + 3.5, z * 0.5 + 3.0), can_sleep=1)
                sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0), restitution=0.7)
                hs.append(b)
    return sc, hs


def test_same_machine_runs_are_bitwise_identical():
    """issue_868: the 8-ball drop (deeply overlapping bouncy balls) run twice — and on 1 and 4 threads — gives identical bits."""
    import oracle_ffi
    runs = []
    for threads in (1, 1, 4):
        oracle_ffi.set_threads(threads)
        try:
            sc, hs = eight_ball_drop()
            w = OracleWorld(sc)
            w.step(200)
            runs.append(w.read()[0][hs, :3].copy())
        finally:
            oracle_ffi.set_threads(1)
    np.testing.assert_array_equal(runs[0], runs[1])
    np.testing.assert_array_equal(runs[0], runs[2])
    assert np.isfinite(runs[0]).all()


# ---- broad_phase_pair_filter.rs ----------------------------------------------------------------------------------------------
def test_no_fixed_fixed_pairs():
    """broad_phase_pair_filter.rs:23-30: the broad phase creates no pair between colliders of two fixed bodies (the default
    ActiveCollisionTypes would drop it every frame), nor between a fixed body's collider and a parentless one."""
    sc = world()
    a = sc.add_body(body_type=S.BODY_FIXED)
    sc.add_collider(a, half_extents=(1.0, 1.0, 1.0))
    b = sc.add_body(body_type=S.BODY_FIXED, translation=(0.5, 0.0, 0.0))
    sc.add_collider(b, half_extents=(1.0, 1.0, 1.0))
    sc.add_collider(-1, half_extents=(1.0, 1.0, 1.0), translation=(0.0, 0.5, 0.0))
    w = OracleWorld(sc)
    w.step(3)
    assert w.stats()["num_pairs"] == 0
    d = w.add_body(translation=(0.2, 1.3, 0.0))                 # a dynamic body does pair with all three
    w.add_collider(d, half_extents=(0.5, 0.5, 0.5))
    w.step(1)
    assert w.stats()["num_pairs"] == 3


# ---- additional_solver_iterations.rs / substep_chain_high_mass_ratio.rs / test_staged.rs:214-330 (ORACLE ONLY: the device ABI does
# ---- not expose additional_solver_iterations yet — DESIGN.md section 9) --------------------------------------------------------
def _heavy_stack(extra):
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED)
    sc.add_collider(g, half_extents=(10.0, 0.5, 10.0), translation=(0.0, -0.5, 0.0))
    light = sc.add_body(translation=(0.0, 0.5, 0.0), can_sleep=1)
    sc.add_collider(light, half_extents=(0.5, 0.5, 0.5), density=1.0)
    heavy = sc.add_body(translation=(0.0, 1.5, 0.0), can_sleep=1)
    sc.add_collider(heavy, half_extents=(0.5, 0.5, 0.5), density=200.0)
    w = OracleWorld(sc)
    w.set_additional_solver_iterations(heavy, extra)
    return w, light, heavy


def test_heavy_stack_stays_stable_with_extra_iterations():
    """additional_solver_iterations.rs:115-141"""
    w, light, heavy = _heavy_stack(16)
    w.step(300)
    pos, vel = w.read()
    assert 0.3 < pos[light, 1] < 0.7 and 1.2 < pos[heavy, 1] < 1.8 and np.linalg.norm(vel[heavy, :3]) < 0.1


def test_heavy_chain_stays_stable_with_extra_iterations():
    """additional_solver_iterations.rs:143-164: a six-link rope with a 100x heavier end ball, extra iterations on the weight"""
    sc = world()
    prev = sc.add_body(body_type=S.BODY_FIXED)
    for i in range(6):
        link = sc.add_body(translation=(0.0, -(i + 1.0), 0.0), can_sleep=1)
        sc.add_collider(link, shape=S.SHAPE_BALL, half_extents=(0.4, 0.0, 0.0), density=100.0 if i == 5 else 1.0)
        sc.add_joint(prev, link, (0.0, -0.5, 0.0), (0.0, 0.5, 0.0), locked_axes=S.LOCK_LIN)
        prev = link
    w = OracleWorld(sc)
    w.set_additional_solver_iterations(prev, 16)
    w.step(300)
    end = w.read()[0][prev, :3]
    assert np.isfinite(end).all() and np.linalg.norm(end) < 20.0 and -7.5 < end[1] < -4.5


def test_extra_iterations_take_effect_and_are_deterministic():
    """additional_solver_iterations.rs:166-194"""
    def run(extra):
        w, light, heavy = _heavy_stack(extra)
        w.set_pose(heavy, [0.1, 3.0, 0.0, 0.0, 0.0, 0.0, 1.0])
        w.step(60)
        pos = w.read()[0]
        return pos[[light, heavy], :3].copy()
    plain, extra1, extra2 = run(0), run(8), run(8)
    np.testing.assert_array_equal(extra1, extra2)
    assert (plain != extra1).any()


def test_substep_chain_high_mass_ratio_stretch():
    """substep_chain_high_mass_ratio.rs:145-161: a 16-link chain with a 1000:1 end ball holds together at least 4x more tightly
    with 16 extra substeps on every body than without."""
    def peak_stretch(extra):
        sc = world()
        rad, num = 0.2, 17
        prev, joints = None, []
        for i in range(num):
            ball_rad = rad * 10.0 if i == num - 1 else rad
            shift1, shift2 = rad * 1.1, ball_rad + rad * 0.1
            z = 0.0 if i == 0 else (i - 1.0) * 2.0 * shift1 + shift1 + shift2
            b = sc.add_body(body_type=S.BODY_FIXED if i == 0 else S.BODY_DYNAMIC, translation=(0.0, 0.0, z))
            sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(ball_rad, 0.0, 0.0))
            if prev is not None:
                a1, a2 = ((0.0, 0.0, 0.0), (0.0, 0.0, -shift1 * 2.0)) if i == 1 else ((0.0, 0.0, shift1), (0.0, 0.0, -shift2))
                sc.add_joint(prev, b, a1, a2, locked_axes=S.LOCK_LIN)
                joints.append((prev, b, np.array(a1), np.array(a2)))
            prev = b
        w = OracleWorld(sc)
        for b in range(1, num):
            w.set_additional_solver_iterations(b, extra)
        peak = 0.0
        for _ in range(300):
            w.step(1)
            pos = w.read()[0]
            for b1, b2, a1, a2 in joints:
                p1 = pos[b1, :3] + rot_matrix(pos[b1, 3:]) @ a1
                p2 = pos[b2, :3] + rot_matrix(pos[b2, 3:]) @ a2
                peak = max(peak, float(np.linalg.norm(p1 - p2)))
        return peak
    baseline, elevated = peak_stretch(0), peak_stretch(16)
    assert elevated < baseline / 4.0, (baseline, elevated)


def test_substep_groups_partition():
    """test_staged.rs:214-330: three groups in descending cadence — {lone} = 8, {chain_a, chain_b} = 4 (the joint lifts the
    non-elevated partner), {plain} = 0; un-elevating every body leaves the single implicit group."""
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED)
    sc.add_collider(g, half_extents=(0.5, 0.5, 0.5))
    plain = _cube(sc, (0.0, 1.001, 0.0), can_sleep=0)
    chain_a = _cube(sc, (10.0, 5.0, 0.0), can_sleep=0)
    chain_b = _cube(sc, (10.0, 3.0, 0.0), can_sleep=0)
    sc.add_joint(chain_a, chain_b, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), locked_axes=S.LOCK_REVOLUTE, basis1=S.AXIS_Z_BASIS, basis2=S.AXIS_Z_BASIS)
    lone = _cube(sc, (-10.0, 5.0, 0.0), can_sleep=0)
    w = OracleWorld(sc)
    w.set_additional_solver_iterations(chain_b, 4)
    w.set_additional_solver_iterations(lone, 8)
    w.step(3)
    ex = w.solve_group_extras()
    assert ex[g] == -1 and ex[lone] == 8 and ex[chain_a] == 4 and ex[chain_b] == 4 and ex[plain] == 0
    w.set_additional_solver_iterations(chain_b, 0)
    w.set_additional_solver_iterations(lone, 0)
    w.step(1)
    assert (w.solve_group_extras()[[plain, chain_a, chain_b, lone]] == 0).all()


def test_solve_groups_do_not_disturb_each_other():
    """Solve groups are constraint-closed: a default-cadence stack next to an elevated chain evolves bit for bit like the same stack
    alone, and the elevated chain like the same chain alone (every body of it elevated)."""
    def build(with_stack, with_chain):
        sc = world()
        ground(sc)
        stack_h = stack(sc, 0.0, 4, can_sleep=0) if with_stack else []
        chain_h = []
        if with_chain:
            prev = sc.add_body(body_type=S.BODY_FIXED, translation=(30.0, 10.0, 0.0))
            for i in range(5):
                b = sc.add_body(translation=(30.0 + (i + 1.0), 10.0, 0.0))
                sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.3, 0.0, 0.0), density=50.0 if i == 4 else 1.0)
                sc.add_joint(prev, b, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0), locked_axes=S.LOCK_LIN)
                chain_h.append(b); prev = b
        w = OracleWorld(sc)
        if chain_h:
            w.set_additional_solver_iterations(chain_h[-1], 6)
        return w, stack_h, chain_h
    both, s_both, c_both = build(True, True)
    only_stack, s_only, _ = build(True, False)
    only_chain, _, c_only = build(False, True)
    for w in (both, only_stack, only_chain):
        w.step(150)
    ex = both.solve_group_extras()
    assert (ex[s_both] == 0).all() and (ex[c_both] == 6).all()
    np.testing.assert_array_equal(both.read()[0][s_both], only_stack.read()[0][s_only])
    np.testing.assert_array_equal(both.read()[1][s_both], only_stack.read()[1][s_only])
    np.testing.assert_array_equal(both.read()[0][c_both], only_chain.read()[0][c_only])


# ---- sensors (ORACLE ONLY so far): miri_scenes.rs:196-229, narrow_phase/intersections.rs ------------------------------------------
def test_sensor_overlap():
    """miri_scenes.rs sensor_overlap: a fixed ball sensor detects the overlapping dynamic ball after one step; the ball falls away
    (no floor, no contact with the sensor) and the intersection ends.  Started / Stopped carry CollisionEventFlags::SENSOR."""
    sc = world()
    sb = sc.add_body(body_type=S.BODY_FIXED)
    sensor = sc.add_collider(sb, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0), active_events=S.ACTIVE_EVENTS_COLLISION)
    ball = sc.add_body(translation=(0.0, 0.4, 0.0))
    ball_co = sc.add_collider(ball, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0))
    w = OracleWorld(sc)
    w.set_sensor(sensor)
    w.step(1)
    assert w.intersection_pair(sensor, ball_co) is True
    ev = w.collision_events()
    assert len(ev) == 1 and ev[0][2] == 1 and ev[0][3] & 1                      # Started, SENSOR
    w.step(120)
    assert w.intersection_pair(sensor, ball_co) is not True
    assert w.read()[0][ball, 1] < -5.0                                          # it fell straight through: sensors exert no force
    ev = w.collision_events()
    assert len(ev) == 1 and ev[0][2] == 0 and ev[0][3] & 1                      # Stopped, SENSOR
    assert w.stats()["num_active_manifolds"] == 0


def test_sensor_shapes_and_trigger_volume():
    """A cuboid trigger volume on the floor: a capsule, a box and a ball dropped through it raise Started when they enter and
    Stopped when they have come to rest below / left it; the sensor never deflects them (intersection tests of all shape pairs)."""
    sc = world()
    g = ground(sc)
    trig = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, 3.0, 0.0))
    tc = sc.add_collider(trig, half_extents=(4.0, 0.5, 4.0), active_events=S.ACTIVE_EVENTS_COLLISION)
    cap = sc.add_body(translation=(-2.0, 6.0, 0.0), rotation=quat_from_scaled_axis((0.0, 0.0, 0.7)))
    sc.add_collider(cap, shape=S.SHAPE_CAPSULE, half_extents=(0.5, 0.25, 1.0))
    box = sc.add_body(translation=(0.0, 6.5, 0.0), rotation=quat_from_scaled_axis((0.3, 0.2, 0.1)))
    sc.add_collider(box, half_extents=(0.3, 0.3, 0.3))
    ball = sc.add_body(translation=(2.0, 7.0, 0.0))
    sc.add_collider(ball, shape=S.SHAPE_BALL, half_extents=(0.3, 0.0, 0.0))
    free = OracleWorld(sc)                                                     # the same world without the trigger being a sensor...
    w = OracleWorld(sc)
    w.set_sensor(tc)
    started, stopped = set(), set()
    for _ in range(240):
        w.step(1)
        for c1, c2, st, fl, _ in w.collision_events():
            assert fl & 1 and tc in (c1, c2)
            (started if st else stopped).add(c1 + c2 - tc)
    assert started == stopped == {tc + 1, tc + 2, tc + 3}
    pos = w.read()[0]
    assert pos[[cap, box, ball], 1].max() < 1.0                                 # all three rest on the ground, below the trigger
    free.step(240)
    assert free.read()[0][[cap, box, ball], 1].min() > 3.0                      # ... where the solid slab catches them


# ---- joint_contact_solve_order.rs: joints are solved BEFORE contacts in every pass ----
def test_joint_contact_solve_order_heavy_cubes_rest_on_sprung_balls():
    """crates/rapier3d/tests/joint_contact_solve_order.rs:29-82 (`Spring Joints` demo): 31 light balls hang from springs, a cube 200
    times heavier is dropped on each; with joints solved after the contacts every cube tunnels through its ball.  The reference
    uses SpringJoint (coupled axes, outside this ABI); the same physics is restated with a prismatic joint along Y whose position
    motor is the spring (stiffness 1e3, damping from 0 to twice critical across the row)."""
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED)
    num, radius = 30, 0.5
    mass = 4.0 / 3.0 * np.pi * radius ** 3
    stiffness = 1.0e3
    critical = 2.0 * np.sqrt(stiffness * mass)
    along_y = (0.0, 0.0, 0.70710678, 0.70710678)       # the joint frame's X axis (the free one) turned onto world Y
    pairs = []
    for i in range(num + 1):
        x = -6.0 + 1.5 * i
        ball = sc.add_body(translation=(x, 4.5, 0.0))
        sc.add_collider(ball, shape=S.SHAPE_BALL, half_extents=(radius, 0.0, 0.0))
        damping = (i / (num / 2.0)) * critical
        sc.add_joint(g, ball, (x, 1.5, 0.0), (0.0, 0.0, 0.0), locked_axes=S.LOCK_PRISMATIC, basis1=along_y, basis2=along_y,
                     motors={0: dict(target_pos=0.0, stiffness=stiffness, damping=float(damping), model=S.MOTOR_FORCE_BASED)})
        cube = sc.add_body(translation=(x, 9.5, 0.0), can_sleep=1)
        sc.add_collider(cube, half_extents=(radius, radius, radius), density=100.0)
        pairs.append((ball, cube))
    w = OracleWorld(sc)
    w.step(300)
    pos, _ = w.read()
    assert np.isfinite(pos).all()
    for i, (ball, cube) in enumerate(pairs):
        assert pos[cube, 1] > pos[ball, 1], f"cube {i} tunnelled through its sprung ball (cube y = {pos[cube, 1]:.3f}, ball y = {pos[ball, 1]:.3f})"


def test_joint_contact_solve_order_with_the_references_spring_joints():
    """crates/rapier3d/tests/joint_contact_solve_order.rs:29-82 as written: SpringJointBuilder::new(0.0, stiffness, damping) with
    local_anchor1 three metres under each ball (GenericJoint::coupled_axes, round 5)."""
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED)
    num, radius = 30, 0.5
    mass = 4.0 / 3.0 * np.pi * radius ** 3
    stiffness = 1.0e3
    critical = 2.0 * np.sqrt(stiffness * mass)
    pairs = []
    for i in range(num + 1):
        x = -6.0 + 1.5 * i
        ball = sc.add_body(translation=(x, 4.5, 0.0))
        sc.add_collider(ball, shape=S.SHAPE_BALL, half_extents=(radius, 0.0, 0.0))
        sc.add_spring_joint(g, ball, (x, 1.5, 0.0), (0.0, 0.0, 0.0), 0.0, stiffness, float((i / (num / 2.0)) * critical))
        cube = sc.add_body(translation=(x, 9.5, 0.0), can_sleep=1)
        sc.add_collider(cube, half_extents=(radius, radius, radius), density=100.0)
        pairs.append((ball, cube))
    w = OracleWorld(sc)
    w.step(300)
    pos, _ = w.read()
    assert np.isfinite(pos).all()
    for i, (ball, cube) in enumerate(pairs):
        assert pos[cube, 1] > pos[ball, 1], f"cube {i} tunnelled through its spring-jointed ball (cube y = {pos[cube, 1]:.3f}, ball y = {pos[ball, 1]:.3f})"


# ---- the reference's tests on meshes and height fields (composite shapes, round 5) ---------------------------------------
def two_triangle_ground(sc, parent=-1, half=20.0, winding=((0, 2, 1), (0, 3, 2))):
    """the flat two-triangle mesh of issues 372 and 524"""
    v = np.array([[-half, 0, -half], [half, 0, -half], [half, 0, half], [-half, 0, half]], np.float32)
    return sc.add_collider(parent, shape=S.SHAPE_TRIMESH, half_extents=(sc.add_trimesh(v, np.array(winding, np.uint32)), 0, 0))


@pytest.mark.parametrize("tilt", [(0.1, 0.0, 0.0), (0.0, 0.0, 0.12), (0.15, 0.0, 0.1), (-0.12, 0.0, 0.08)], ids=["x", "z", "xz", "neg"])
def test_thin_slab_dropped_tilted_settles_on_a_trimesh(tilt):
    """issue_524_thin_slab_trimesh_tunnel.rs:61-128: a 2 x 0.06 x 2 slab dropped tilted from 5 m on a two-triangle mesh, default
    parameters (no explicit CCD): the automatic fast-body-against-fixed sweep must keep it above the sheet (y > -0.1 throughout)
    and it must lie still on it after 600 steps.  The `neg` tilt lands across the shared diagonal, is turned flat by its first
    corner contact while its centre keeps falling, and passes only because a piece that starts a step touching a triangle is swept
    again as its core ball (oracle/ro_ccd.h: ccd_core_of)."""
    sc = world()
    fixed = sc.add_body(body_type=S.BODY_FIXED)
    two_triangle_ground(sc, fixed)
    slab = sc.add_body(translation=(0.0, 5.0, 0.0), rotation=quat_from_scaled_axis(tilt), can_sleep=1)
    sc.add_collider(slab, half_extents=(1.0, 0.03, 1.0))
    w = OracleWorld(sc)
    for i in range(600):
        w.step(1)
        y = float(w.read()[0][slab, 1])
        assert y > -0.1, f"slab tunnelled through the mesh at step {i} (tilt {tilt}, y = {y})"
    pos, vel = w.read()
    assert 0.0 < pos[slab, 1] < 0.2, f"slab did not settle on the mesh (y = {pos[slab, 1]})"
    assert np.linalg.norm(vel[slab, :3]) < 0.05


@pytest.mark.parametrize("what", ["ball", "capsule"])
def test_bodies_on_a_trimesh_fall_asleep(what):
    """issue_372_trimesh_sleep.rs:81-111: a ball (linvel 0.03) / a lying capsule (linvel 0.012) dropped on a parentless two-triangle
    mesh must fall asleep within 2000 steps"""
    sc = world()
    two_triangle_ground(sc)
    if what == "ball":
        b = sc.add_body(translation=(-5.0, 1.0, 5.0), linvel=(0.03, 0.0, 0.0), can_sleep=1)
        sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0))
    else:
        b = sc.add_body(translation=(-5.0, 1.0, 5.0), rotation=quat_from_scaled_axis((np.pi / 2, 0.0, 0.0)), linvel=(0.012, 0.0, 0.0), can_sleep=1)
        sc.add_collider(b, shape=S.SHAPE_CAPSULE, half_extents=(0.4, 0.3, 1.0))
    w = OracleWorld(sc)
    for _ in range(2000):
        w.step(1)
        if w.sleeping()[b]:
            break
    else:
        pytest.fail(f"{what} never fell asleep on the mesh (velocity {w.read()[1][b]})")


def test_capsule_crosses_flush_heightfield_seam_without_sinking():
    """issue_182_heightfield_seam.rs:55-119: two flat 16 x 16 height fields side by side; a frictionless upright capsule with locked
    rotations settles at y = 0.6, is driven at 2 m/s across the seam at x = 8 and must neither sink (y > rest - 0.05 at every step)
    nor stop (x > 12 after 480 steps)."""
    sc = world()
    for x in (0.0, 16.0):
        hf = sc.add_heightfield(np.zeros((17, 17), np.float32), (16.0, 1.0, 16.0))
        sc.add_collider(-1, shape=S.SHAPE_TRIMESH, half_extents=(hf, 0, 0), translation=(x, 0.0, 0.0))
    ch = sc.add_body(translation=(0.0, 0.7, 0.0), locked_axes=8 | 16 | 32, can_sleep=1)
    sc.add_collider(ch, shape=S.SHAPE_CAPSULE, half_extents=(0.3, 0.3, 1.0), friction=0.0)
    w = OracleWorld(sc)
    w.step(120)
    rest_y = float(w.read()[0][ch, 1])
    assert abs(rest_y - 0.6) < 0.1
    for i in range(480):
        vel = w.read()[1]
        w.set_vel(ch, (2.0, float(vel[ch, 1]), 0.0), tuple(float(c) for c in vel[ch, 3:]))
        w.step(1)
        pos = w.read()[0]
        assert pos[ch, 1] > rest_y - 0.05, f"capsule sank at the seam at step {i}: x = {pos[ch, 0]}, y = {pos[ch, 1]}"
    assert pos[ch, 0] > 12.0 and abs(pos[ch, 1] - rest_y) < 0.05


def test_heightfield_stress_stays_consistent():
    """heightfield_solver_graph.rs:5-62: 2048 cubes and balls poured on a 50 x 50 rolling height field with raised borders, 200
    steps; the reference asserts that the solver's graph bookkeeping survives (no panic) — here: finite, and nothing under the
    terrain's lowest point."""
    sc = world()
    n = 50
    i, j = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    x = i.astype(np.float32) * np.float32(50.0) / np.float32(n)
    z = j.astype(np.float32) * np.float32(50.0) / np.float32(n)
    h = ((np.cos(x) + np.sin(z)) * np.float32(1.5)).astype(np.float32)
    h[0, :] = h[n, :] = 8.0
    h[:, 0] = h[:, n] = 8.0
    g = sc.add_body(body_type=S.BODY_FIXED)
    sc.add_collider(g, shape=S.SHAPE_TRIMESH, half_extents=(sc.add_heightfield(h, (50.0, 1.0, 50.0)), 0, 0))
    num, rad = 8, 0.5
    shift = rad * 2.5
    cx, cy = shift * (num // 2), shift / 2.0
    for a in range(num):
        for b in range(num * 4):
            for c in range(num):
                bd = sc.add_body(translation=(a * shift - cx, b * shift + cy + 3.0, c * shift - cx), can_sleep=1)
                if (a + b + c) % 2 == 0:
                    sc.add_collider(bd, half_extents=(rad, rad, rad))
                else:
                    sc.add_collider(bd, shape=S.SHAPE_BALL, half_extents=(rad, 0.0, 0.0))
    w = OracleWorld(sc)
    for _ in range(4):
        w.step(50)
        pos, vel = w.read()
        assert np.isfinite(pos).all() and np.isfinite(vel).all()
    assert pos[1:, 1].min() > -3.0


def test_contact_force_event_is_populated_under_contact_clustering():
    """contact_force_event_clustering.rs:44-127: a cube resting on a two-triangle mesh (several manifolds, one solver cluster) with
    CONTACT_FORCE_EVENTS and threshold 0: the last event of 60 steps carries a positive total and max force, a unit direction, and
    max <= total."""
    sc = world()
    fixed = sc.add_body(body_type=S.BODY_FIXED)
    gc = two_triangle_ground(sc, fixed, half=2.0, winding=((0, 1, 2), (0, 2, 3)))
    box = sc.add_body(translation=(0.0, 0.55, 0.0))
    bc = sc.add_collider(box, half_extents=(0.5, 0.5, 0.5), active_events=S.ACTIVE_EVENTS_CONTACT_FORCE, contact_force_event_threshold=0.0)
    w = OracleWorld(sc)
    last = None
    for _ in range(60):
        w.step(1)
        meta, vals = w.force_events()
        for m, v in zip(meta, vals):
            assert {int(m[0]), int(m[1])} == {gc, bc}
            last = v
    nclusters, per_cluster = w.pair_clusters(gc, bc)
    assert nclusters >= 1 and 1 <= per_cluster[0] <= 4          # contact clustering applied to the pair
    assert last is not None, "no contact force event was emitted"
    total, total_mag, direction, max_mag = last[:3], last[3], last[4:7], last[7]
    assert total_mag > 0.0 and max_mag > 0.0 and np.linalg.norm(total) > 0.0
    assert abs(np.linalg.norm(direction) - 1.0) < 1.0e-4
    assert max_mag <= total_mag * 1.001
