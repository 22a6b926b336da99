"""Randomised differential test (GPU, through the C ABI, vs the oracle): seeded scenes mixing every implemented feature — the three
shapes, compound bodies, kinematic bodies, locked axes, sleeping, restitution / friction rules, collision groups, all joint kinds
with limits and motors, events, both friction models, joint warm start — driven by random user actions (impulses, velocity and
pose writes, wake-ups, body / collider / joint removal, body and joint insertion, motor changes).  State, sleeping flags and drained
events must match bit for bit all along."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld, lib

import os

TRACE = os.environ.get("RP_FUZZ_TRACE") is not None
SEEDS = list(range(int(os.environ.get("RP_FUZZ_FIRST", "0")), int(os.environ.get("RP_FUZZ_LAST", "16"))))


def _rand_quat(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return tuple(float(x) for x in q)


_CONVEX = [False]   # set by _scene(convex=True): cylinders and cones join the draw (the other variants keep their random streams)


def _rand_collider(rng, scale=1.0):
    kind = rng.integers(0, (5 + (1 if _CONVEX[0] > 1 else 0) + (3 if _CONVEX[0] > 5 else 0)) if _CONVEX[0] else 3)   # (> 5: the round variants join)
    kw = dict(density=float(rng.uniform(0.5, 3.0)), friction=float(rng.choice([0.0, 0.3, 0.5, 1.0])),
              restitution=float(rng.choice([0.0, 0.0, 0.3, 0.8])), friction_rule=int(rng.integers(0, 6)), restitution_rule=int(rng.integers(0, 6)))
    if kind == 0:
        return dict(shape=S.SHAPE_BALL, half_extents=(float(rng.uniform(0.2, 0.5)) * scale, 0.0, 0.0), **kw)
    if kind == 1:
        return dict(shape=S.SHAPE_CUBOID, half_extents=tuple(float(x) * scale for x in rng.uniform(0.15, 0.6, size=3)), **kw)
    if kind == 3:
        return dict(shape=S.SHAPE_CYLINDER, half_extents=(float(rng.uniform(0.15, 0.5)) * scale, float(rng.uniform(0.15, 0.5)) * scale, 0.0), **kw)
    if kind >= 6:       # RoundShape<S>: round cuboid / cylinder / cone
        br = float(rng.uniform(0.03, 0.1)) * scale
        if kind == 6:
            return dict(shape=S.SHAPE_ROUND_CUBOID, half_extents=tuple(float(x) * scale for x in rng.uniform(0.12, 0.45, size=3)), border_radius=br, **kw)
        return dict(shape=S.SHAPE_ROUND_CYLINDER if kind == 7 else S.SHAPE_ROUND_CONE, half_extents=(float(rng.uniform(0.15, 0.4)) * scale, float(rng.uniform(0.15, 0.4)) * scale, 0.0), border_radius=br, **kw)
    if kind == 5:       # one of the polyhedra _scene registered (scaled shapes are not: a polyhedron has the size it was registered with)
        return dict(shape=S.SHAPE_CONVEX, half_extents=(float(rng.integers(0, _CONVEX[0] - 1)), 0.0, 0.0), **kw)
    if kind == 4:
        return dict(shape=S.SHAPE_CONE, half_extents=(float(rng.uniform(0.2, 0.5)) * scale, float(rng.uniform(0.2, 0.45)) * scale, 0.0), **kw)
    return dict(shape=S.SHAPE_CAPSULE, half_extents=(float(rng.uniform(0.2, 0.6)) * scale, float(rng.uniform(0.15, 0.35)) * scale, float(rng.integers(0, 3))), **kw)


def _rand_joint(rng, sc, b1, b2, p1, p2):
    """a joint whose two anchors coincide in world space at the midpoint of the two bodies (unrotated bodies only)"""
    mid = (np.asarray(p1) + np.asarray(p2)) / 2
    a1, a2 = tuple(float(x) for x in mid - p1), tuple(float(x) for x in mid - p2)
    kind = rng.integers(0, 5)
    limits, motors = None, None
    if kind == 0:
        locked = S.LOCK_LIN
        if rng.random() < 0.5:
            motors = {3 + int(rng.integers(0, 3)): dict(target_pos=float(rng.uniform(-0.5, 0.5)), stiffness=40.0, damping=4.0)}
    elif kind == 1:
        locked = S.LOCK_REVOLUTE
        if rng.random() < 0.6:
            limits = {3: (-float(rng.uniform(0.2, 1.5)), float(rng.uniform(0.2, 1.5)))}
        if rng.random() < 0.6:
            motors = {3: dict(target_vel=float(rng.uniform(-3, 3)), damping=float(rng.uniform(1, 20)), max_force=float(rng.choice([S.F32_MAX, 5.0, 50.0])))}
    elif kind == 2:
        locked = S.LOCK_PRISMATIC
        limits = {0: (-float(rng.uniform(0.1, 1.0)), float(rng.uniform(0.1, 1.0)))}
        if rng.random() < 0.5:
            motors = {0: dict(target_pos=float(rng.uniform(-0.5, 0.5)), stiffness=200.0, damping=20.0, max_force=float(rng.choice([S.F32_MAX, 30.0])),
                              model=int(rng.integers(0, 2)))}
    elif kind == 3:
        locked = S.LOCK_ALL
    else:
        locked = int(rng.integers(1, 64))                        # any mask of locked axes
        free = [a for a in range(6) if not locked & (1 << a)]
        if free and rng.random() < 0.5:
            a = int(rng.choice(free))
            limits = {a: (-0.4, 0.6)}
    basis = _rand_quat(rng) if rng.random() < 0.5 else (0.0, 0.0, 0.0, 1.0)
    return sc.add_joint(b1, b2, a1, a2, locked_axes=locked, contacts_enabled=int(rng.random() < 0.7), basis1=basis, basis2=basis, limits=limits, motors=motors)


def _scene(seed, n=40, spread=3.5, per_layer=4, calm=False, convex=False):
    _CONVEX[0] = int(convex)       # 0: the three basic shapes, 1: + cylinders and cones, n > 1: + n - 1 registered convex polyhedra
    rng = np.random.default_rng(seed)
    sc = S.Scene(name=f"fuzz{seed}", gravity=(0.0, -9.81, 0.0))
    for k in range(max(0, int(convex) - 1)):
        sc.add_convex_polyhedron((np.random.default_rng(900 + k).standard_normal((10 + 6 * k, 3)) * (0.2 + 0.05 * k)).astype(np.float32) + np.float32([0.05 * k, 0.0, 0.0]))
    sc.params["friction_model"] = S.FRICTION_COULOMB if seed % 4 == 3 else S.FRICTION_SIMPLIFIED
    sc.params["warmstart_joints"] = int(seed % 3 == 1)
    g = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
    sc.add_collider(g, half_extents=(12.0, 0.5, 12.0), friction=0.6, active_events=3 if seed % 2 else 0, contact_force_event_threshold=40.0)
    for k in range(3):                                            # fixed obstacles
        f = sc.add_body(body_type=S.BODY_FIXED, translation=(float(rng.uniform(-4, 4)), 0.4, float(rng.uniform(-4, 4))), rotation=_rand_quat(rng))
        sc.add_collider(f, **_rand_collider(rng, 1.5))
    plat = sc.add_body(body_type=S.BODY_KINEMATIC_VELOCITY, translation=(-3.0, 0.2, 3.0), linvel=(0.3, 0.0, -0.2), angvel=(0.0, 0.4, 0.0), can_sleep=1)
    sc.add_collider(plat, half_extents=(1.5, 0.2, 1.5), friction=1.0)
    kpos = sc.add_body(body_type=S.BODY_KINEMATIC_POSITION, translation=(3.0, 0.3, -3.0), can_sleep=1)
    sc.add_collider(kpos, half_extents=(1.2, 0.3, 1.2), friction=0.8)
    pos = []
    for i in range(n):
        p = (float(rng.uniform(-spread, spread)), float(1.0 + 0.9 * (i // per_layer) + rng.uniform(0, 0.3)), float(rng.uniform(-spread, spread)))
        if calm:                                                   # a jittered, non-overlapping grid: the bodies rain into a pile
            side = int(np.ceil(np.sqrt(per_layer)))
            k = i % per_layer
            cell = 2.0 * spread / side
            p = (float(-spread + (k % side + 0.5) * cell + rng.uniform(-0.05, 0.05)), float(0.8 + 0.75 * (i // per_layer)), float(-spread + (k // side + 0.5) * cell + rng.uniform(-0.05, 0.05)))
        unrot = i % 3 == 0
        b = sc.add_body(translation=p, rotation=(0.0, 0.0, 0.0, 1.0) if unrot else _rand_quat(rng),
                        linvel=tuple(float(x) for x in rng.uniform(-1, 1, size=3)), angvel=tuple(float(x) for x in rng.uniform(-2, 2, size=3)),
                        can_sleep=int(rng.random() < 0.8), linear_damping=float(rng.choice([0.0, 0.0, 0.2])), angular_damping=float(rng.choice([0.0, 0.1, 1.0])),
                        gravity_scale=float(rng.choice([1.0, 1.0, 0.5])), additional_mass=float(rng.choice([0.0, 0.0, 2.0])),
                        dominance=int(rng.choice([0, 0, 0, 1])), gyroscopic=int(rng.random() < 0.8),
                        locked_axes=int(rng.choice([0, 0, 0, 0x38, 0x02, 0x15])))
        c = _rand_collider(rng, 0.55 if calm else 1.0)
        sc.add_collider(b, active_events=int(rng.choice([0, 0, 1, 3])), contact_force_event_threshold=float(rng.choice([0.0, 20.0])),
                        memberships=int(rng.choice([0xFFFFFFFF, 0xFFFFFFFF, 0x1])), filter=int(rng.choice([0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFE])), **c)
        if rng.random() < 0.25 and not calm:                      # compound body
            sc.add_collider(b, translation=tuple(float(x) for x in rng.uniform(-0.5, 0.5, size=3)), rotation=_rand_quat(rng), **_rand_collider(rng, 0.7))
        pos.append((b, p, unrot))
    unrot = [(b, p) for b, p, u in pos if u]
    for k in range(0, (0 if calm else len(unrot) - 1), 2):                         # joints between unrotated neighbours
        (b1, p1), (b2, p2) = unrot[k], unrot[k + 1]
        _rand_joint(rng, sc, b1, b2, np.array(p1), np.array(p2))
    (b1, p1) = unrot[-1]
    sc.add_joint(g, b1, tuple(float(x) for x in (np.array(p1) + (0, 0.5, 0) - (0.0, -0.5, 0.0))), (0.0, 0.5, 0.0), locked_axes=S.LOCK_LIN)   # a pendulum on the ground body
    return sc, rng


def _check(g, o, alive, msg):
    gp, gv = g.read_bodies(); op, ov = o.read()
    np.testing.assert_array_equal(gp[alive], op[alive], err_msg=msg + " poses")
    np.testing.assert_array_equal(gv[alive], ov[alive], err_msg=msg + " velocities")
    np.testing.assert_array_equal(g.sleeping()[alive], o.sleeping()[alive], err_msg=msg + " sleeping")
    gc = g.collision_events(); oc = o.collision_events()
    key = lambda e: (int(e[4]), int(e[0]), int(e[1]), int(e[2]))
    assert sorted(map(key, gc)) == sorted(map(key, oc)), msg + " collision events"
    gm, gvv = g.contact_force_events(); om, ovv = o.force_events()
    assert len(gm) == len(om), msg + " force event count"
    if len(gm):
        gi = np.lexsort((gm[:, 1], gm[:, 0], gm[:, 2])); oi = np.lexsort((om[:, 1], om[:, 0], om[:, 2]))
        np.testing.assert_array_equal(gm[gi], om[oi], err_msg=msg + " force event meta")
        np.testing.assert_array_equal(gvv[gi], ovv[oi], err_msg=msg + " force event values")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1000, 1001, 1002])
def test_fuzz_pile_bit_exact(seed):
    """the same generator at 600 bodies dropped into a walled pit: one giant island on the global multi-kernel path (> 1024
    manifolds, parallel colour stages + serial tail), islands around it on the LDS path"""
    _run(seed, steps=260, n=700, spread=2.2, per_layer=49, walls=True, calm=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_bit_exact(seed):
    _run(seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3000, 3001, 3002, 3003])
def test_fuzz_growth_bit_exact(seed, monkeypatch):
    """RP_SPARE_ROWS=1: every inserted body / collider makes the device world outgrow its arrays, so each insertion goes through
    the state carry-over (and every joint insertion does anyway)"""
    monkeypatch.setenv("RP_SPARE_ROWS", "1")
    _run(seed)


def _random_params(sc, rng):
    """IntegrationParameters away from their defaults: substep count, inner PGS / stabilisation sweeps, warm-start coefficient,
    contact recycling, friction in the bias pass, time step, softness, correction limits, length unit"""
    p = sc.params
    p["num_solver_iterations"] = int(rng.choice([1, 2, 3, 4, 6]))
    p["num_internal_pgs_iterations"] = int(rng.choice([1, 1, 2, 3]))
    p["num_internal_stabilization_iterations"] = int(rng.choice([0, 1, 1, 2]))
    p["warmstart_coefficient"] = float(rng.choice([1.0, 1.0, 0.5, 0.0]))
    p["contact_recycling"] = int(rng.random() < 0.7)
    p["friction_in_bias_pass"] = int(rng.random() < 0.3)
    p["dt"] = float(rng.choice([1.0 / 60.0, 1.0 / 120.0, 0.016, 1.0 / 30.0]))
    p["contact_natural_frequency"] = float(rng.choice([30.0, 20.0, 60.0]))
    p["static_contact_natural_frequency"] = float(rng.choice([60.0, 30.0, 90.0]))
    p["contact_damping_ratio"] = float(rng.choice([10.0, 5.0, 1.0]))
    p["joint_natural_frequency"] = float(rng.choice([1.0e6, 50.0]))
    p["joint_damping_ratio"] = float(rng.choice([1.0, 0.5]))
    p["normalized_max_corrective_velocity"] = float(rng.choice([3.0, 1.0, 10.0]))
    p["normalized_prediction_distance"] = float(rng.choice([0.02, 0.05, 0.002]))
    p["normalized_allowed_linear_error"] = float(rng.choice([0.005, 0.001]))
    p["normalized_max_linear_velocity"] = float(rng.choice([400.0, 20.0]))
    p["normalized_contact_recycle_distance"] = float(rng.choice([0.05, 0.01]))
    p["length_unit"] = float(rng.choice([1.0, 1.0, 2.0]))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(2000, 2016)))
def test_fuzz_params_bit_exact(seed):
    """the same scenes and actions under randomised IntegrationParameters"""
    _run(seed, steps=160, params=True)


class OracleTwin:
    """A second oracle behind the PhysicsWorld method names the driver uses: the CPU twin of the differential test (the same random
    scenes and user actions, the multi-threaded oracle against the single-threaded one) keeps the driver itself and the oracle's
    determinism under user actions covered by the `-m "not gpu"` suite."""

    def __init__(self, scene):
        self.o = OracleWorld(scene)

    def step(self, n=1):
        import oracle_ffi
        oracle_ffi.set_threads(4)
        try:
            self.o.step(n)
        finally:
            oracle_ffi.set_threads(1)

    def read_bodies(self): return self.o.read()
    def sleeping(self): return self.o.sleeping()
    def collision_events(self): return self.o.collision_events()
    def contact_force_events(self): return self.o.force_events()
    def counters(self): return {"overflow_flags": 0}
    def apply_impulse(self, h, impulse=None, torque_impulse=None): self.o.apply_impulse(int(h[0]), None if impulse is None else impulse[0], None if torque_impulse is None else torque_impulse[0])
    def add_force(self, h, force=None, torque=None, reset=False): self.o.add_force(int(h[0]), force[0], torque[0], reset)
    def wake_up(self, h): self.o.wake_up(int(h[0]))
    def remove_body(self, b): self.o.remove_body(b)
    def remove_collider(self, c): self.o.remove_collider(c)
    def remove_impulse_joint(self, j): self.o.remove_joint(j)
    def set_joint_motor(self, j, axis, **kw): self.o.set_joint_motor(j, axis, **kw)
    def set_next_kinematic_position(self, h, p): self.o.set_next_kinematic_position(int(h[0]), p)
    def set_additional_solver_iterations(self, h, n): self.o.set_additional_solver_iterations(int(h[0]), int(n))

    def write_bodies(self, h, pos7=None, vel6=None):
        if pos7 is not None: self.o.set_pose(int(h[0]), pos7[0])
        if vel6 is not None: self.o.set_vel(int(h[0]), vel6[0][:3], vel6[0][3:])

    def insert_body(self, body):
        h = lib().ro_add_body(self.o._w, np.array([body], S.BODY_DTYPE).ctypes.data); self.o.n += 1
        return h

    def insert_collider(self, col, parent): return lib().ro_add_collider(self.o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, int(parent))
    def insert_impulse_joint(self, b1, b2, jd): return lib().ro_add_joint(self.o._w, np.array([jd], S.JOINT_DTYPE).ctypes.data)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 2001, 2004])
def test_fuzz_driver_on_the_oracle_twin(seed):
    """CPU: the fuzz driver with the 4-thread oracle in the device's place (same scenes, same actions, bit-exact expectations)"""
    _run(seed, steps=120, params=seed >= 2000, world=OracleTwin)


@pytest.mark.parametrize("seed", [10, 11, 12, 2002, 2009])
def test_fuzz_solve_groups_on_the_oracle_twin(seed):
    """CPU: random extra substep counts on a third of the bodies — several solve groups with their own cadence — must not depend on
    the oracle's thread count either"""
    _run(seed, steps=120, params=seed >= 2000, world=OracleTwin, extras=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [10, 11, 12, 13, 2002, 2009])
def test_fuzz_solve_groups_bit_exact(seed):
    """the same scenes and actions with random additional_solver_iterations on a third of the bodies: up to five solve groups with
    their own substep counts (rp_groups.h), bodies / joints / colliders coming and going under them"""
    _run(seed, steps=160, params=seed >= 2000, extras=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [60, 61, 62, 63, 2003, 2012])
def test_fuzz_in_a_batch_of_sub_worlds_bit_exact(seed):
    """the driver's scene and actions inside a batch of three overlapping sub-worlds (rp_world_begin_subworld): removals, insertions,
    sleeping, joints, events, CCD and both friction models with two other worlds in the same place"""
    _run(seed, params=seed >= 2000, batch=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [70, 71, 72, 73, 74, 2006, 2010])
def test_fuzz_coupled_joints_bit_exact(seed):
    """ropes, springs and coupled angular axes (GenericJoint::coupled_axes) under the driver's actions — motor changes on coupled axes,
    removals, sleeping, both friction models, joint warm start"""
    _run(seed, params=seed >= 2000, coupled=True)


@pytest.mark.parametrize("seed", [70, 2006])
def test_fuzz_coupled_joints_on_the_oracle_twin(seed):
    _run(seed, params=seed >= 2000, coupled=True, world=OracleTwin)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [20, 21, 22, 23, 2005])
def test_fuzz_sensors_bit_exact(seed):
    """the same scenes and actions with sensor colliders: a trigger volume in the middle of the pile, sensor obstacles, sensor
    parts of compound bodies (intersection graph, Started / Stopped | SENSOR events, removal of intersecting colliders)"""
    _run(seed, steps=200, params=seed >= 2000, sensors=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [30, 31, 32, 2007])
def test_fuzz_halfspace_ground_bit_exact(seed):
    """the same scenes and actions on a ground PLANE (RP_SHAPE_HALFSPACE on the fixed ground body) with a slanted plane beside the
    pile: pairs that never recycle (full narrow-phase update every step), the plane against every shape and compound body"""
    _run(seed, steps=200, params=seed >= 2000, halfspace=True)


def _couple_some_joints(sc, rng):
    """GenericJoint::coupled_axes on about two thirds of the scene's joints (their own random stream: the other variants keep theirs):
    ropes and springs (the three linear axes coupled, whatever the angular axes do) and pairs of coupled angular axes with a cone limit"""
    for j in sc.joints:
        r = rng.random()
        locked = int(j["locked_axes"])
        if r < 0.25:        # RopeJoint on top of the joint's angular locks
            j["locked_axes"] = locked & 0x38; j["coupled_axes"] = 7
            j["limit_axes"] = (int(j["limit_axes"]) & 0x38) | 1; j["limits"][0] = (0.0, float(rng.uniform(0.3, 1.2)))
            j["motor_axes"] = int(j["motor_axes"]) & 0x38
        elif r < 0.5:       # SpringJoint, sometimes with a travel limit on top
            j["locked_axes"] = locked & 0x38; j["coupled_axes"] = 7
            j["limit_axes"] = int(j["limit_axes"]) & 0x38
            if rng.random() < 0.4:
                j["limit_axes"] = int(j["limit_axes"]) | 1; j["limits"][0] = (0.0, float(rng.uniform(0.6, 1.5)))
            j["motor_axes"] = (int(j["motor_axes"]) & 0x38) | 1
            j["motors"][0] = S.motor_desc(target_pos=float(rng.uniform(0.2, 0.8)), stiffness=float(rng.uniform(50.0, 400.0)), damping=float(rng.uniform(1.0, 10.0)), model=int(rng.integers(0, 2)))
        elif r < 0.67 and (locked & 0x38) == 0:   # two coupled angular axes: a cone limit on the third axis' swing (+ a no-op motor on one of them)
            pair = [(3, 4), (3, 5), (4, 5)][int(rng.integers(0, 3))]
            j["coupled_axes"] = (1 << pair[0]) | (1 << pair[1])
            j["limit_axes"] = (int(j["limit_axes"]) & 7) | (1 << pair[0]); j["limits"][pair[0]] = (0.0, float(rng.uniform(0.2, 0.9)))
            j["motor_axes"] = (int(j["motor_axes"]) & 7) | ((1 << pair[1]) if rng.random() < 0.5 else 0)


def _run(seed, steps=240, walls=False, params=False, world=None, extras=False, sensors=False, halfspace=False, batch=False, coupled=False, **kw):
    sc, rng = _scene(seed, **kw)
    if coupled:
        _couple_some_joints(sc, np.random.default_rng(seed + 777))
    if halfspace:
        c0 = sc.colliders[0]                                       # the ground slab (top face at y = 0) becomes the plane y = 0
        c0["shape"] = S.SHAPE_HALFSPACE; c0["half_extents"] = (0.0, 1.0, 0.0); c0["translation"] = (0.0, 0.5, 0.0)
        sc.add_collider(0, shape=S.SHAPE_HALFSPACE, half_extents=(-0.6, 0.0, 0.8), translation=(5.0, 0.5, -5.0), friction=0.3)
    if sensors:
        trig = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, 1.2, 0.0))
        sc.add_collider(trig, half_extents=(2.5, 0.8, 2.5), sensor=1, active_events=S.ACTIVE_EVENTS_COLLISION)
        for c in range(len(sc.colliders) - 1):
            parent = sc.collider_parents[c]
            first = c == 0 or sc.collider_parents[c - 1] != parent
            if c in (1, 2) or (not first and rng.random() < 0.5):   # two of the fixed obstacles, half of the extra colliders of compound bodies
                sc.colliders[c]["sensor"] = 1
                sc.colliders[c]["active_events"] = S.ACTIVE_EVENTS_COLLISION
    if params:
        _random_params(sc, rng)
    if walls:
        for k, (x, z, hx, hz) in enumerate(((3.2, 0, 0.3, 3.5), (-3.2, 0, 0.3, 3.5), (0, 3.2, 3.5, 0.3), (0, -3.2, 3.5, 0.3))):
            sc.add_collider(0, half_extents=(hx, 6.0, hz), translation=(x, 6.5, z))
    dyn = [i for i, b in enumerate(sc.bodies) if int(b["body_type"]) == S.BODY_DYNAMIC]
    if batch:
        # the fuzzed scene as sub-world 0 of a BATCH (rp_world_begin_subworld): two more random scenes occupy the same space as
        # sub-worlds 1 and 2 — they never meet the fuzzed one — and everything the driver inserts later lands in sub-world 2
        others = [_scene(seed + 500 + k, **kw)[0] for k in range(2)]
        for other in others:
            other.gravity, other.params = sc.gravity, sc.params.copy()
        sc = S.batch([sc] + others)
    g, o = (world(sc) if world else PhysicsWorld.from_scene(sc)), OracleWorld(sc)
    nb0 = len(sc.bodies)
    alive = list(range(nb0))
    jb = {j: (int(sc.joints[j]["body1"]), int(sc.joints[j]["body2"])) for j in range(len(sc.joints))}   # live joints -> their bodies
    col_parent = list(sc.collider_parents); ncol = len(col_parent); removed_cols = set()

    def note_collider(idx, parent):   # a new collider takes the slot removed last (arena reuse) or a fresh row
        nonlocal ncol
        if idx < ncol:
            col_parent[idx] = parent; removed_cols.discard(idx)
        else:
            assert idx == ncol
            col_parent.append(parent); ncol += 1
    log = []
    if extras:
        for b in dyn:
            if rng.random() < 0.33:
                n_extra = int(rng.choice([1, 2, 4, 7]))
                g.set_additional_solver_iterations([b], n_extra); o.set_additional_solver_iterations(b, n_extra)
    for step in range(1, steps + 1):
        if step % 3 == 0:                                         # the position-based platform follows a script
            t = step / 60.0
            kp = np.array([3.0 - 0.5 * t, 0.3 + 0.1 * np.sin(t), -3.0 + 0.4 * t, 0.0, np.sin(0.15 * t), 0.0, np.cos(0.15 * t)], np.float32)
            g.set_next_kinematic_position([5], kp); o.set_next_kinematic_position(5, kp)
        if step % 7 == 0:                                         # a random user action
            act = int(rng.integers(0, 13))
            live_dyn = [b for b in dyn if b in alive]
            b = int(rng.choice(live_dyn))
            log.append((step, act, b))
            if act == 0:
                imp = rng.uniform(-3, 3, size=3).astype(np.float32)
                g.apply_impulse([b], impulse=[imp]); o.apply_impulse(b, impulse=imp)
            elif act == 1:
                v = rng.uniform(-2, 2, size=6).astype(np.float32)
                g.write_bodies([b], vel6=[v]); o.set_vel(b, v[:3], v[3:])
            elif act == 2:
                p = np.array([rng.uniform(-3, 3), rng.uniform(2, 6), rng.uniform(-3, 3), 0, 0, 0, 1], np.float32)
                g.write_bodies([b], pos7=[p]); o.set_pose(b, p)
            elif act == 3:
                g.wake_up([b]); o.wake_up(b)
            elif act == 4 and len(live_dyn) > 10:
                g.remove_body(b); o.remove_body(b); alive.remove(b)
                removed_cols.update(c for c in range(ncol) if col_parent[c] == b)   # its colliders go with it
                jb = {j: bb for j, bb in jb.items() if b not in bb}       # RigidBodySet::remove takes the attached joints along
            elif act == 5:
                body = S.body_desc(translation=(float(rng.uniform(-2, 2)), float(rng.uniform(4, 7)), float(rng.uniform(-2, 2))), rotation=_rand_quat(rng), can_sleep=1)
                col = S.collider_desc(**_rand_collider(rng))
                hb = g.insert_body(body); hc = g.insert_collider(col, hb)
                ob = lib().ro_add_body(o._w, np.array([body], S.BODY_DTYPE).ctypes.data); o.n += 1
                oc = lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
                assert int(hb) & 0xFFFFFFFF == ob and int(hc) & 0xFFFFFFFF == oc
                alive.append(ob); dyn.append(ob)
                note_collider(oc, ob)
            elif act == 6 and jb:
                j = int(rng.choice(sorted(jb)))
                g.remove_impulse_joint(j); o.remove_joint(j); del jb[j]
            elif act == 7 and len(live_dyn) > 2:
                b2 = int(rng.choice([x for x in live_dyn if x != b]))
                jd = np.zeros((), S.JOINT_DTYPE)
                jd["body1"], jd["body2"] = b, b2
                jd["local_anchor1"], jd["local_anchor2"] = (0.0, 0.5, 0.0), (0.0, -0.5, 0.0)
                jd["local_basis1"] = jd["local_basis2"] = (0, 0, 0, 1)
                jd["locked_axes"], jd["contacts_enabled"] = S.LOCK_LIN, int(rng.random() < 0.5)
                for k in range(6):
                    jd["motors"][k] = S.motor_desc()
                hj = g.insert_impulse_joint(b, b2, jd)
                oj = lib().ro_add_joint(o._w, np.array([jd], S.JOINT_DTYPE).ctypes.data)
                assert (int(g.joint_handles()[oj]) if hasattr(g, "joint_handles") else hj) == hj   # (a generational handle; the fuzz names joints by their insertion ordinal, like the oracle)
                jb[oj] = (b, b2)
            elif act == 9:
                f = rng.uniform(-5, 5, size=3).astype(np.float32); tq = rng.uniform(-1, 1, size=3).astype(np.float32)
                reset = bool(rng.random() < 0.5)
                g.add_force([b], force=[f], torque=[tq], reset=reset); o.add_force(b, force=f, torque=tq, reset=reset)
            elif act == 10:
                cols = [c for c in range(ncol) if col_parent[c] == b and c not in removed_cols]
                if len(cols) > 1:                                 # drop one collider of a compound body
                    c = int(rng.choice(cols))
                    g.remove_collider(c); o.remove_collider(c); removed_cols.add(c)
            elif act == 11:                                       # attach another collider to a live body
                col = S.collider_desc(translation=tuple(float(x) for x in rng.uniform(-0.4, 0.4, size=3)), **_rand_collider(rng, 0.6))
                hc = g.insert_collider(col, b)
                oc = lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, b)
                assert int(hc) & 0xFFFFFFFF == oc
                note_collider(oc, b)
            elif act == 12:
                tqi = rng.uniform(-0.5, 0.5, size=3).astype(np.float32)
                g.apply_impulse([b], torque_impulse=[tqi]); o.apply_impulse(b, torque_impulse=tqi)
            elif act == 8 and jb:
                j = int(rng.choice(sorted(jb)))
                kw = dict(target_vel=float(rng.uniform(-2, 2)), damping=float(rng.uniform(1, 10)))
                axis = int(rng.integers(0, 6))
                g.set_joint_motor(j, axis, **kw); o.set_joint_motor(j, axis, **kw)
        g.step(1); o.step(1)
        if os.environ.get("RP_FUZZ_WATCH"):
            wb, ws = (int(x) for x in os.environ["RP_FUZZ_WATCH"].split(":"))
            if os.environ.get("RP_FUZZ_WATCH_ISL") and 180 <= step <= ws + 1:
                gi_, oi_ = g.island_labels(), o.island_labels()
                print(f"ISL step {step}: gpu {[int(gi_[x]) for x in (wb, 43, len(gi_) - 1)]} ora {[int(oi_[x]) for x in (wb, 43, len(oi_) - 1)]} stats gpu {g.island_stats()} ora {o.island_stats() if hasattr(o, 'island_stats') else None}")
            if ws - 6 <= step <= ws + 1:
                gp_, gv_ = g.read_bodies(); op_, ov_ = o.read()
                print(f"POSE step {step} body {wb}: gpu v {np.round(gv_[wb], 3).tolist()} | ora v {np.round(ov_[wb], 3).tolist()} | others moved fast: {[(i_, np.round(ov_[i_][:3], 1).tolist(), np.round(gv_[i_][:3], 1).tolist()) for i_ in range(len(ov_)) if abs(ov_[i_][:3]).max() > 20]}")
            if abs(step - ws) <= 1:
                print(f"WATCH step {step} body {wb}: sleeping gpu {bool(g.sleeping()[wb])} ora {bool(o.sleeping()[wb])} pairs gpu {g.counters()['num_pairs']} ora {o.stats()['num_pairs']}")
        if step % 10 == 0 or step < 4 or TRACE:
            try:
                _check(g, o, alive, f"seed {seed} step {step}")
            except AssertionError:
                if TRACE:
                    gp, gv = g.read_bodies(); op, ov = o.read()
                    bad = [b for b in alive if (gp[b] != op[b]).any() or (gv[b] != ov[b]).any()]
                    print(f"TRACE seed {seed}: first divergence at step {step}, bodies {bad[:10]}, types {[int(sc.bodies[b]['body_type']) if b < len(sc.bodies) else 0 for b in bad[:10]]}; last actions {log[-6:]}; edits {[a for a in log if a[1] in (4, 5, 6, 7, 10, 11)]}")
                    gc_, os_ = g.counters(), o.stats()
                    print("TRACE counters", {k: gc_[k] for k in ("num_pairs", "num_manifolds", "num_solver_contacts", "num_colors")}, os_)
                    gm, gn, gi = g.contacts(); om, on, oi = o.manifolds()
                    gk = {(a, b_): (c, n, tuple(i)) for (a, b_, c, n), i in zip(gm.tolist(), gi.tolist())}
                    ok = {(a, b_): (c, n, tuple(i)) for (a, b_, c, n), i in zip(om.tolist(), oi.tolist())}
                    for key in sorted(set(gk) | set(ok)):
                        if gk.get(key) != ok.get(key):
                            print(f"TRACE manifold {key}: gpu {gk.get(key)} ora {ok.get(key)}")
                    gs_, os2_ = g.sleeping(), o.sleeping()
                    print("TRACE sleeping differs for", [b_ for b_ in alive if gs_[b_] != os2_[b_]], "bad body sleeping gpu/ora", [(b_, bool(gs_[b_]), bool(os2_[b_])) for b_ in bad[:4]])
                    for b_ in bad[:2]:
                        print("TRACE body", b_, "gpu", gp[b_], gv[b_], "\nTRACE       ora", op[b_], ov[b_])
                    try:
                        print("TRACE islands gpu", g.island_labels()[bad[:3]].tolist(), "ora", o.island_labels()[bad[:3]].tolist())
                    except Exception as e_:
                        print("TRACE islands n/a", e_)
                    gjc, gji = g.read_joints(); ojc, oji = o.read_joints()
                    print("TRACE joints live", jb, "impulse rows that differ", [(j_, gji[j_].tolist(), oji[j_].tolist()) for j_ in range(len(gji)) if (gji[j_] != oji[j_]).any()][:6], "all joints of the scene touching the bad bodies", [(j_, int(sc.joints[j_]['body1']), int(sc.joints[j_]['body2'])) for j_ in range(len(sc.joints)) if int(sc.joints[j_]['body1']) in bad or int(sc.joints[j_]['body2']) in bad])
                    print("TRACE manifold keys", sorted(gk))
                    print("TRACE scene collider parents", list(enumerate(sc.collider_parents))[30:], "nb0", nb0, "joints", {j: bb for j, bb in jb.items() if 34 in bb or 43 in bb})
                    from collections import Counter
                    print("TRACE gpu manifold keys seen more than once:", [(k, v) for k, v in Counter((a, b_) for a, b_, c, n in gm.tolist()).items() if v > 1], "gpu manifolds", len(gm), "oracle", len(om))
                    print("TRACE collider parents of the duplicated:", [(int(h) & 0xFFFFFFFF) for h in g.collider_handles() if int(h) >> 32], "col_parent tail", col_parent[-4:], "removed cols", sorted(removed_cols))
                    print("TRACE handles", [hex(int(h)) for h in g.body_handles()[-6:]], [hex(int(h)) for h in g.collider_handles() if int(h) >> 32])
                raise
    c = g.counters()
    assert c["overflow_flags"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [40, 41, 42, 43, 2011])
def test_fuzz_cylinders_and_cones_bit_exact(seed):
    """the same scenes and actions with cylinders and cones in the draw (rp_convex.h: GJK / EPA manifolds under joints, both friction
    models, events, sleeping, compound bodies, removals and insertions; seed 43 runs the Coulomb model)"""
    try:
        _run(seed, steps=200, params=seed >= 2000, convex=True)
    finally:
        _CONVEX[0] = False


@pytest.mark.parametrize("seed", [40, 43])
def test_fuzz_cylinders_and_cones_on_the_oracle_twin(seed):
    try:
        _run(seed, steps=120, world=OracleTwin, convex=True)
    finally:
        _CONVEX[0] = False


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [50, 51, 52, 2013])
def test_fuzz_convex_polyhedra_bit_exact(seed):
    """... and with four registered convex polyhedra in the draw (shared between bodies, parts of compound bodies, attached to and
    removed from live bodies)"""
    try:
        _run(seed, steps=200, params=seed >= 2000, convex=5)
    finally:
        _CONVEX[0] = False


@pytest.mark.parametrize("seed", [50])
def test_fuzz_convex_polyhedra_on_the_oracle_twin(seed):
    try:
        _run(seed, steps=120, world=OracleTwin, convex=5)
    finally:
        _CONVEX[0] = False


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [60, 61, 62, 2019])
def test_fuzz_round_shapes_bit_exact(seed):
    """... and with the round variants (RoundShape<S>: border radii through the same GJK / EPA path) in the draw, next to five registered
    polyhedra, cylinders and cones"""
    try:
        _run(seed, steps=200, params=seed >= 2000, convex=6)
    finally:
        _CONVEX[0] = False


@pytest.mark.parametrize("seed", [60])
def test_fuzz_round_shapes_on_the_oracle_twin(seed):
    try:
        _run(seed, steps=120, world=OracleTwin, convex=6)
    finally:
        _CONVEX[0] = False
