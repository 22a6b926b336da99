"""GenericJoint::coupled_axes on the oracle (CPU): RopeJoint (rope_joint.rs:31-38), SpringJoint (spring_joint.rs:31-40) and two coupled
angular axes (joint_constraint_helper.rs:725-790) — outcome-level checks in the style of the reference's own joint tests, plus the
scene of crates/rapier3d/tests/issue_792_coupled_angular_spring.rs.  The device twin: tests/test_gpu_parity.py (lockstep)."""
import numpy as np

from rapier_amd import scenes as S
from oracle_ffi import OracleWorld


def _world():
    return S.Scene(name="coupled", gravity=(0.0, -9.81, 0.0))


def rope_scene(max_dist=2.0):
    s = _world()
    a = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, 5.0, 0.0))
    b = s.add_body(translation=(0.5, 4.5, 0.0), linvel=(2.0, 0.0, 0.5))
    s.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.25, 0.0, 0.0))
    s.add_rope_joint(a, b, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), max_dist)
    return s


def spring_scene(rest=1.0, k=200.0, c=4.0):
    s = _world()
    a = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, 5.0, 0.0))
    b = s.add_body(translation=(0.0, 3.2, 0.0))
    s.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.25, 0.0, 0.0))
    s.add_spring_joint(a, b, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), rest, k, c)
    return s


def cone_scene():
    """issue_792_coupled_angular_spring.rs: the linear axes locked, ANG_X | ANG_Z coupled with limits [0, 0.5] and (no-op) spring motors,
    a kinematic parent; the dynamic body is pushed sideways after 50 steps"""
    s = S.Scene(name="issue_792", gravity=(0.0, -9.81, 0.0))
    k = s.add_body(body_type=S.BODY_KINEMATIC_POSITION)
    s.add_collider(k, shape=S.SHAPE_BALL, half_extents=(1.0, 0.0, 0.0))
    d = s.add_body(translation=(0.0, -5.0, 0.0), linvel=(0.1, 0.0, 0.1))
    s.add_collider(d, shape=S.SHAPE_BALL, half_extents=(1.0, 0.0, 0.0))
    s.add_joint(k, d, (0.0, 0.0, 0.0), (0.0, -5.0, 0.0), locked_axes=S.LOCK_LIN, limits={3: (0.0, 0.5), 5: (0.0, 0.5)},
                motors={3: dict(target_pos=0.0, stiffness=0.0, damping=0.5), 5: dict(target_pos=0.0, stiffness=0.0, damping=0.5)},
                coupled_axes=(1 << 3) | (1 << 5))
    return s, d


def test_rope_holds_the_maximum_distance_and_is_slack_below_it():
    w = OracleWorld(rope_scene(2.0))
    far, slack_seen = 0.0, False
    for _ in range(600):
        w.step(1)
        pos, _ = w.read()
        d = float(np.linalg.norm(pos[1, :3] - pos[0, :3]))
        far = max(far, d); slack_seen |= d < 1.5
    assert np.isfinite(pos).all()
    assert 1.98 < far < 2.03, far            # taut at the limit (soft constraint: a little stretch), never beyond
    assert slack_seen                          # free inside the sphere: it started 0.7 from the anchor


def test_spring_oscillates_about_its_loaded_length_and_settles():
    rest, k, c = 1.0, 200.0, 4.0
    sc = spring_scene(rest, k, c)
    w = OracleWorld(sc)
    m = 1000.0 * 0 + float(4.0 / 3.0 * np.pi * 0.25 ** 3 * sc.colliders[0]["density"])
    target = rest + m * 9.81 / k
    lens = []
    for _ in range(1500):
        w.step(1)
        pos, vel = w.read()
        lens.append(float(np.linalg.norm(pos[1, :3] - pos[0, :3])))
    assert min(lens[:120]) < target < max(lens[:120])          # it swings through the loaded length
    assert abs(lens[-1] - target) < 0.02 and abs(float(vel[1, 1])) < 0.02, (lens[-1], target)


def test_two_coupled_angular_axes_limit_the_cone_and_the_issue_792_scene_runs():
    sc, d = cone_scene()
    w = OracleWorld(sc)
    worst = 0.0
    for i in range(300):
        if i == 50:
            w.apply_impulse(d, (5.0, 0.0, 0.0), (0.0, 0.0, 0.0)) if hasattr(w, "apply_impulse") else None
        w.step(1)
        pos, _ = w.read()
        assert np.isfinite(pos).all()
        q = pos[d, 3:7]
        y = np.array([2 * (q[0] * q[1] - q[3] * q[2]), 1 - 2 * (q[0] ** 2 + q[2] ** 2), 2 * (q[1] * q[2] + q[3] * q[0])])   # the body's Y axis
        worst = max(worst, float(np.arccos(np.clip(y[1], -1.0, 1.0))))
    assert worst < 0.6, worst                  # the cone of half-angle 0.5 (soft: a little overshoot)
    assert worst > 0.2                         # ... and it really swung out after the push


def test_row_count_follows_the_reference_rules():
    """a coupled linear motor + limit on top of locked angular axes: 3 locks + 1 coupled motor + 1 coupled limit rows, none per coupled axis"""
    s = _world()
    a = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, 5.0, 0.0))
    b = s.add_body(translation=(0.0, 3.0, 0.0))
    s.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.25, 0.0, 0.0))
    s.add_joint(a, b, (0, 0, 0), (0, 0, 0), locked_axes=0b111000, coupled_axes=S.LOCK_LIN, limits={0: (0.0, 2.5)},
                motors={0: dict(target_pos=1.5, stiffness=80.0, damping=6.0, model=S.MOTOR_FORCE_BASED)})
    w = OracleWorld(s)
    w.step(400)
    pos, vel = w.read()
    assert np.isfinite(pos).all() and abs(float(np.linalg.norm(pos[1, :3] - pos[0, :3])) - 1.5) < 0.35
    assert abs(float(pos[1, 6])) > 0.999       # the locked angular axes held
