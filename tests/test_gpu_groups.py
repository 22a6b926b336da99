"""Substep solve-groups on the device (RigidBody::additional_solver_iterations; island_manager/substep_groups.rs,
staged_island_solver/init.rs:52-100): the scenes of additional_solver_iterations.rs, substep_chain_high_mass_ratio.rs and
test_staged.rs:214-330 through the C ABI, bit for bit against the oracle (which passes those tests' own assertions in
tests/test_reference_kats.py)."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld
from test_reference_kats import _cube, ground, stack, world

pytestmark = pytest.mark.gpu


def _lockstep(sc, checkpoints):
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    done = 0
    for cp in checkpoints:
        g.step(cp - done); o.step(cp - done); done = cp
        gp, gv = g.read_bodies(); op, ov = o.read()
        np.testing.assert_array_equal(gp, op, err_msg=f"poses @ {cp}")
        np.testing.assert_array_equal(gv, ov, err_msg=f"velocities @ {cp}")
    assert g.counters()["overflow_flags"] == 0
    return g, o


def _heavy_stack(extra):
    sc = world()
    g = sc.add_body(body_type=S.BODY_FIXED)
    sc.add_collider(g, half_extents=(10.0, 0.5, 10.0), translation=(0.0, -0.5, 0.0))
    light = sc.add_body(translation=(0.0, 0.5, 0.0), can_sleep=1)
    sc.add_collider(light, half_extents=(0.5, 0.5, 0.5), density=1.0)
    heavy = sc.add_body(translation=(0.1, 3.0, 0.0), can_sleep=1, additional_solver_iterations=extra)
    sc.add_collider(heavy, half_extents=(0.5, 0.5, 0.5), density=200.0)
    return sc, light, heavy


@pytest.mark.parametrize("extra", [0, 8, 16])
def test_heavy_stack_bit_exact(extra):
    sc, light, heavy = _heavy_stack(extra)
    g, _ = _lockstep(sc, [1, 10, 60, 300])
    pos, vel = g.read_bodies()
    if extra == 16:  # additional_solver_iterations.rs:115-141: the 200:1 stack holds
        assert 0.3 < pos[light, 1] < 0.7 and 1.2 < pos[heavy, 1] < 1.8


def test_heavy_chain_bit_exact():
    """additional_solver_iterations.rs:143-164: a six-link rope with a 100x heavier end ball, extra iterations on the weight."""
    sc = world()
    prev = sc.add_body(body_type=S.BODY_FIXED)
    for i in range(6):
        link = sc.add_body(translation=(0.0, -(i + 1.0), 0.0), can_sleep=1, additional_solver_iterations=16 if i == 5 else 0)
        sc.add_collider(link, shape=S.SHAPE_BALL, half_extents=(0.4, 0.0, 0.0), density=100.0 if i == 5 else 1.0)
        sc.add_joint(prev, link, (0.0, -0.5, 0.0), (0.0, 0.5, 0.0), locked_axes=S.LOCK_LIN)
        prev = link
    g, _ = _lockstep(sc, [1, 5, 50, 300])
    end = g.read_bodies()[0][prev, :3]
    assert np.linalg.norm(end) < 20.0 and -7.5 < end[1] < -4.5


def test_high_mass_ratio_chain_bit_exact():
    """substep_chain_high_mass_ratio.rs: a 16-link chain with a 1000:1 end ball, 16 extra substeps on every body."""
    sc = world()
    rad, num = 0.2, 17
    prev = None
    for i in range(num):
        ball_rad = rad * 10.0 if i == num - 1 else rad
        shift1, shift2 = rad * 1.1, ball_rad + rad * 0.1
        z = 0.0 if i == 0 else (i - 1.0) * 2.0 * shift1 + shift1 + shift2
        b = sc.add_body(body_type=S.BODY_FIXED if i == 0 else S.BODY_DYNAMIC, translation=(0.0, 0.0, z), additional_solver_iterations=0 if i == 0 else 16)
        sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(ball_rad, 0.0, 0.0))
        if prev is not None:
            a1, a2 = ((0.0, 0.0, 0.0), (0.0, 0.0, -shift1 * 2.0)) if i == 1 else ((0.0, 0.0, shift1), (0.0, 0.0, -shift2))
            sc.add_joint(prev, b, a1, a2, locked_axes=S.LOCK_LIN)
        prev = b
    _lockstep(sc, [1, 10, 100, 300])


def test_three_groups_and_runtime_changes_bit_exact():
    """test_staged.rs:214-330: {lone} = 8, {chain_a, chain_b} = 4 (the joint lifts the partner), {plain} = 0 — then the counts are
    changed at run time (RigidBody::set_additional_solver_iterations) down to the single implicit group and up again."""
    sc = world()
    g0 = sc.add_body(body_type=S.BODY_FIXED)
    sc.add_collider(g0, half_extents=(0.5, 0.5, 0.5))
    plain = _cube(sc, (0.0, 1.001, 0.0), can_sleep=0)
    chain_a = _cube(sc, (10.0, 5.0, 0.0), can_sleep=0)
    chain_b = _cube(sc, (10.0, 3.0, 0.0), can_sleep=0)
    sc.add_joint(chain_a, chain_b, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), locked_axes=S.LOCK_REVOLUTE, basis1=S.AXIS_Z_BASIS, basis2=S.AXIS_Z_BASIS)
    lone = _cube(sc, (-10.0, 5.0, 0.0), can_sleep=0)
    sc.bodies[chain_b]["additional_solver_iterations"] = 4
    sc.bodies[lone]["additional_solver_iterations"] = 8
    g, o = _lockstep(sc, [1, 3, 20])
    for counts in ((0, 0), (3, 0), (3, 11)):
        g.set_additional_solver_iterations([chain_b, lone], counts)
        o.set_additional_solver_iterations(chain_b, counts[0]); o.set_additional_solver_iterations(lone, counts[1])
        for _ in range(3):
            g.step(5); o.step(5)
            gp, gv = g.read_bodies(); op, ov = o.read()
            np.testing.assert_array_equal(gp, op); np.testing.assert_array_equal(gv, ov)


@pytest.mark.parametrize("coulomb", [False, True])
def test_groups_next_to_stacks_kinematic_platform_and_sleep(coulomb):
    """A default-cadence stack, an elevated jointed chain, an elevated heavy stack riding a kinematic platform (lifted to its
    group) and sleeping allowed — every piece of the group solver at once, both friction models."""
    sc = world()
    if coulomb:
        sc.params["friction_model"] = S.FRICTION_COULOMB
    ground(sc)
    stack(sc, 0.0, 4, can_sleep=1)
    prev = sc.add_body(body_type=S.BODY_FIXED, translation=(30.0, 10.0, 0.0))
    for i in range(5):
        b = sc.add_body(translation=(30.0 + (i + 1.0), 10.0, 0.0), additional_solver_iterations=6 if i == 4 else 0)
        sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.3, 0.0, 0.0), density=50.0 if i == 4 else 1.0)
        sc.add_joint(prev, b, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0), locked_axes=S.LOCK_LIN)
        prev = b
    plat = sc.add_body(body_type=S.BODY_KINEMATIC_VELOCITY, translation=(-20.0, 2.0, 0.0), linvel=(0.3, 0.0, 0.0))
    sc.add_collider(plat, half_extents=(2.0, 0.25, 2.0))
    lo = sc.add_body(translation=(-20.0, 2.75, 0.0), can_sleep=1)
    sc.add_collider(lo, half_extents=(0.5, 0.5, 0.5), density=1.0)
    hi = sc.add_body(translation=(-20.0, 3.75, 0.0), can_sleep=1, additional_solver_iterations=12)
    sc.add_collider(hi, half_extents=(0.5, 0.5, 0.5), density=150.0, restitution=0.3)
    g, o = _lockstep(sc, [1, 2, 10, 60, 200])
    assert np.array_equal(g.sleeping(), o.sleeping())
