"""The stepping regression tests the reference keeps next to its pipeline (src/pipeline/physics_pipeline/test.rs) restated with their
scenes, step counts and assertions: on the oracle (CPU) and — marked gpu — on the device through the C ABI, where the result must
also equal the oracle's bit for bit.  These are the corner cases of `PhysicsPipeline::step()`: worlds without a single collider or
contact, bodies / colliders removed before the first step, a kinematic body overlapping a fixed one, dt = 0 with a joint, user forces.
Not restated: `rigid_body_removal_snapshot_handle_determinism` (serde), `ccd_respects_filter_contact_pair_hook` (hooks: no hook
crosses this ABI), `rigid_body_type_changed_dynamic_is_in_active_set` / `contact_force_events_follow_runtime_active_events_flips`
(set_body_type / runtime ActiveEvents flips are not in the ABI), `test_multi_sap_disable_body` (dim2)."""
import numpy as np
import pytest

from rapier_amd import scenes as S
from oracle_ffi import OracleWorld


def _device(scene):
    from rapier_amd import PhysicsWorld
    return PhysicsWorld.from_scene(scene)


class _Oracle:
    """the oracle behind the calls the tests below make on a world (body / collider indices are the handles' indices)"""

    def __init__(self, scene):
        self.o = OracleWorld(scene)

    def step(self, n=1): self.o.step(n)
    def read(self): return self.o.read()
    def remove_body(self, b): self.o.remove_body(b)
    def remove_collider(self, c): self.o.remove_collider(c)
    def add_force(self, b, force, reset=False): self.o.add_force(b, force=force, reset=reset)


class _Device:
    def __init__(self, scene):
        self.w = _device(scene)

    def step(self, n=1): self.w.step(n)
    def read(self): return self.w.read_bodies()
    def remove_body(self, b): self.w.remove_body([b])
    def remove_collider(self, c): self.w.remove_collider([c])
    def add_force(self, b, force, reset=False): self.w.add_force([b], force=force, reset=reset)


def _worlds(scene, gpu):
    return [_Oracle(scene)] + ([_Device(scene)] if gpu else [])


def _same(ws, what):
    """every world finite; the device equal to the oracle bit for bit"""
    states = [w.read() for w in ws]
    for p, v in states:
        assert np.isfinite(p).all() and np.isfinite(v).all(), what
    for p, v in states[1:]:
        np.testing.assert_array_equal(p, states[0][0], err_msg=what + " poses")
        np.testing.assert_array_equal(v, states[0][1], err_msg=what + " velocities")
    return states[0]


# ---- scenes -----------------------------------------------------------------------------------------------------------------------
def _kinematic_and_fixed():
    """test.rs:15-48: a fixed and a position-based kinematic body at the origin, each with a ball of radius 10; no gravity; one step"""
    s = S.Scene(name="kinematic_and_fixed_contact", gravity=(0.0, 0.0, 0.0))
    h1 = s.add_body(body_type=S.BODY_FIXED)
    s.add_collider(h1, shape=S.SHAPE_BALL, half_extents=(10.0, 0, 0))
    h2 = s.add_body(body_type=S.BODY_KINEMATIC_POSITION)
    s.add_collider(h2, shape=S.SHAPE_BALL, half_extents=(10.0, 0, 0))
    return s


def _four_bare_bodies():
    """test.rs:52-104: two dynamic bodies, a position-based kinematic body and a fixed body — no colliders — removed before the first step"""
    s = S.Scene(name="removal_before_step", gravity=(0.0, 0.0, 0.0))
    s.add_body(); s.add_body(); s.add_body(body_type=S.BODY_KINEMATIC_POSITION); s.add_body(body_type=S.BODY_FIXED)
    return s


def _ball_body():
    """test.rs:260-304: one dynamic body with a unit ball, gravity -9.81"""
    s = S.Scene(name="collider_removal_before_step", gravity=(0.0, -9.81, 0.0))
    b = s.add_body()
    s.add_collider(b, shape=S.SHAPE_BALL, half_extents=(1.0, 0, 0))
    return s


def _joint_dt0():
    """test.rs:375-440: a fixed and a dynamic body (additional mass 1, no colliders) on a revolute joint about Z, anchors (0, 1, 0) /
    (0, -3, 0); IntegrationParameters { dt: 0.0, .. }"""
    s = S.Scene(name="joint_step_delta_time_0", gravity=(0.0, -9.81, 0.0))
    s.params["dt"] = 0.0
    h = s.add_body(body_type=S.BODY_FIXED, additional_mass=1.0)
    d = s.add_body(additional_mass=1.0)
    s.add_joint(h, d, (0.0, 1.0, 0.0), (0.0, -3.0, 0.0), locked_axes=S.LOCK_REVOLUTE, basis1=S.AXIS_Z_BASIS, basis2=S.AXIS_Z_BASIS)
    return s


def _unit_mass_body():
    """test.rs:538-620: one collider-less dynamic body of additional mass 1, no gravity"""
    s = S.Scene(name="user_force_persists", gravity=(0.0, 0.0, 0.0))
    s.add_body(additional_mass=1.0)
    return s


# ---- the tests, on the oracle alone and on oracle + device ---------------------------------------------------------------------------
GPU = [pytest.param(False, id="oracle"), pytest.param(True, id="device", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("gpu", GPU)
def test_kinematic_and_fixed_contact_crash(gpu):
    ws = _worlds(_kinematic_and_fixed(), gpu)
    for w in ws:
        w.step(1)
    p, v = _same(ws, "one step")
    np.testing.assert_array_equal(p[:, :3], 0.0)   # neither body is pushed: the pair has no dynamic side (update.rs:334-396)
    np.testing.assert_array_equal(v, 0.0)


@pytest.mark.parametrize("gpu", GPU)
def test_rigid_body_removal_before_step(gpu):
    ws = _worlds(_four_bare_bodies(), gpu)
    for w in ws:
        for b in range(4):
            w.remove_body(b)
        w.step(1)
    _same(ws, "stepping a world whose bodies were all removed")


@pytest.mark.parametrize("gpu", GPU)
def test_collider_removal_before_step(gpu):
    ws = _worlds(_ball_body(), gpu)
    for w in ws:
        w.remove_collider(0)
        w.remove_body(0)
        w.step(10)
    _same(ws, "ten steps after removing the only collider, then its body")


@pytest.mark.parametrize("gpu", GPU)
def test_joint_step_delta_time_0(gpu):
    ws = _worlds(_joint_dt0(), gpu)
    for w in ws:
        w.step(1)
    p, v = _same(ws, "dt = 0 with a revolute joint")   # (the reference asserts exactly this: every pose component finite)
    np.testing.assert_array_equal(p[1, :3], 0.0)        # and with dt = 0 nothing may move


@pytest.mark.parametrize("gpu", GPU)
def test_user_force_persists_across_steps(gpu):
    ws = _worlds(_unit_mass_body(), gpu)
    for w in ws:
        w.add_force(0, (1.0, 0.0, 0.0))
        w.step(1)
    v1 = _same(ws, "first step")[1][0, 0]
    for w in ws:
        w.step(1)                                      # (the force is NOT added again: it persists, issue #903)
    v2 = _same(ws, "second step")[1][0, 0]
    assert v1 > 0.0 and abs(v2 - 2.0 * v1) < 1.0e-5, (v1, v2)
    for w in ws:
        w.add_force(0, None, reset=True)               # RigidBody::reset_forces(true)
        w.step(1)
    v3 = _same(ws, "after reset_forces")[1][0, 0]
    assert abs(v3 - v2) < 1.0e-5, (v2, v3)


def _bare_chain(can_sleep=0):
    """not a reference scene: a world WITHOUT ANY COLLIDER that still has dynamics — a chain of six point masses (additional mass, principal
    inertia from nothing: the reference's collider-less joint tests above build their bodies this way) hanging from a fixed body on
    spherical joints, the last link on a revolute joint, one free body under a user force"""
    s = S.Scene(name=f"bare_chain_{can_sleep}", gravity=(0.0, -9.81, 0.0))
    prev = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, 10.0, 0.0))
    for k in range(6):
        b = s.add_body(translation=(1.0 * (k + 1), 10.0, 0.0), additional_mass=1.0 + 0.5 * k, can_sleep=can_sleep)
        if k < 5:
            s.add_joint(prev, b, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0))
        else:
            s.add_joint(prev, b, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0), locked_axes=S.LOCK_REVOLUTE, basis1=S.AXIS_Z_BASIS, basis2=S.AXIS_Z_BASIS)
        prev = b
    s.add_body(translation=(-5.0, 2.0, 0.0), additional_mass=2.0, linvel=(0.0, 3.0, 1.0), can_sleep=can_sleep)
    return s


@pytest.mark.parametrize("gpu", GPU)
@pytest.mark.parametrize("can_sleep", [0, 1])
def test_world_without_colliders_still_moves(gpu, can_sleep):
    ws = _worlds(_bare_chain(can_sleep), gpu)
    start = ws[0].read()[0].copy()
    for w in ws:
        w.add_force(7, (0.5, 0.0, 0.0))
    for n in (1, 20, 200):
        for w in ws:
            w.step(n)
        p, v = _same(ws, f"{n} more steps")
    assert np.abs(p[7, :3] - start[7, :3]).max() > 1.0, p                      # the free body flies (gravity + the user force)
    # point masses have no angular inertia (additional_mass adds none): they cannot turn, so the joints make the chain rigid — it must
    # hang where it was built, every link within the joints' softness of its place
    assert np.abs(p[1:7, :3] - start[1:7, :3]).max() < 2e-2, p


def test_physics_world_builds_steps_and_queries_a_scene_oracle():
    _bouncing_ball_check([_Oracle(_bouncing_ball())])


@pytest.mark.gpu
def test_physics_world_builds_steps_and_queries_a_scene_device():
    _bouncing_ball_check(_worlds(_bouncing_ball(), True))


def _bouncing_ball():
    """crates/rapier3d/tests/issue_836_physics_world.rs: a ball (radius 0.5, restitution 0.7) dropped from y = 10 onto a thin ground slab
    (half-height 0.1), 200 steps (the ray cast at the end of the reference test belongs to the query pipeline: out of scope)"""
    s = S.Scene(name="issue_836", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=S.BODY_FIXED)
    s.add_collider(g, half_extents=(100.0, 0.1, 100.0))
    b = s.add_body(translation=(0.0, 10.0, 0.0), can_sleep=1)
    s.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0, 0), restitution=0.7)
    return s


def _bouncing_ball_check(ws):
    min_y = np.inf
    for _ in range(200):
        for w in ws:
            w.step(1)
        min_y = min(min_y, float(_same(ws, "bouncing ball")[0][1, 1]))
    final_y = float(ws[0].read()[0][1, 1])
    assert final_y < 9.0 and min_y > 0.4, (final_y, min_y)


# ---- degenerate worlds (not reference scenes): whatever the world holds — or does not hold — a step must leave the device equal to the oracle ----
def _degenerate(kind):
    s = S.Scene(name=f"degenerate_{kind}", gravity=(0.0, -9.81, 0.0))
    if kind == "empty":
        pass
    elif kind == "only_fixed":
        for k in range(3):
            f = s.add_body(body_type=S.BODY_FIXED, translation=(0.4 * k, 0.0, 0.0))
            s.add_collider(f, half_extents=(0.5, 0.5, 0.5))
    elif kind == "one_falling_ball":
        b = s.add_body(translation=(0.0, 5.0, 0.0), angvel=(1.0, 2.0, 3.0))
        s.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0, 0))
    elif kind == "bare_bodies_and_a_fixed_collider":
        g = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
        s.add_collider(g, half_extents=(5.0, 0.5, 5.0))
        for k in range(4):
            s.add_body(translation=(float(k), 3.0, 0.0), additional_mass=1.0, linvel=(0.0, 1.0, 0.0))
    elif kind == "dt0_stack":
        s = S.box_stack(3)
        s.params["dt"] = 0.0
    elif kind == "no_gravity_at_rest":
        s = S.box_stack(3)
        s.gravity = (0.0, 0.0, 0.0)
    elif kind == "far_from_the_origin":
        g = s.add_body(body_type=S.BODY_FIXED, translation=(3.0e5, -0.5, -2.0e5))
        s.add_collider(g, half_extents=(5.0, 0.5, 5.0))
        for k in range(3):
            b = s.add_body(translation=(3.0e5, 0.5 + 1.01 * k, -2.0e5))
            s.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    elif kind == "massless":
        g = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
        s.add_collider(g, half_extents=(5.0, 0.5, 5.0))
        b = s.add_body(translation=(0.0, 1.0, 0.0))
        s.add_collider(b, half_extents=(0.5, 0.5, 0.5), density=0.0)       # zero mass: MassProperties with inv_mass 0 — nothing can move it
        c = s.add_body(translation=(0.2, 2.2, 0.0))
        s.add_collider(c, half_extents=(0.5, 0.5, 0.5))
    elif kind == "one_iteration":
        s = S.box_stack(4)
        s.params["num_solver_iterations"] = 1
    else:
        raise ValueError(kind)
    return s


@pytest.mark.parametrize("gpu", GPU)
@pytest.mark.parametrize("kind", ["empty", "only_fixed", "one_falling_ball", "bare_bodies_and_a_fixed_collider", "dt0_stack", "no_gravity_at_rest",
                                  "far_from_the_origin", "massless", "one_iteration"])
def test_degenerate_worlds(gpu, kind):
    ws = _worlds(_degenerate(kind), gpu)
    for n in (1, 2, 30, 120):
        for w in ws:
            w.step(n)
        _same(ws, f"{kind}: {n} more steps")


class _Script:
    """the same edit script played on the oracle and on the device (handles = indices: rows are appended in insertion order on both)"""

    def __init__(self, scene, gpu):
        self.o = OracleWorld(scene)
        self.g = _device(scene) if gpu else None

    def step(self, n):
        self.o.step(n)
        if self.g is not None:
            self.g.step(n)

    def add_body(self, **kw):
        b = self.o.add_body(**kw)
        if self.g is not None:
            hb = self.g.insert_body(S.body_desc(**kw))
            assert int(hb) & 0xFFFFFFFF == b
        return b

    def add_collider(self, parent, **kw):
        c = self.o.add_collider(parent, **kw)
        if self.g is not None:
            hc = self.g.insert_collider(S.collider_desc(**kw), parent)
            assert int(hc) & 0xFFFFFFFF == c
        return c

    def remove_body(self, b):
        self.o.remove_body(b)
        if self.g is not None:
            self.g.remove_body([b])

    def remove_collider(self, c):
        self.o.remove_collider(c)
        if self.g is not None:
            self.g.remove_collider([c])

    def check(self, what, alive=None):
        op, ov = self.o.read()
        assert np.isfinite(op).all() and np.isfinite(ov).all(), what
        if self.g is not None:
            gp, gv = self.g.read_bodies()
            rows = slice(None) if alive is None else alive
            np.testing.assert_array_equal(gp[rows], op[rows], err_msg=what + " poses")
            np.testing.assert_array_equal(gv[rows], ov[rows], err_msg=what + " velocities")
            c = self.g.counters()
            assert c["overflow_flags"] == 0, c
        return op, ov


@pytest.mark.parametrize("gpu", GPU)
def test_a_world_that_starts_empty_and_is_edited_while_it_runs(gpu):
    """not a reference scene: every edit lands on a world in an unusual state — no body at all, bodies but no collider, a body that gains
    its first collider while it falls, the ground's collider removed under a resting body, the last dynamic body removed"""
    w = _Script(S.Scene(name="starts_empty", gravity=(0.0, -9.81, 0.0)), gpu)
    w.step(2)
    w.check("an empty world")
    ground = w.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
    w.step(1)
    gc = w.add_collider(ground, half_extents=(5.0, 0.5, 5.0))
    w.step(1)
    w.check("a fixed body and its collider")
    box = w.add_body(translation=(0.0, 2.0, 0.0), additional_mass=1.0)
    w.step(5)
    p, _ = w.check("a collider-less body falls")
    assert p[box, 1] < 2.0
    w.add_collider(box, half_extents=(0.5, 0.5, 0.5))
    w.step(60)
    p, v = w.check("... gains a collider and lands")
    assert 0.45 < p[box, 1] < 0.55 and abs(v[box, 1]) < 0.05, (p[box], v[box])
    ball = w.add_body(translation=(0.2, 3.0, 0.1))
    w.add_collider(ball, shape=S.SHAPE_BALL, half_extents=(0.3, 0, 0))
    w.step(40)
    w.check("a ball dropped onto the box")
    w.remove_collider(gc)
    w.step(20)
    p, _ = w.check("the ground's collider removed: everything falls again")
    assert p[box, 1] < 0.0
    w.remove_body(box)
    w.remove_body(ball)
    w.step(3)
    w.check("no dynamic body left", alive=[ground])
    again = w.add_body(translation=(1.0, 1.0, 1.0), linvel=(1.0, 0.0, 0.0))
    w.add_collider(again, half_extents=(0.2, 0.2, 0.2))
    w.step(10)
    w.check("and a new one inserted after that", alive=[ground, again])


@pytest.mark.parametrize("gpu", GPU)
@pytest.mark.parametrize("scene", ["stack", "joints"])
def test_integration_parameters_may_change_between_steps(gpu, scene):
    """PhysicsPipeline::step takes its IntegrationParameters per call (physics_pipeline/mod.rs): a variable time step, another
    iteration count or softer contacts from one step to the next are ordinary use.  rp_params_set on a live world = the next steps'
    parameters; the device must follow the oracle bit for bit through every change"""
    sc = S.box_stack(4) if scene == "stack" else S.jointed_pairs(2)
    o = OracleWorld(sc)
    g = _device(sc) if gpu else None
    base = sc.params.copy()
    changes = [dict(dt=1.0 / 120.0), dict(dt=1.0 / 30.0, num_solver_iterations=2), dict(dt=1.0 / 60.0, num_solver_iterations=6, num_internal_pgs_iterations=2),
               dict(contact_natural_frequency=15.0, normalized_prediction_distance=0.05), dict(dt=0.0), dict(warmstart_coefficient=0.5, num_internal_stabilization_iterations=2), dict()]
    for k, ch in enumerate(changes):
        p = base.copy()
        for name, val in ch.items():
            p[name] = val
        o.set_params(p)
        if g is not None:
            g.set_integration_parameters(p)
        for n in (1, 7):
            o.step(n)
            if g is not None:
                g.step(n)
            op, ov = o.read()
            assert np.isfinite(op).all() and np.isfinite(ov).all()
            if g is not None:
                gp, gv = g.read_bodies()
                np.testing.assert_array_equal(gp, op, err_msg=f"change {k} {ch}: poses after {n}")
                np.testing.assert_array_equal(gv, ov, err_msg=f"change {k} {ch}: velocities after {n}")


@pytest.mark.parametrize("gpu", GPU)
def test_b3d_large_world_at_reduced_size(gpu):
    """examples3d/b3d_large_world.rs (box3d's `large_world` benchmark) with a 40 x 40 floor instead of 1000 x 1000 (the oracle's
    sort-and-sweep broad phase takes seconds per pass over a million boxes): parentless fixed cuboids, a sphere dropped every 5 steps
    into the running world; every sphere comes to rest on the floor (y = 0.25 + 0.5) and falls asleep; oracle = device bit for bit"""
    grid, spheres = 40, 16
    w = _Script(S.large_world(grid), gpu)
    dropped, balls = 0, []
    for step in range(260):
        if dropped < spheres and step > 0 and step % 5 == 0:
            b = w.add_body(translation=S.large_world_drop(dropped, grid, spheres=spheres), can_sleep=1)
            w.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0, 0))
            balls.append(b); dropped += 1
        w.step(1)
        if step % 20 == 0 or step == 259:
            p, v = w.check(f"large_world step {step}")
    assert dropped == spheres
    np.testing.assert_allclose(p[balls, 1], 0.75, atol=0.01)
    assert np.abs(v[balls]).max() < 1e-3


def test_large_world_generator_matches_the_reference_formulas():
    """examples3d/b3d_large_world.rs:27-41, :55-63 in closed form: floor box (i, j) at x = -half_span + (i + 0.5) * cell (f32 arithmetic),
    i outer / j inner, no parent, half extents (5, 0.25, 5); sphere idx on a side x side grid (side = 10 for 100 spheres) over the
    inner 80 % of the floor at y = 1.5"""
    grid, cell = 30, np.float32(10.0)
    s = S.large_world(grid)
    assert len(s.bodies) == 0 and len(s.colliders) == grid * grid and set(s.collider_parents) == {-1}
    half_span = np.float32(0.5) * cell * np.float32(grid)
    for (i, j) in ((0, 0), (0, 1), (7, 19), (29, 29)):
        c = s.colliders[i * grid + j]
        want = (-half_span + (np.float32(i) + np.float32(0.5)) * cell, np.float32(0.0), -half_span + (np.float32(j) + np.float32(0.5)) * cell)
        np.testing.assert_array_equal(np.asarray(c["translation"], np.float32), np.asarray(want, np.float32))
        np.testing.assert_array_equal(np.asarray(c["half_extents"], np.float32), np.asarray((5.0, 0.25, 5.0), np.float32))
        assert int(c["shape"]) == S.SHAPE_CUBOID
    # spheres: side = 10; idx 0 -> cell (0, 0), idx 37 -> (7, 3)
    inset = np.float32(0.1) * np.float32(2.0) * half_span
    usable = np.float32(2.0) * half_span - np.float32(2.0) * inset
    for idx, (gi, gj) in ((0, (0, 0)), (37, (7, 3)), (99, (9, 9))):
        x = -half_span + inset + (np.float32(gi) + np.float32(0.5)) * (usable / np.float32(10))
        z = -half_span + inset + (np.float32(gj) + np.float32(0.5)) * (usable / np.float32(10))
        assert S.large_world_drop(idx, grid) == (float(x), 1.5, float(z))


def test_gpu_tools_compile():
    """the GPU-side tools of the round (they only run on the GPU box) at least parse"""
    import os
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("large_world.py", "shard_dryrun.py", "tile_diag.py", "pass_profile.py", "lp_steady.py", "bench_configs.py"):
        py_compile.compile(os.path.join(root, "tools", name), doraise=True)
