"""GPU parity tests proper (run with -m gpu on an MI355X): the HIP path through the C ABI against the
CPU oracle on identical inputs.

Tolerance: the HIP kernels evaluate the same f32 expressions in the same order as the oracle
(-ffp-contract=off on both sides, IEEE div/sqrt, identical colour order), so the stated bar is
BIT-EXACT equality of body poses and velocities; `ATOL` exists only so a future, deliberately
re-associated kernel has one place to state its tolerance (BASELINE.json allows 1e-4 relative).
"""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld

pytestmark = pytest.mark.gpu
ATOL = 0.0


def _compare(scene, checkpoints, atol=ATOL):
    g = PhysicsWorld.from_scene(scene)
    o = OracleWorld(scene)
    done = 0
    for cp in checkpoints:
        g.step(cp - done)
        o.step(cp - done)
        done = cp
        gp, gv = g.read_bodies()
        op, ov = o.read()
        assert np.isfinite(gp).all() and np.isfinite(gv).all()
        if atol == 0.0:
            np.testing.assert_array_equal(gp, op, err_msg=f"{scene.name} poses @ step {cp}")
            np.testing.assert_array_equal(gv, ov, err_msg=f"{scene.name} velocities @ step {cp}")
        else:
            np.testing.assert_allclose(gp, op, atol=atol, rtol=0)
            np.testing.assert_allclose(gv, ov, atol=atol * 60, rtol=0)
    c = g.counters()
    st = o.stats()
    assert c["overflow_flags"] == 0 and c["quarantined"] == 0
    assert c["num_manifolds"] == st["num_active_manifolds"]
    assert c["num_pairs"] == st["num_pairs"]
    return g, o


def test_box_stack_bit_exact():
    _compare(S.box_stack(3), [1, 2, 10, 60])


def test_pyramid10_bit_exact_300_steps():
    g, o = _compare(S.pyramid10(), [1, 10, 100, 300])
    gm, gn, gi = g.contacts()
    om, on, oi = o.manifolds()
    # same manifolds, colours, counts and impulses (order-insensitive: key by collider pair)
    gk = {(a, b): (c, n, tuple(i)) for (a, b, c, n), i in zip(gm.tolist(), gi.tolist())}
    ok = {(a, b): (c, n, tuple(i)) for (a, b, c, n), i in zip(om.tolist(), oi.tolist())}
    assert gk == ok


def test_many_pyramids_bit_exact():
    """BASELINE config C3 (b3d_many_pyramids, N=10,780, M=28,420)."""
    g, _ = _compare(S.many_pyramids(), [1, 5, 40])
    c = g.counters()
    assert c["num_manifolds"] == 28420 and c["num_dynamic_bodies"] == 10780


def test_many_pyramids_1000_steps_bit_exact():
    """The north-star accuracy check (positions after 1000 steps vs the CPU path; BASELINE.json allows
    1e-4 relative): full-size b3d_many_pyramids, checkpoints at 100 and 1000 steps, equality of every bit.
    The oracle runs its body-disjoint loops on the host cores (results do not depend on the thread count)."""
    import os
    import oracle_ffi
    oracle_ffi.set_threads(max(1, min(os.cpu_count() or 1, 16)))
    try:
        g, o = _compare(S.many_pyramids(), [100, 1000])
    finally:
        oracle_ffi.set_threads(1)
    c = g.counters()
    assert c["fast_steps"] > 800 and c["replayed_steps"] == 0, c


def test_large_pyramid_bit_exact():
    """BASELINE config C2 (single island) at base 60 for oracle speed (1,830 cubes)."""
    _compare(S.large_pyramid(60), [1, 10, 30])


def _oracle_threads():
    import os
    import oracle_ffi
    oracle_ffi.set_threads(max(1, min(os.cpu_count() or 1, 16)))
    return oracle_ffi


def test_large_pyramid_full_size_bit_exact():
    """BASELINE config C2 at its stated size: b3d_large_pyramid base 200 = 20,100 cuboids in ONE island (~59,900 manifolds, the
    global path with every colour a parallel stage), 1 / 3 / 10 steps, every bit (the first steps, before the tiling stands; the
    1000-step run of the same world on LDS tiles + lean graphs is tests/test_gpu_fullsize.py)."""
    ffi = _oracle_threads()
    try:
        g, _ = _compare(S.large_pyramid(200), [1, 3, 10])
    finally:
        ffi.set_threads(1)
    c = g.counters()
    assert c["num_dynamic_bodies"] == 20100 and c["num_manifolds"] > 59000, c


def test_many_pyramids_c4_single_gpu_bit_exact():
    """BASELINE config C4 (b3d_many_pyramids scaled to 54 x 54 = 2,916 pyramids = 160,380 cuboids) on ONE GPU: 1 and 5 steps (120 steps
    with fast steps, replays and full steps: tests/test_gpu_fullsize.py)."""
    ffi = _oracle_threads()
    try:
        g, _ = _compare(S.many_pyramids(54, 54), [1, 5])
    finally:
        ffi.set_threads(1)
    c = g.counters()
    assert c["num_dynamic_bodies"] == 160380 and c["num_manifolds"] == 2916 * 145, c


def test_joint_grid_full_size_60_steps_bit_exact():
    """BASELINE config C5 at its stated size (100 x 100 balls, 19,800 spherical joints) for 60 steps, joint colours and impulses
    included (1000 steps: tests/test_gpu_fullsize.py)."""
    ffi = _oracle_threads()
    try:
        g, o = _compare(S.joint_grid(100), [1, 20, 60])
    finally:
        ffi.set_threads(1)
    _compare_joints(g, o)


def _world_with_env(scene, **env):
    """PhysicsWorld built while the given environment switches are set (the library reads them in rp_world_create)."""
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return PhysicsWorld.from_scene(scene)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("scene", ["large_pyramid60", "joint_grid40", "tumble_coulomb", "joint_chain_boxes"])
def test_dataflow_launch_equals_per_stage_launches(scene):
    """The global path as ONE dataflow launch (rp_flow.hip: per-body ticket hand-offs) against the same path as one launch per
    colour stage (RP_NO_FLOW=1): identical bits, whatever ran ahead of what."""
    make = {"large_pyramid60": lambda: S.large_pyramid(60), "joint_grid40": lambda: S.joint_grid(40),
            "tumble_coulomb": lambda: _with_param(S.tumble(64, seed=7), "friction_model", S.FRICTION_COULOMB),
            "joint_chain_boxes": lambda: S.joint_chain(6, with_boxes=True)}[scene]
    a = _world_with_env(make(), RP_FORCE_MULTI=1, RP_FLOW=1)     # RP_FLOW=1: the dataflow launch whatever the world holds
    b = _world_with_env(make(), RP_FORCE_MULTI=1, RP_NO_FLOW=1)  # per-stage launches (+ the body-centric warm start under the twist model)
    done = 0
    for cp in (1, 7, 40, 120):
        a.step(cp - done); b.step(cp - done); done = cp
        (ap, av), (bp, bv) = a.read_bodies(), b.read_bodies()
        np.testing.assert_array_equal(ap, bp, err_msg=f"{scene} poses @ {cp}")
        np.testing.assert_array_equal(av, bv, err_msg=f"{scene} velocities @ {cp}")
    assert a.counters()["overflow_flags"] == 0


def _with_param(scene, key, value):
    scene.params[key] = value
    return scene


@pytest.mark.parametrize("npgs,nstab", [(0, 1), (1, 0), (0, 0), (2, 2)])
def test_jointed_world_with_unusual_inner_iteration_counts(npgs, nstab):
    """num_internal_pgs_iterations = 0 leaves no joint event between a substep's joint-row update and its integrate on a body's hand-off
    chain, so such jointed worlds keep to the per-stage launches (rp_api.hip: flow_now); every combination matches the oracle, on the
    default launch and with the global path forced onto several launches"""
    def make():
        sc = S.joint_chain(6, with_boxes=True)
        sc.params["num_internal_pgs_iterations"] = npgs
        sc.params["num_internal_stabilization_iterations"] = nstab
        return sc
    _compare(make(), [1, 7, 60])
    a = _world_with_env(make(), RP_FORCE_MULTI=1)
    o = OracleWorld(make())
    a.step(60); o.step(60)
    (ap, av), (op, ov) = a.read_bodies(), o.read()
    np.testing.assert_array_equal(ap, op); np.testing.assert_array_equal(av, ov)
    sc2 = S.joint_grid(24)
    sc2.params["num_internal_pgs_iterations"] = npgs
    sc2.params["num_internal_stabilization_iterations"] = nstab
    _compare(sc2, [1, 20])


def test_tumble_dynamic_scene_bit_exact():
    """Rotated cuboids + balls with velocities: full updates, edge/edge SAT, reduction, pair
    deletion, recolouring, restitution, damping."""
    _compare(S.tumble(64, seed=7), [1, 5, 30, 120, 240])


def test_tumble_cuboids_only_second_seed():
    _compare(S.tumble(40, seed=21, balls=False), [20, 150])


def test_crowded_grid_cell_goes_through_the_large_list():
    """48 small balls dropped as one tight cluster among unit cubes: the broad phase's grid cell is as wide as the cubes, so the
    cluster's fat AABBs fill one 32-slot hash bucket and the colliders that find it full are paired through the brute-force list
    instead (rp_broadphase.hip, bp_build) — same pair set (update.rs:35-602: every pair of intersecting fat AABBs), same bits"""
    s = S.Scene(name="crowded_cell", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(12.0, 0.5, 12.0))
    for ix in range(8):
        for iz in range(8):
            b = s.add_body(translation=(1.6 * (ix - 3.5), 0.5, 1.6 * (iz - 3.5)))
            s.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    for k in range(48):
        ix, iy, iz = k % 4, (k // 4) % 4, k // 16
        b = s.add_body(translation=(0.8 + 0.11 * ix, 1.5 + 0.11 * iy, 0.8 + 0.11 * iz), linvel=(0.0, -1.0, 0.0))
        s.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.05, 0, 0), friction=0.3)
    gw, _ = _compare(s, [1, 5, 40, 150])
    gw2 = PhysicsWorld.from_scene(s); gw2.step(1)
    assert gw2.counters()["bp_large_list"] > 1, gw2.counters()  # the ground slab + the balls that met the full bucket


@pytest.mark.parametrize("coeff", [1.0, 0.5, 0.0])
def test_resting_impulse_kat_on_gpu(coeff):
    """total_contact_impulse.rs:13-75 through the C ABI."""
    for cuboid in (True, False):
        sc = S.Scene(name="kat1", gravity=(0.0, -9.81, 0.0))
        sc.params["warmstart_coefficient"] = coeff
        sc.add_collider(-1, half_extents=(10.0, 0.5, 10.0), translation=(0.0, -0.5, 0.0))
        b = sc.add_body(translation=(0.0, 0.5, 0.0), additional_mass=1.0)
        if cuboid:
            sc.add_collider(b, half_extents=(0.5, 0.5, 0.5), density=0.0)
        else:
            sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0, 0), density=0.0)
        w = PhysicsWorld.from_scene(sc)
        w.step(300)
        expected = 9.81 / 60.0
        assert abs(w.total_contact_impulse() - expected) <= expected * 1e-2


def test_ball_rests_on_floor_on_gpu():
    sc = S.Scene(name="kat2", gravity=(0.0, -9.81, 0.0))
    sc.add_collider(-1, half_extents=(10.0, 0.5, 10.0))
    b = sc.add_body(translation=(0.0, 4.0, 0.0))
    sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0, 0))
    w = PhysicsWorld.from_scene(sc)
    w.step(200)
    assert abs(w.read_bodies()[0][b, 1] - 1.0) < 0.02


def _compare_joints(g, o):
    gc, gi = g.read_joints()
    oc, oi = o.read_joints()
    np.testing.assert_array_equal(gc, oc, err_msg="joint colours")
    np.testing.assert_array_equal(gi, oi, err_msg="joint impulses")


def test_joint_chain_bit_exact():
    """Spherical joints with large motion: rows rebuilt every substep, joints before contacts."""
    g, o = _compare(S.joint_chain(8), [1, 10, 100, 300])
    _compare_joints(g, o)


def test_joint_chain_with_contacts_bit_exact():
    """Jointed boxes falling on a slab: joints and contacts on the same bodies, joint colours avoid the
    contact colours (recolouring when contacts appear)."""
    g, o = _compare(S.joint_chain(6, with_boxes=True), [1, 30, 120, 240])
    _compare_joints(g, o)


def test_joint_grid_bit_exact():
    """BASELINE config C5 (b3d_joint_grid) at 24 x 24 for oracle speed: parallel joint colours + serial overflow."""
    g, o = _compare(S.joint_grid(24), [1, 10, 60, 150])
    _compare_joints(g, o)


def test_joint_net_stays_bounded_on_gpu():
    """joint_stability.rs:105-175 through the C ABI (reduced horizon): positions bounded, no runaway velocity."""
    sc = S.joint_net(32)
    w = PhysicsWorld.from_scene(sc)
    w.step(1000)
    pos, vel = w.read_bodies()
    assert np.isfinite(pos).all()
    assert np.linalg.norm(pos[:, :3], axis=1).max() < 500.0
    assert np.linalg.norm(vel[:, :3], axis=1).max() < 100.0


def test_joint_grid_full_size():
    """b3d_joint_grid at full size (100 x 100, 19,800 joints): 5 steps bit-exact, then properties."""
    sc = S.joint_grid(100)
    g, o = _compare(sc, [1, 5])
    _compare_joints(g, o)
    g.step(200)
    pos, vel = g.read_bodies()
    assert np.isfinite(pos).all() and np.abs(vel).max() < 100.0


def test_fast_path_abort_and_replay_bit_exact():
    """A settled stack runs on the steady-state fast graph; a velocity kick makes k_fast_front give up
    (recycle tests fail, fat AABBs are left) and the host replays those steps on the full graph.  The
    result must not depend on which graph ran."""
    sc = S.many_pyramids(rows=1, cols=2)
    g = PhysicsWorld.from_scene(sc)
    o = OracleWorld(sc)
    g.step(150); o.step(150)
    c0 = g.counters()
    assert c0["fast_steps"] > 50, c0
    top = len(sc.bodies) - 1
    kick = np.array([[3.0, 2.0, 0.5, 0.0, 1.0, 0.0]], np.float32)
    g.write_bodies([top], vel6=kick)
    o.set_vel(top, kick[0, :3], kick[0, 3:])
    for cp in (1, 20, 150):
        g.step(cp); o.step(cp)
        gp, gv = g.read_bodies()
        op, ov = o.read()
        np.testing.assert_array_equal(gp, op)
        np.testing.assert_array_equal(gv, ov)
    c1 = g.counters()
    assert c1["full_steps"] > c0["full_steps"]


def test_fused_step_with_more_islands_than_workgroups():
    """16 x 16 pyramids = 256 islands > the 240 co-resident workgroups of the fused fast step: some
    workgroups own two islands (the second one is validated in the prologue, before the arrival)."""
    import os
    import oracle_ffi
    sc = S.many_pyramids(rows=16, cols=16)
    oracle_ffi.set_threads(max(1, min(os.cpu_count() or 1, 16)))
    try:
        g, _ = _compare(sc, [30, 160])
    finally:
        oracle_ffi.set_threads(1)
    c = g.counters()
    assert c["fast_steps"] > 60 and c["replayed_steps"] == 0, c


def test_fused_and_unfused_fast_paths_agree(monkeypatch):
    sc = S.many_pyramids(rows=2, cols=3)
    a = PhysicsWorld.from_scene(sc)
    a.step(150)
    pa, va = a.read_bodies()
    monkeypatch.setenv("RP_NO_FUSED", "1")
    b = PhysicsWorld.from_scene(sc)
    b.step(150)
    pb, vb = b.read_bodies()
    assert a.counters()["fast_steps"] > 50 and b.counters()["fast_steps"] > 50
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(va, vb)


def test_fast_and_full_graphs_agree(monkeypatch):
    sc = S.many_pyramids(rows=2, cols=2)
    a = PhysicsWorld.from_scene(sc)
    a.step(200)
    pa, va = a.read_bodies()
    assert a.counters()["fast_steps"] > 100
    monkeypatch.setenv("RP_NO_FAST", "1")
    b = PhysicsWorld.from_scene(sc)
    b.step(200)
    pb, vb = b.read_bodies()
    assert b.counters()["fast_steps"] == 0
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(va, vb)


@pytest.mark.parametrize("override", [
    {"warmstart_joints": 1},
    {"friction_in_bias_pass": 1},
    {"num_internal_pgs_iterations": 2, "num_internal_stabilization_iterations": 2},
    {"num_solver_iterations": 2, "warmstart_coefficient": 0.5},
    {"num_internal_stabilization_iterations": 0},
])
def test_integration_parameter_variants_bit_exact(override):
    """Non-default IntegrationParameters (integration_parameters.rs:181-304) through both solver paths:
    LDS islands (pyramid) and the global path with joints (jointed boxes on a slab)."""
    for mk in (lambda: S.pyramid10(), lambda: S.joint_chain(5, with_boxes=True)):
        sc = mk()
        for k, v in override.items():
            sc.params[k] = v
        _compare(sc, [1, 20, 90])


def _same_state(g, o, msg):
    gp, gv = g.read_bodies()
    op, ov = o.read()
    np.testing.assert_array_equal(gp, op, err_msg=msg)
    np.testing.assert_array_equal(gv, ov, err_msg=msg)


def test_remove_body_mid_simulation_bit_exact():
    """RigidBodySet::remove in a settled stack: its pairs are deleted by the next broad-phase pass, every
    other pair keeps its warm-start data, the boxes above fall."""
    sc = S.box_stack(5)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(80); o.step(80)
    victim = 3  # body 0 = slab, 1..5 = boxes
    g.remove_body(victim); o.remove_body(victim)
    for n in (1, 30, 120):
        g.step(n); o.step(n)
        _same_state(g, o, f"after removing body {victim}, +{n}")
    with pytest.raises(Exception):
        g.remove_body(victim)  # stale handle
    pos, _ = g.read_bodies()
    assert pos[4, 1] < 3.0  # the box that sat on the victim came down


def test_remove_joint_and_collider_bit_exact():
    sc = S.joint_chain(8)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(40); o.step(40)
    g.remove_impulse_joint(3); o.remove_joint(3)  # the chain splits
    g.step(60); o.step(60)
    _same_state(g, o, "after removing joint 3")
    gc, gi = g.read_joints()
    oc, oi = o.read_joints()
    keep = np.arange(len(gc)) != 3
    np.testing.assert_array_equal(gi[keep], oi[keep])
    sc2 = S.pyramid10()
    g2, o2 = PhysicsWorld.from_scene(sc2), OracleWorld(sc2)
    g2.step(50); o2.step(50)
    g2.remove_collider(30); o2.remove_collider(30)  # a cube in the pyramid loses its shape: neighbours fall through it
    for n in (1, 60):
        g2.step(n); o2.step(n)
        _same_state(g2, o2, f"after removing collider 30, +{n}")


def test_insert_into_live_world_bit_exact():
    """RigidBodySet::insert / ColliderSet::insert_with_parent into a stepped world: the rows are appended
    in place, every existing pair keeps its warm-start data, so the result matches the oracle bit for bit."""
    sc = S.box_stack(3)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(60); o.step(60)
    body = S.body_desc(translation=(0.1, 6.0, 0.05), linvel=(0.0, -1.0, 0.0))
    col = S.collider_desc(half_extents=(0.4, 0.4, 0.4), density=2.0)
    hb = g.insert_body(body)
    g.insert_collider(col, hb)
    from oracle_ffi import lib
    ob = lib().ro_add_body(o._w, np.array([body], S.BODY_DTYPE).ctypes.data)
    lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
    o.n += 1
    assert int(hb) & 0xFFFFFFFF == ob
    for n in (1, 40, 200):
        g.step(n); o.step(n)
        _same_state(g, o, f"after inserting a box, +{n}")
    pos, _ = g.read_bodies()
    assert pos[ob, 1] < 5.0 and g.quarantined().size == 0


def test_insert_beyond_capacity_rebuilds_from_current_state():
    """More inserts than the spare device rows: the device world is rebuilt from the CURRENT body states."""
    sc = S.box_stack(2)
    g = PhysicsWorld.from_scene(sc)
    g.step(60)
    before, _ = g.read_bodies()
    n_new = 400
    descs = np.array([S.body_desc(translation=(-8.0 + 1.6 * (i % 10) + (2.0 if i % 10 >= 5 else 0.0), 0.6 + 1.2 * (i // 100), -8.0 + 1.6 * ((i // 10) % 10)))
                      for i in range(n_new)], S.BODY_DTYPE)  # 4 layers of 10 x 10 boxes on the slab, clear of the stack at the origin
    handles = g.insert_bodies(descs)
    g.insert_colliders(np.array([S.collider_desc() for _ in range(n_new)], S.COLLIDER_DTYPE), handles)
    after, _ = g.read_bodies()
    np.testing.assert_array_equal(after[:3], before[:3])
    g.step(240)
    pos, vel = g.read_bodies()
    assert np.isfinite(pos).all() and np.abs(vel).max() < 1.0 and pos[:, 1].min() > -1.0


# ---- sleeping: timers, whole-island sleep, island-wide wake-up — rp_sleep.hip vs the oracle ----
def _same_sleep_state(g, o, msg):
    _same_state(g, o, msg)
    np.testing.assert_array_equal(g.sleeping(), o.sleeping(), err_msg=msg + " (sleeping flags)")


def test_sleeping_stack_kick_and_wake_up_bit_exact():
    sc = S.box_stack(3).enable_sleep()
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    done = 0
    for cp in (20, 33, 36, 40, 60, 110):
        g.step(cp - done); o.step(cp - done); done = cp
        _same_sleep_state(g, o, f"box_stack sleep @ {cp}")
    assert g.sleeping()[1:].all() and g.counters()["num_manifolds"] == 0 and g.counters()["num_sleeping_bodies"] == 3
    kick = np.array([[1.0, 0.0, 0.0, 0.0, 0.0, 0.0]], np.float32)
    g.write_bodies([3], vel6=kick); o.set_vel(3, kick[0, :3], kick[0, 3:])   # set_linvel(.., wake_up = true)
    for n in (1, 1, 30, 150):
        g.step(n); o.step(n)
        _same_sleep_state(g, o, f"box_stack after the kick, +{n}")
    assert g.sleeping()[1:].all()
    g.wake_up([2]); o.wake_up(2)            # IslandManager::wake_up on the middle box wakes the island
    g.step(1); o.step(1)
    _same_sleep_state(g, o, "box_stack after wake_up")
    assert not g.sleeping().any()
    g.step(60); o.step(60)
    _same_sleep_state(g, o, "box_stack asleep again")
    assert g.sleeping()[1:].all()


def test_sleep_impact_wakes_one_island_bit_exact():
    """A cube dropped on a sleeping stack: begin-touch wakes the struck island only (contacts.rs:333-351); the woken
    island's other pairs re-enter the solver one step later (their hints were count-cleared, solver_graph.rs:21-49)."""
    sc = S.sleep_impact()
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    seen_partial = False
    for cp in range(5, 261, 5):
        g.step(5); o.step(5)
        _same_sleep_state(g, o, f"sleep_impact @ {cp}")
        sl = g.sleeping()
        seen_partial |= (not sl[1:4].any()) and sl[4:7].all()
    assert seen_partial and g.sleeping()[1:].all()


def test_sleep_many_islands_bit_exact():
    sc = S.many_pyramids(rows=2, cols=2).enable_sleep()
    g, o = _compare(sc, [10, 30, 35, 40, 60])
    np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    assert g.sleeping()[1:].all() and g.counters()["num_sleeping_bodies"] == 220
    before = g.counters()["fast_steps"]
    g.step(200); o.step(200)                                  # a fully sleeping world retires idle steps (one tiny kernel each)
    _same_sleep_state(g, o, "sleeping world, 200 idle steps")
    assert g.counters()["fast_steps"] - before >= 150
    g.wake_up([7]); o.wake_up(7)                              # ... until something wakes up
    g.step(3); o.step(3)
    _same_sleep_state(g, o, "sleeping world woken")
    assert g.sleeping().sum() == 220 - 55
    sc2 = S.tumble(40, seed=11).enable_sleep()
    g2, o2 = PhysicsWorld.from_scene(sc2), OracleWorld(sc2)
    for cp in range(50, 601, 50):
        g2.step(50); o2.step(50)
        _same_sleep_state(g2, o2, f"tumble sleep @ {cp}")


def test_sleep_user_changes_wake_partners_bit_exact():
    """Collider removal wakes every body that had a pair with it (pair_management.rs:88-99); a teleported body wakes
    its contact partners (:236-258)."""
    sc = S.box_stack(4).enable_sleep()
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(70); o.step(70)
    assert g.sleeping()[1:].all()
    g.remove_body(1); o.remove_body(1)
    for n in (1, 40, 120):
        g.step(n); o.step(n)
        _same_sleep_state(g, o, f"sleeping stack, bottom box removed, +{n}")
    assert g.sleeping()[2:].all()
    pose = np.array([[3.0, 0.5, 0.0, 0.0, 0.0, 0.0, 1.0]], np.float32)
    g.write_bodies([2], pos7=pose); o.set_pose(2, pose[0])   # the (new) bottom box is teleported away
    for n in (1, 40, 120):
        g.step(n); o.step(n)
        _same_sleep_state(g, o, f"sleeping stack, bottom box teleported, +{n}")
    pos, _ = g.read_bodies()
    assert pos[3, 1] < 0.6 and g.sleeping()[2:].all()


def test_sleep_full_size_many_pyramids():
    """b3d_many_pyramids with the builder's default can_sleep(true): all 196 islands fall asleep on their own."""
    sc = S.many_pyramids().enable_sleep()
    g, o = _compare(sc, [35, 60])
    np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    g.step(60)
    c = g.counters()
    assert c["num_sleeping_bodies"] == 10780 and c["num_manifolds"] == 0


def test_user_forces_and_impulses_bit_exact():
    """RigidBody::{add_force, add_torque, reset_forces, apply_impulse, apply_torque_impulse} (rigid_body.rs:1145-1343)."""
    sc = S.box_stack(3).enable_sleep()
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(60); o.step(60)
    assert g.sleeping()[1:].all()
    g.apply_impulse([3], impulse=(0.4, 2.5, 0.1), torque_impulse=(0.0, 0.3, 0.1)); o.apply_impulse(3, (0.4, 2.5, 0.1), (0.0, 0.3, 0.1))   # wakes the stack
    for n in (1, 10, 40):
        g.step(n); o.step(n)
        _same_sleep_state(g, o, f"after the impulse, +{n}")
    g.add_force([2], force=(0.0, 25.0, 3.0), torque=(0.0, 0.0, 1.0)); o.add_force(2, (0.0, 25.0, 3.0), (0.0, 0.0, 1.0))   # persistent thrust > weight
    for n in (1, 20):
        g.step(n); o.step(n)
        _same_sleep_state(g, o, f"under thrust, +{n}")
    pos, vel = g.read_bodies()
    assert vel[2, 1] > 1.0                                      # the middle box lifts the top one
    g.add_force([2], reset=True); o.add_force(2, reset=True)    # reset_forces + reset_torques
    for n in (1, 60, 120):
        g.step(n); o.step(n)
        _same_sleep_state(g, o, f"thrust removed, +{n}")
    assert g.sleeping()[1:].all()


def test_locked_axes_bit_exact():
    g, o = _compare(S.locked_axes_scene(), [1, 2, 10, 60, 240])
    pos, _ = g.read_bodies()
    np.testing.assert_array_equal(pos[4, :3], np.array(S.locked_axes_scene().bodies[4]["translation"]))   # fully locked body


def test_compound_bodies_bit_exact():
    """Several colliders per body at arbitrary pos_wrt_parent: summed MassProperties (offset centre of mass, principal
    frame from the diagonalised tensor), pairs sharing a body, gyroscopic term in a non-trivial principal frame."""
    g, o = _compare(S.compound_bodies(12), [1, 2, 10, 60, 200, 400])
    pos, vel = g.read_bodies()
    assert pos[1:, 1].min() > 0.0 and np.abs(vel).max() < 5.0
    # attaching a second collider to a live body (in place) moves its centre of mass
    sc = S.box_stack(1)
    g2, o2 = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g2.step(30); o2.step(30)
    extra = S.collider_desc(half_extents=(0.2, 0.2, 0.2), translation=(0.6, 0.4, 0.0), density=6.0)
    g2.insert_collider(extra, 1)
    from oracle_ffi import lib
    lib().ro_add_collider(o2._w, np.array([extra], S.COLLIDER_DTYPE).ctypes.data, 1)
    for n in (1, 30, 120):
        g2.step(n); o2.step(n)
        _same_state(g2, o2, f"collider attached to a live body, +{n}")


def test_joints_with_sleeping_and_kinematic_bodies_bit_exact():
    """Joints link sleep islands, a sleeping joint leaves the solver selection, a kick on one body wakes its jointed partner;
    a joint may hang from a kinematic body (impulse_joint_set.rs:504-572)."""
    sc = S.jointed_pairs(3).enable_sleep()
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    done = 0
    for cp in (30, 60, 90, 120, 200):
        g.step(cp - done); o.step(cp - done); done = cp
        _same_sleep_state(g, o, f"jointed pairs with sleep @ {cp}")
    sl = g.sleeping()
    assert sl[1:10].all()                                       # the stack and the three revolute pairs sleep (pairs as joint-linked islands)
    kick = np.array([[0.0, 2.0, 0.5, 0.0, 0.0, 0.0]], np.float32)
    g.write_bodies([5], vel6=kick); o.set_vel(5, kick[0, :3], kick[0, 3:])   # body 5 = second cube of the first pair
    for n in (1, 1, 20, 150):
        g.step(n); o.step(n)
        _same_sleep_state(g, o, f"jointed pair kicked, +{n}")
    gc, gi = g.read_joints(); oc, oi = o.read_joints()
    np.testing.assert_array_equal(gc, oc); np.testing.assert_array_equal(gi, oi)
    g2, o2 = _compare(S.kinematic_crane(5), [1, 10, 60, 200, 400])
    np.testing.assert_array_equal(g2.sleeping(), o2.sleeping())
    pos, _ = g2.read_bodies()
    assert pos[1, 0] == pytest.approx(-3.0 + 400 / 60.0, abs=1e-3)                          # the trolley follows its velocity
    assert np.linalg.norm(pos[2, :3] - pos[1, :3]) == pytest.approx(0.7, abs=2e-2)           # and drags the chain along


def test_joint_limits_bit_exact():
    g, o = _compare(S.limited_joints(), [1, 2, 10, 40, 120, 300])
    gc, gi = g.read_joints(); oc, oi = o.read_joints()
    np.testing.assert_array_equal(gc, oc); np.testing.assert_array_equal(gi, oi)
    pos, _ = g.read_bodies()
    assert pos[2, 1] == pytest.approx(3.5, abs=5e-3)           # the slider rests on its lower stop


def test_joint_motors_bit_exact():
    """Motor rows (motor_angular / motor_linear, both motor models, force caps, a motorised axis that is also limited, three
    motors on one joint, motors between dynamic bodies) built and solved before the lock rows; then a motor changed at run time
    through rp_impulse_joints_set_motor, and the same scene with warm-started joints (JointMotor::impulse seeds)."""
    g, o = _compare(S.motorised_joints(), [1, 2, 10, 40, 120, 300])
    np.testing.assert_array_equal(g.joint_motor_impulses(), o.joint_motor_impulses())
    _, vel = g.read_bodies()
    assert vel[2, 5] == pytest.approx(3.0, abs=1e-3)            # the wheel reached its target speed
    g.set_joint_motor(0, 3, target_vel=-2.0, damping=5.0); o.set_joint_motor(0, 3, target_vel=-2.0, damping=5.0)
    g.set_joint_motor(1, 0, target_pos=0.2, stiffness=400.0, damping=40.0, max_force=60.0, model=S.MOTOR_FORCE_BASED)
    o.set_joint_motor(1, 0, target_pos=0.2, stiffness=400.0, damping=40.0, max_force=60.0, model=S.MOTOR_FORCE_BASED)
    for n in (1, 9, 90):
        g.step(n); o.step(n)
        _same_state(g, o, f"motors retargeted, +{n}")
    np.testing.assert_array_equal(g.joint_motor_impulses(), o.joint_motor_impulses())
    sc = S.motorised_joints()
    sc.params["warmstart_joints"] = 1
    g, o = _compare(sc, [1, 2, 10, 60, 150])
    np.testing.assert_array_equal(g.joint_motor_impulses(), o.joint_motor_impulses())


def test_joint_motor_wakes_sleeping_bodies_bit_exact():
    """issue_692 on the device: a motor set on a sleeping jointed pair wakes it and drives it."""
    sc = S.Scene(name="motor_wake", gravity=(0.0, -9.81, 0.0))
    kin = sc.add_body(body_type=S.BODY_KINEMATIC_POSITION, can_sleep=1)
    dyn = sc.add_body(translation=(0.0, -2.0, 0.0), can_sleep=1)
    sc.add_collider(dyn, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0))
    sc.add_joint(kin, dyn, (0.0, 0.0, 0.0), (0.0, 2.0, 0.0), locked_axes=S.LOCK_REVOLUTE)
    g, o = _compare(sc, [1, 10, 150])
    assert g.sleeping()[dyn] and o.sleeping()[dyn]
    g.set_joint_motor(0, 3, target_vel=2.0, damping=100.0); o.set_joint_motor(0, 3, target_vel=2.0, damping=100.0)
    for n in (1, 5, 50):
        g.step(n); o.step(n)
        _same_state(g, o, f"motor set on a sleeping pair, +{n}")
        np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    assert not g.sleeping()[dyn] and np.linalg.norm(g.read_bodies()[1][dyn, 3:]) > 0.1


def test_reference_sleep_wake_scenes_bit_exact():
    """Scenes of the reference's sleep_wake.rs / joint_assembly_persistence.rs / joint_stability.rs (restated against the oracle
    in tests/test_reference_kats.py) through the device path: a toppling domino wave running through sleeping dominoes, a
    sub-gate impact by a body inserted into the stepped world, a fixed joint anchor moved while its bob sleeps, hanging
    chains on limited prismatic joints."""
    import test_reference_kats as K
    sc = K._sw_world()
    for i in range(10):
        b = sc.add_body(translation=(i * 0.4, 2.0, 0.0), rotation=K.quat_from_scaled_axis((0.0, 0.0, -0.2)) if i == 0 else (0, 0, 0, 1), can_sleep=1)
        sc.add_collider(b, half_extents=(0.1, 2.0, 1.0))
    g, o = _compare(sc, [1, 30, 120, 400, 900, 1500])
    np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    assert g.sleeping()[1:].all()

    sc = K._sw_world()
    g0 = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 70.0))
    sc.add_collider(g0, half_extents=(20.0, 0.5, 10.0), friction=0.0)
    target = sc.add_body(translation=(0.0, 0.5, 70.0), can_sleep=1)
    sc.add_collider(target, half_extents=(0.5, 0.5, 0.5), friction=0.0)
    g, o = _compare(sc, [1, 120])
    assert g.sleeping()[target]
    body = S.body_desc(translation=(-3.0, 0.5, 70.0), linvel=(0.6, 0.0, 0.0), can_sleep=1)
    col = S.collider_desc(half_extents=(0.5, 0.5, 0.5), friction=0.0)
    hb = g.insert_body(body); g.insert_collider(col, hb)
    ob = o.add_body(translation=(-3.0, 0.5, 70.0), linvel=(0.6, 0.0, 0.0), can_sleep=1); o.add_collider(ob, half_extents=(0.5, 0.5, 0.5), friction=0.0)
    for n in (1, 199, 40):
        g.step(n); o.step(n)
        _same_state(g, o, f"sub-gate impact, +{n}")
        np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    assert not g.sleeping()[target] and g.read_bodies()[1][target, 0] > 0.25

    sc = S.Scene(name="moved_anchor", gravity=(0.0, -9.81, 0.0))
    anchor = sc.add_body(body_type=S.BODY_FIXED)
    bob = sc.add_body(translation=(0.0, -2.0, 0.0), can_sleep=1)
    sc.add_collider(bob, shape=S.SHAPE_BALL, half_extents=(0.2, 0.0, 0.0))
    sc.add_joint(anchor, bob, (0.0, 0.0, 0.0), (0.0, 2.0, 0.0), locked_axes=S.LOCK_LIN)
    g, o = _compare(sc, [1, 40])
    assert g.sleeping()[bob]
    new_pose = [5.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]
    g.write_bodies([anchor], pos7=[new_pose]); o.set_pose(anchor, new_pose)
    for n in (1, 9, 290):
        g.step(n); o.step(n)
        _same_state(g, o, f"fixed anchor moved, +{n}")
    assert abs(np.linalg.norm(g.read_bodies()[0][bob, :3] - np.array([5.0, 0.0, 0.0])) - 2.0) < 0.3

    sc = S.Scene(name="prismatic_chains", gravity=(0.0, -9.81, 0.0))
    for chain in range(10):
        parent = sc.add_body(body_type=S.BODY_FIXED, translation=(chain * 4.0, 0.0, 0.0))
        sc.add_collider(parent, half_extents=(0.4, 0.4, 0.4))
        for i in range(10):
            child = sc.add_body(translation=(chain * 4.0, -(i + 1) * 1.0, 0.0))
            sc.add_collider(child, half_extents=(0.4, 0.4, 0.4))
            basis = K._axis_basis_z(np.pi / 4 if i % 2 == 0 else 3 * np.pi / 4)
            sc.add_joint(parent, child, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), locked_axes=S.LOCK_PRISMATIC, basis1=basis, basis2=basis, limits={0: (-1.5, 1.5)})
            parent = child
    _compare(sc, [1, 10, 100, 1000])


def test_capsules_bit_exact():
    """Capsule colliders (ColliderBuilder::capsule_x/y/z): capsule-capsule (segment-segment closest points), cuboid-capsule in
    both collider orders (SAT against the segment, support face clipped against it), capsule-ball in both orders, capsule mass
    properties with their principal frame, a compound dumbbell; with sleeping too."""
    g, o = _compare(S.capsules(6), [1, 2, 10, 40, 120, 300, 600])
    gm, _, gi = g.contacts(); om, _, oi = o.manifolds()
    assert len(gm) == len(om)
    _compare(S.capsules(4).enable_sleep(), [1, 30, 200, 500])


def test_insert_joint_into_live_world_bit_exact():
    """ImpulseJointSet::insert into a stepped world (the joint arrays have no spare rows: the device world moves to larger ones
    and carries every row over): contacts keep their warm-start data and colours, the existing joints their colours and
    impulses, so the result matches the oracle bit for bit; also after a removal (tombstoned device index) and with
    warm-started joints."""
    for ws in (0, 1):
        sc = S.joint_chain(6, with_boxes=True)
        sc.params["warmstart_joints"] = ws
        g, o = _compare(sc, [1, 30, 150])
        n = len(sc.bodies)
        a, b = n - 1, n - 2
        jd = np.zeros((), S.JOINT_DTYPE)
        jd["body1"], jd["body2"] = a, b
        jd["local_anchor1"], jd["local_anchor2"] = (0.0, 0.6, 0.0), (0.0, -0.6, 0.0)
        jd["local_basis1"] = jd["local_basis2"] = (0, 0, 0, 1)
        jd["locked_axes"], jd["contacts_enabled"] = S.LOCK_LIN, 1
        for k in range(6):
            jd["motors"][k] = S.motor_desc()
        from oracle_ffi import lib
        hj = g.insert_impulse_joint(a, b, jd)
        oj = lib().ro_add_joint(o._w, np.array([jd], S.JOINT_DTYPE).ctypes.data)
        assert hj == oj
        for k in (1, 9, 90):
            g.step(k); o.step(k)
            _same_state(g, o, f"joint inserted (warmstart_joints={ws}), +{k}")
        _compare_joints(g, o)
        g.remove_impulse_joint(1); o.remove_joint(1)
        g.step(20); o.step(20)
        jd["body1"], jd["body2"] = 2, 4
        hj = g.insert_impulse_joint(2, 4, jd); oj = lib().ro_add_joint(o._w, np.array([jd], S.JOINT_DTYPE).ctypes.data)
        # the handle names the arena slot freed above under generation 1 (impulse_joint_set.rs:48, arena.rs:260-290); the joint itself is
        # appended to the edge list, where the oracle numbers it
        assert hj == (1 << 32) | 1 and int(g.joint_handles()[oj]) == hj
        for k in (1, 9, 120):
            g.step(k); o.step(k)
            _same_state(g, o, f"joint removed, another inserted (warmstart_joints={ws}), +{k}")
        _compare_joints(g, o)


def test_reference_stress_scenes_bit_exact():
    """The two stress scenes behind the reference's bitwise goldens (which need the Rust build): simd_backend_determinism.rs
    (12x3x12 jittered pile + spherical chain, 200 steps) and parallel_path_parity.rs (14x2x14 pile that falls asleep, then ten
    rounds of twelve kicked cubes dropped onto it, 40 steps each) — against the oracle, bit for bit, sleeping flags included."""
    g, o = _compare(S.reference_pile(12, 3, 12, chain=True), [1, 10, 60, 200])
    np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    _compare_joints(g, o)
    sc = S.reference_pile(14, 2, 14, chain=False)
    g, o = _compare(sc, [1, 60, 220])
    np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    assert g.sleeping()[1:].mean() > 0.9                                   # the pile is (mostly) asleep before the drops
    from oracle_ffi import lib
    for rnd in range(10):
        for body, col in S.reference_cluster(rnd):
            hb = g.insert_body(body); g.insert_collider(col, hb)
            ob = lib().ro_add_body(o._w, np.array([body], S.BODY_DTYPE).ctypes.data); o.n += 1
            lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
            assert int(hb) & 0xFFFFFFFF == ob
        for n in (1, 19, 20):
            g.step(n); o.step(n)
            _same_state(g, o, f"drop round {rnd}, +{n}")
            np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    assert g.counters()["overflow_flags"] == 0 and g.counters()["quarantined"] == 0


def test_many_colliders_per_body_bit_exact():
    """issue_970 / issue_730 scenes (tests/test_reference_kats.py): a dynamic body made of 2,000 boxes — 2,000 manifolds between the
    same two bodies, i.e. ~1,870 of them on the serial overflow colour — and 500 balls raining on 400 sibling colliders of one
    fixed body (cuboids and capsules)."""
    import test_reference_kats as K
    sc, body = K.multi_collider_slab()
    g, o = _compare(sc, [1, 5, 40, 80])
    np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    assert g.sleeping()[body] and g.counters()["num_pairs"] == 2000
    sc, balls = K.separate_colliders_scene()
    g, o = _compare(sc, [1, 30, 100])
    np.testing.assert_array_equal(g.sleeping(), o.sleeping())


def test_runs_are_bitwise_identical_and_match_the_oracle():
    """issue_868 (8 deeply overlapping bouncy balls, a parentless ground collider): two device runs from identical initial
    conditions give identical bits — and the oracle's."""
    import test_reference_kats as K
    res = []
    for _ in range(2):
        sc, hs = K.eight_ball_drop()
        g, o = _compare(sc, [1, 20, 200])
        res.append(g.read_bodies()[0][hs].copy())
    np.testing.assert_array_equal(res[0], res[1])


def test_body_churn_bit_exact():
    """The fountain churn of solver_graph_stale_refs.rs:24-79 (cuboids / balls): one body inserted every step, the outermost
    ones removed beyond 60 live bodies — collider removal, pair deletion, in-place appends and capacity rebuilds, sleep / wake
    transitions — bit-exact against the oracle all along."""
    sc = S.Scene(name="churn", gravity=(0.0, -9.81, 0.0))
    gb = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -2.1, 0.0))
    sc.add_collider(gb, half_extents=(40.0, 2.1, 40.0))
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    alive = []
    for step_id in range(1, 260):
        g.step(1); o.step(1)
        body = S.body_desc(translation=(0.0, 10.0, 0.0), can_sleep=1)
        col = S.collider_desc(shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0)) if step_id % 3 == 0 else \
            S.collider_desc(half_extents=(0.5, 0.5, 0.5) if step_id % 3 == 2 else (0.5, 0.25, 0.5))
        hb = g.insert_body(body); g.insert_collider(col, hb)
        ob = o.add_body(translation=(0.0, 10.0, 0.0), can_sleep=1)
        from oracle_ffi import lib
        lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
        assert int(hb) & 0xFFFFFFFF == ob
        alive.append(ob)
        if len(alive) > 60:
            pos = o.read()[0]
            order = sorted(alive, key=lambda h: -(abs(pos[h, 0]) + abs(pos[h, 2])))
            for h in order[:len(alive) - 60]:
                g.remove_body(h); o.remove_body(h); alive.remove(h)
        if step_id % 20 == 0 or step_id > 250:
            gp, gv = g.read_bodies(); op, ov = o.read()
            np.testing.assert_array_equal(gp[alive], op[alive], err_msg=f"churn poses @ {step_id}")
            np.testing.assert_array_equal(gv[alive], ov[alive], err_msg=f"churn velocities @ {step_id}")
            np.testing.assert_array_equal(g.sleeping()[alive], o.sleeping()[alive])
    assert g.counters()["overflow_flags"] == 0 and g.counters()["quarantined"] == 0


def test_contact_disabling_joints_bit_exact():
    """GenericJoint::contacts_enabled = false: the pairs between the two jointed bodies are cleared (pair_update.rs:191-201)."""
    g, o = _compare(S.overlapping_chain(6, 0), [1, 2, 10, 60, 200])
    pos, vel = g.read_bodies()
    assert np.abs(vel).max() < 0.5 and np.abs(np.diff(pos[1:, 0]) - 1.0).max() < 0.05   # the overlapping links rest peacefully
    assert g.counters()["num_manifolds"] == o.stats()["num_active_manifolds"]
    g.remove_impulse_joint(2); o.remove_joint(2)               # the links it joined collide again
    for n in (1, 5, 60):
        g.step(n); o.step(n)
        _same_state(g, o, f"contact-disabling joint removed, +{n}")
    _compare(S.overlapping_chain(4, 1), [1, 10, 60])           # same chain with contacts enabled: the links push each other


def test_revolute_and_fixed_joints_bit_exact():
    """Locked angular axes (JointConstraintHelper::lock_angular): the jointed pair of test_staged.rs:86-148, a door on a
    hinge, two welded cubes; then 80 pairs so the joints fill a parallel colour (>= 64 joints)."""
    for n in (1, 80):
        g, o = _compare(S.jointed_pairs(n), [1, 2, 10, 60, 200])
        gc, gi = g.read_joints()
        oc, oi = o.read_joints()
        np.testing.assert_array_equal(gc, oc)
        np.testing.assert_array_equal(gi, oi)
    pos, vel = g.read_bodies()
    nb = len(pos)
    door, w1, w2 = nb - 3, nb - 2, nb - 1
    assert pos[door, 1] == pytest.approx(1.5, abs=1e-3) and abs(vel[door, 3]) < 1e-3 and abs(vel[door, 5]) < 1e-3   # the hinge only lets it turn about Y
    assert np.linalg.norm(pos[w2, :3] - pos[w1, :3]) == pytest.approx(1.2, abs=2e-3)                       # the weld holds


# ---- FrictionModel::Coulomb (SURVEY §8a SV1 twin): rp_coulomb_pair.h in k_island_solve_coul (islands), rp_coulomb.h on the global path and
# in k_island_generic (RP_ISL_GENERIC=1) vs the oracle ----
def _coulomb(scene):
    scene.params["friction_model"] = S.FRICTION_COULOMB
    return scene


def test_coulomb_friction_bit_exact():
    _compare(_coulomb(S.box_stack(3)), [1, 2, 10, 60])
    sl = _coulomb(S.box_stack(1)); sl.bodies[1]["linvel"] = (3.0, 0.0, 0.5)
    _compare(sl, [1, 5, 20, 80])
    g, o = _compare(_coulomb(S.pyramid10()), [1, 10, 100, 300])
    gm, gn, gi = g.contacts()
    om, on, oi = o.manifolds()
    gk = {(a, b): (c, n, tuple(i)) for (a, b, c, n), i in zip(gm.tolist(), gi.tolist())}
    ok = {(a, b): (c, n, tuple(i)) for (a, b, c, n), i in zip(om.tolist(), oi.tolist())}
    assert gk == ok
    _compare(_coulomb(S.tumble(40, seed=11)), [1, 30, 90, 250])   # restitution, balls, pair churn


def test_coulomb_multi_mode_and_model_switch():
    """Full-size b3d_many_pyramids under Coulomb friction: 196 islands on k_island_solve_coul (one workgroup per island, the lane pair
    of rp_coulomb_pair.h), fused steps and launches of several steps included; 300 steps against the 16-thread oracle."""
    of = _oracle_threads()
    try:
        g, o = _compare(_coulomb(S.many_pyramids()), [1, 5, 30, 300])
    finally:
        of.set_threads(1)
    c = g.counters()
    assert c["num_manifolds"] == 28420 and c["fused_steps"] > 200 and c["replayed_steps"] == 0, c
    # switching the model on a live world rebuilds the device world from the current state and keeps simulating
    sc = S.pyramid10()
    w = PhysicsWorld.from_scene(sc)
    w.step(30)
    p = w.integration_parameters.as_array().copy()
    p["friction_model"] = S.FRICTION_COULOMB
    w.set_integration_parameters(p)
    w.step(120)
    pos, vel = w.read_bodies()
    assert np.isfinite(pos).all() and np.abs(vel).max() < 0.05 and pos[1:, 1].max() == pytest.approx(9.5, abs=0.05)


def test_coulomb_islands_next_to_a_global_component():
    """Coulomb world with both kinds of components: pyramids (islands on k_island_generic) and a 20 x 20 pile too large for an
    island (global path, dataflow launch) — the two share the constraint planes from opposite ends."""
    sc = _coulomb(S.many_pyramids(rows=2, cols=2))
    for i in range(20):
        for j in range(20):
            b = sc.add_body(translation=(40.0 + 1.01 * i, 0.5 + 1.0 * ((i + j) % 2), 1.01 * j))
            sc.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    g, o = _compare(sc, [1, 10, 60, 150])
    assert g.counters()["num_manifolds"] > 580 + 400


def test_coulomb_model_through_the_generic_island_kernel(monkeypatch):
    """RP_ISL_GENERIC=1: the Coulomb islands on k_island_generic<true> (rows in HBM, one lane per manifold — the form of rounds 2-5, kept
    as the A/B of the lane pair): same bits."""
    monkeypatch.setenv("RP_ISL_GENERIC", "1")
    g, o = _compare(_coulomb(S.pyramid10()), [1, 10, 100])
    assert g.counters()["fused_steps"] == 0
    _compare(_coulomb(S.many_pyramids(rows=2, cols=2)), [1, 5, 40])
    _compare(_coulomb(S.tumble(40, seed=11)), [1, 30, 90])


def test_twist_model_through_the_generic_island_kernel(monkeypatch):
    """RP_ISL_GENERIC=1 sends the islands of a twist-model world through k_island_generic<false> instead of k_island_solve: the
    same scenes must come out bit for bit (the kernel's stage order is the global path's)."""
    monkeypatch.setenv("RP_ISL_GENERIC", "1")
    _compare(S.pyramid10(), [1, 10, 100, 300])
    _compare(S.many_pyramids(rows=2, cols=2), [1, 5, 40])
    _compare(S.tumble(40, seed=11), [1, 30, 90, 250])
    sc = S.sleep_impact(); _compare(sc, [60, 200])


# ---- kinematic bodies (SURVEY §8a MISC: interpolate_kinematic_velocities) ----
def test_kinematic_velocity_platform_bit_exact():
    g, o = _compare(S.kinematic_platform(False), [1, 2, 10, 60, 150])
    np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    gm, _, gi = g.contacts()
    om, _, oi = o.manifolds()
    assert {(a, b): (c, n, tuple(i)) for (a, b, c, n), i in zip(gm.tolist(), gi.tolist())} == \
           {(a, b): (c, n, tuple(i)) for (a, b, c, n), i in zip(om.tolist(), oi.tolist())}   # same colours and impulses
    pos, vel = g.read_bodies()
    assert pos[1, 0] == pytest.approx(1.5, abs=1e-3) and vel[2, 0] == pytest.approx(0.6, abs=0.1)  # the box rides the platform


def test_kinematic_position_platform_sleep_and_wake_bit_exact():
    sc = S.kinematic_platform(True).enable_sleep()
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)

    def target(k):
        t = k / 60.0
        return np.array([0.6 * t, 1.0 + 0.15 * t, 0.0, 0.0, np.sin(0.1 * t), 0.0, np.cos(0.1 * t)], np.float32)

    for k in range(1, 91):
        g.set_next_kinematic_position([1], target(k)); o.set_next_kinematic_position(1, target(k))
        g.step(1); o.step(1)
        if k % 10 == 0:
            _same_sleep_state(g, o, f"kinematic position platform @ {k}")
    pos, vel = g.read_bodies()
    np.testing.assert_array_equal(pos[1], target(90))        # lands exactly on the pose the user asked for
    assert vel[1, 0] == pytest.approx(0.6, abs=1e-3) and vel[1, 4] == pytest.approx(0.2, abs=1e-3)
    for n in (1, 30, 60):                                    # no new target: the platform stops where it is
        g.step(n); o.step(n)
        _same_sleep_state(g, o, f"kinematic platform at rest, +{n}")
    # (an idle position-based platform keeps a rounding-sized interpolated velocity, so only "exactly zero" velocity-based
    # platforms ever satisfy the kinematic sleep rule, rigid_body_components.rs:1464-1468)


def test_kinematic_velocity_platform_sleep_and_wake_bit_exact():
    sc = S.kinematic_platform(False).enable_sleep()
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(40); o.step(40)
    _same_sleep_state(g, o, "moving platform")
    assert not g.sleeping()[1:5].any()                       # a moving platform keeps its island awake
    stop = np.zeros((1, 6), np.float32)
    g.write_bodies([1], vel6=stop); o.set_vel(1, stop[0, :3], stop[0, 3:])
    for n in (1, 40, 60, 100):
        g.step(n); o.step(n)
        _same_sleep_state(g, o, f"platform stopped, +{n}")
    assert g.sleeping()[1:].all()                            # platform (exactly zero velocity) and boxes sleep as one island
    go = np.array([[0.0, 0.0, 0.5, 0.0, 0.0, 0.0]], np.float32)
    g.write_bodies([1], vel6=go); o.set_vel(1, go[0, :3], go[0, 3:])   # set_linvel(.., wake_up = true) wakes the island
    for n in (1, 1, 30):
        g.step(n); o.step(n)
        _same_sleep_state(g, o, f"platform moving again, +{n}")
    assert not g.sleeping()[1:5].any() and g.sleeping()[5]   # the free cube on the floor sleeps on


# ---- events (SURVEY §8a MISC emit_contact_force_events, §8f.3 collision events) ----
def _sorted_events(ev):
    return sorted(tuple(int(x) for x in e) for e in ev)


def _same_events(g, o, msg):
    assert _sorted_events(g.collision_events()) == _sorted_events(o.collision_events()), msg
    gm, gv = g.contact_force_events()
    om, ov = o.force_events()
    go = np.lexsort((gm[:, 1], gm[:, 0], gm[:, 2])) if len(gm) else np.zeros(0, int)
    oo = np.lexsort((om[:, 1], om[:, 0], om[:, 2])) if len(om) else np.zeros(0, int)
    np.testing.assert_array_equal(gm[go], om[oo], err_msg=msg + " (force event pairs / steps / started)")
    np.testing.assert_array_equal(gv[go], ov[oo], err_msg=msg + " (force event values)")


def test_collision_and_contact_force_events_bit_exact():
    sc = S.box_stack(2, gap=0.5).enable_events(S.ACTIVE_EVENTS_COLLISION | S.ACTIVE_EVENTS_CONTACT_FORCE, 15.0)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(80); o.step(80)
    _same_state(g, o, "events scene")
    _same_events(g, o, "box stack events")
    g.remove_body(1); o.remove_body(1)
    g.step(40); o.step(40)
    _same_events(g, o, "events after removing the lower box")
    assert len(g.collision_events()) == 0                       # drained
    sc2 = S.tumble(40, seed=11).enable_events(S.ACTIVE_EVENTS_COLLISION | S.ACTIVE_EVENTS_CONTACT_FORCE, 2.0)
    g2, o2 = PhysicsWorld.from_scene(sc2), OracleWorld(sc2)
    for cp in (30, 120, 250):
        g2.step(cp); o2.step(cp)
        _same_state(g2, o2, f"tumble with events +{cp}")
        _same_events(g2, o2, f"tumble events +{cp}")
    sc3 = S.many_pyramids(rows=2, cols=2).enable_events(S.ACTIVE_EVENTS_COLLISION, 0.0)   # collision events ride the fast path
    g3, o3 = PhysicsWorld.from_scene(sc3), OracleWorld(sc3)
    g3.step(100); o3.step(100)
    _same_state(g3, o3, "pyramids with collision events")
    _same_events(g3, o3, "pyramid collision events")
    assert g3.counters()["fast_steps"] > 0


def test_pair_pool_grows_before_it_overflows(monkeypatch):
    """1,300 tumbling bodies with ONE pair slot per collider row to start with (RP_PAIRS_PER_COLLIDER=1: 2,905 slots): the pile needs
    several times that.  rp_step watches the pool through the hint buffer and moves the world to arrays with twice the slots whenever
    70 % are in use — every pair keeps its manifold, impulses and colour (the growth carry-over) — so nothing overflows and nothing
    differs from the oracle"""
    monkeypatch.setenv("RP_PAIRS_PER_COLLIDER", "1")
    sc = S.tumble(1300, seed=4)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for n in (1, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 30, 60, 100):   # (rp_step per frame, like a game loop)
        g.step(n); o.step(n)
        _same_state(g, o, f"tumble with a growing pair pool, +{n}")
    c = g.counters()
    assert c["overflow_flags"] == 0 and c["num_pairs"] > 2905, c
    assert c["num_pairs"] == o.stats()["num_pairs"]


def test_first_step_pool_overflow_keeps_device_writes_and_step_count(monkeypatch):
    """the very first broad-phase pass of a dense world overflows a pair pool of ONE slot per collider row, so step_once
    rebuilds the device world with twice the slots and runs the step again.  (i) What the user wrote into device rows between the
    auto-finalize (step(0)) and that step — an impulse, a force, a velocity — must survive the rebuild; (ii) the host's step counter
    must not fall behind the device's: the first aborted fast step afterwards would otherwise be taken for a retired one and never be
    replayed.  36 pyramids settle onto the fast path; a kick then makes fast steps abort."""
    import os
    import oracle_ffi
    monkeypatch.setenv("RP_PAIRS_PER_COLLIDER", "1")
    sc = S.many_pyramids(rows=6, cols=6)
    oracle_ffi.set_threads(max(1, min(os.cpu_count() or 1, 16)))
    try:
        g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
        g.step(0)                                  # the device world exists; no step has run
        dyn = [i for i, b in enumerate(sc.bodies) if int(b["body_type"]) == S.BODY_DYNAMIC]
        a, b, c = dyn[5], dyn[300], dyn[-1]
        g.apply_impulse([a], impulse=(40.0, 90.0, -15.0)); o.apply_impulse(a, impulse=(40.0, 90.0, -15.0))
        g.add_force([b], force=(0.0, 500.0, 100.0)); o.add_force(b, force=(0.0, 500.0, 100.0))
        kick = np.array([[1.0, 4.0, 0.5, 0.0, 2.0, 0.0]], np.float32)
        g.write_bodies([c], vel6=kick); o.set_vel(c, kick[0, :3], kick[0, 3:])
        for n in (1, 1, 30, 170):
            g.step(n); o.step(n)
            _same_state(g, o, f"first-step pool overflow, +{n}")
        c0 = g.counters()
        assert c0["overflow_flags"] == 0 and c0["num_pairs"] > len(sc.colliders) + 1024, c0   # more pairs than the first pool held
        assert c0["fast_steps"] > 20, c0
        g.write_bodies([c], vel6=kick); o.set_vel(c, kick[0, :3], kick[0, 3:])
        for n in (1, 20, 100):
            g.step(n); o.step(n)
            _same_state(g, o, f"after the kick, +{n}")
        c1 = g.counters()
        assert c1["replayed_steps"] > 0 or c1["full_steps"] > c0["full_steps"], c1
    finally:
        oracle_ffi.set_threads(1)


def test_event_queues_hold_every_pair_of_a_large_world():
    """27 x 27 pyramids with collision events on every collider: ~106,000 pairs begin to touch within the first steps — more events than
    the 65,536 slots the queues used to have.  The queues are sized by the pair pool (a step raises at most one event per pair): nothing
    is dropped, the drained stream equals the oracle's"""
    import oracle_ffi
    sc = S.many_pyramids(rows=27, cols=27).enable_events(S.ACTIVE_EVENTS_COLLISION, 0.0)
    oracle_ffi.set_threads(16)
    try:
        g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
        g.step(3); o.step(3)
        ge, oe = g.collision_events(), o.collision_events()
    finally:
        oracle_ffi.set_threads(1)
    assert len(oe) > 65536 and len(ge) == len(oe), (len(ge), len(oe))
    key = lambda e: np.lexsort((e[:, 2], e[:, 1], e[:, 0], e[:, 4]))
    np.testing.assert_array_equal(ge[key(ge)], oe[key(oe)])


def test_contact_force_events_ride_the_fast_graph():
    """Worlds with ActiveEvents::CONTACT_FORCE_EVENTS take the two-kernel fast graph + k_force_events once they have settled; a kick
    in the middle aborts fast steps on the device (replayed on the full graph) without losing or duplicating a force event."""
    sc = S.many_pyramids(rows=2, cols=2).enable_events(S.ACTIVE_EVENTS_COLLISION | S.ACTIVE_EVENTS_CONTACT_FORCE, 20.0)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for k in range(6):
        g.step(40); o.step(40)
        _same_state(g, o, f"force events on the fast graph +{40 * (k + 1)}")
        _same_events(g, o, f"force events on the fast graph +{40 * (k + 1)}")
        if k == 3:  # knock the top cube of a pyramid off: new pairs, full updates, then settling again
            top = 55
            v = np.array([[3.0, 1.0, 0.5, 0.0, 0.0, 0.0]], np.float32)
            g.write_bodies([top], vel6=v); o.set_vel(top, v[0, :3], v[0, 3:])
    c = g.counters()
    assert c["fast_steps"] > 60 and c["overflow_flags"] == 0


def test_worlds_stepped_concurrently_from_threads():
    """Three worlds on one GPU, each stepped from its own host thread (one thread per world is the ABI's contract): their streams
    interleave on the device — fused fast steps next to rebuild kernels with grid barriers (rp_gridbar.h), which all assume their
    workgroups resident — and every world still matches the oracle bit for bit."""
    import threading
    scenes = [S.tumble(64, seed=3), S.many_pyramids(rows=3, cols=3).enable_sleep(), S.joint_chain(6, with_boxes=True)]
    worlds = [PhysicsWorld.from_scene(sc) for sc in scenes]
    errors = []

    def drive(w):
        try:
            for _ in range(30):
                w.step(5)
                w.sync()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=drive, args=(w,)) for w in worlds]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for sc, w in zip(scenes, worlds):
        o = OracleWorld(sc)
        o.step(150)
        _same_state(w, o, f"{sc.name} stepped next to two other worlds")
        assert w.counters()["overflow_flags"] == 0


def test_out_of_scope_inputs_are_refused():
    """Unknown joint axis masks and body types are refused loudly, not mis-simulated."""
    from rapier_amd import RapierHipError
    w = PhysicsWorld()
    b = w.insert_body(S.body_desc(translation=(0.0, 1.0, 0.0)))
    w.insert_collider(S.collider_desc(), b)
    sc = S.Scene(name="tmp")
    sc.add_body(); sc.add_body()
    sc.add_joint(0, 1, (0, 0, 0), (0, 0, 0), locked_axes=0x7F)
    with pytest.raises(RapierHipError):
        w.insert_impulse_joints(sc.joint_array())  # not a JointAxesMask of locked axes
    with pytest.raises(RapierHipError):
        w.insert_body(S.body_desc(body_type=7))    # not a RigidBodyType
    p = S.default_params(); p["friction_model"] = 5
    with pytest.raises(RapierHipError):
        w.set_integration_parameters(p)            # not a FrictionModel


def test_golden_fixtures_on_gpu():
    import glob
    import os
    from golden.make_golden import CASES
    for f in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))):
        d = np.load(f)
        scene, steps = CASES[os.path.basename(f)[:-4]]()
        w = PhysicsWorld.from_scene(scene)
        w.step(steps)
        pos, vel = w.read_bodies()
        np.testing.assert_array_equal(pos, d["pos"])
        np.testing.assert_array_equal(vel, d["vel"])


def test_empty_and_contactless_worlds():
    w = PhysicsWorld(gravity=(0, -9.81, 0))
    w.step(3)
    pos, vel = w.read_bodies()
    assert pos.shape == (0, 7)
    sc = S.Scene(name="free", gravity=(0.0, -10.0, 0.0))
    b = sc.add_body(translation=(0.0, 10.0, 0.0), linvel=(1.0, 0.0, 0.0))
    sc.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    _compare(sc, [1, 30])


def test_graph_and_eager_paths_agree(monkeypatch):
    """The captured hipGraph replay and the eager launch sequence must produce identical state."""
    sc = S.many_pyramids(rows=2, cols=2)
    a = PhysicsWorld.from_scene(sc)
    a.step(40)
    pa, va = a.read_bodies()
    monkeypatch.setenv("RP_NO_GRAPH", "1")
    b = PhysicsWorld.from_scene(sc)
    b.step(40)
    pb, vb = b.read_bodies()
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(va, vb)


def test_stage_counters_from_hip_events(monkeypatch):
    """Counters / StagesCounters / CollisionDetectionCounters mirror: timed steps report every stage, and timing does not
    change the simulation."""
    sc = S.many_pyramids(rows=2, cols=2)
    ref = PhysicsWorld.from_scene(sc)
    ref.step(50)
    monkeypatch.setenv("RP_NO_FAST", "1")
    w = PhysicsWorld.from_scene(sc)
    w.step(10)
    w.enable_timers(True)
    w.step(40)
    c = w.counters()
    assert c["full_steps"] >= 50 and c["fast_steps"] == 0
    for k in ("step_time_ms", "collision_detection_ms", "broad_phase_ms", "narrow_phase_ms", "island_construction_ms", "solver_ms", "velocity_resolution_ms"):
        assert 0.0 < c[k] < 50.0, (k, c[k])
    assert c["collision_detection_ms"] == pytest.approx(c["broad_phase_ms"] + c["narrow_phase_ms"] + c["island_construction_ms"], rel=0.05)
    w.enable_timers(False)
    pa, va = ref.read_bodies(); pb, vb = w.read_bodies()
    np.testing.assert_array_equal(pa, pb); np.testing.assert_array_equal(va, vb)


def test_full_size_properties_many_pyramids():
    """Size-independent properties at the full BASELINE size: finite, settled, weight carried by the
    ground contacts (sum of ground impulses = N m g dt), every colour body-disjoint."""
    sc = S.many_pyramids()
    w = PhysicsWorld.from_scene(sc)
    w.step(300)
    pos, vel = w.read_bodies()
    assert np.isfinite(pos).all()
    assert np.abs(vel).max() < 0.05
    meta, nrm, imp = w.contacts()
    ground = meta[:, 2] == 127
    expected = 10780 * 100.0 * 10.0 / 60.0
    assert abs(imp[ground].sum() - expected) <= expected * 1e-3
    parents = sc.parent_array()
    for color in np.unique(meta[:, 2]):
        sel = meta[meta[:, 2] == color]
        b = np.concatenate([parents[sel[:, 0]], parents[sel[:, 1]]])
        b = b[b > 0]  # body 0 is the fixed ground
        assert len(np.unique(b)) == len(b)


# ---- GenericJoint::coupled_axes: rope, spring, two coupled angular axes (oracle side: tests/test_coupled_joints_oracle.py) ----
def test_rope_spring_and_coupled_angular_limits_bit_exact():
    from test_coupled_joints_oracle import rope_scene, spring_scene, cone_scene
    for sc, cps in ((rope_scene(2.0), [1, 2, 30, 200, 600]), (spring_scene(), [1, 2, 50, 400, 1200]), (cone_scene()[0], [1, 2, 20, 100, 300])):
        g, o = _compare(sc, cps)
        gj = g.joint_impulses() if hasattr(g, "joint_impulses") else None
    # a mixed world: a chain of ropes and springs between stacked boxes, next to ordinary joints (row counts 1 .. 6 in one joint set)
    sc = S.joint_chain(4, with_boxes=True)
    a = sc.add_body(body_type=S.BODY_FIXED, translation=(6.0, 6.0, 0.0))
    prev = a
    for k in range(5):
        b = sc.add_body(translation=(6.0 + 0.4 * (k + 1), 6.0 - 0.5 * (k + 1), 0.1 * k))
        sc.add_collider(b, half_extents=(0.2, 0.2, 0.2))
        if k % 2 == 0:
            sc.add_rope_joint(prev, b, (0.0, -0.2, 0.0), (0.0, 0.2, 0.0), 0.6)
        else:
            sc.add_spring_joint(prev, b, (0.0, -0.2, 0.0), (0.0, 0.2, 0.0), 0.4, 300.0, 3.0)
        prev = b
    _compare(sc, [1, 5, 60, 240, 480])
