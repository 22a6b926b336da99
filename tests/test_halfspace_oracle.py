"""Half-spaces in the oracle (ColliderBuilder::halfspace; parry contact_manifold_halfspace_pfm / convex_ball with a HalfSpace,
HalfSpace::aabb, MassProperties::zero): outcome tests on the CPU restatement — the device path is compared with it bit for bit in
test_gpu_halfspace.py."""
import numpy as np

from rapier_amd import scenes as S
from oracle_ffi import OracleWorld


def _drop_scene(ground):
    s = S.Scene(name="hs_drop", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=S.BODY_FIXED)
    if ground == "halfspace":
        s.add_collider(g, shape=S.SHAPE_HALFSPACE, half_extents=(0.0, 1.0, 0.0))
    else:
        s.add_collider(g, half_extents=(50.0, 0.5, 50.0), translation=(0.0, -0.5, 0.0))
    b = s.add_body(translation=(0.0, 2.0, 0.0), rotation=(0.1, 0.2, 0.05, 0.97)); s.add_collider(b, half_extents=(0.5, 0.3, 0.4))
    b = s.add_body(translation=(3.0, 2.0, 0.0)); s.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0))
    b = s.add_body(translation=(6.0, 2.0, 0.0), rotation=(0.3, 0.0, 0.2, 0.93)); s.add_collider(b, shape=S.SHAPE_CAPSULE, half_extents=(0.6, 0.3, 1.0))
    return s


def test_every_shape_rests_on_a_plane_like_on_a_slab():
    rest = {}
    for ground in ("halfspace", "slab"):
        w = OracleWorld(_drop_scene(ground)); w.step(300)
        pos, vel = w.read()
        assert np.abs(vel[1:]).max() < 1e-3
        rest[ground] = pos[1:, 1]
    np.testing.assert_allclose(rest["halfspace"], [0.3, 0.5, 0.3], atol=2e-3)   # smallest half extent, ball radius, capsule radius
    np.testing.assert_allclose(rest["halfspace"], rest["slab"], atol=1e-4)


def test_pairs_with_a_half_space_never_recycle():
    """the recycle extent of a half-space is |(MAX/2, MAX/2, MAX/2)| = inf (pair_update.rs:588-596), so the drift test can never
    pass: such pairs take a full narrow-phase update every step, in the reference too"""
    w = OracleWorld(_drop_scene("halfspace")); w.step(200)
    st = w.stats()
    assert st["num_pairs"] == 3 and st["num_full_updates"] == 3 and st["num_recycled"] == 0
    w = OracleWorld(_drop_scene("slab")); w.step(200)
    assert w.stats()["num_recycled"] == 3


def test_frictionless_slope_accelerates_at_g_sin_theta():
    s = S.Scene(name="hs_slope", gravity=(0.0, -9.81, 0.0))
    s.add_collider(-1, shape=S.SHAPE_HALFSPACE, half_extents=(-0.6, 0.8, 0.0), friction=0.0)   # 36.87 degrees
    b = s.add_body(translation=(0.0, 0.31, 0.0), rotation=(0.0, 0.0, 0.3162278, 0.9486833))    # a box lying on the slope
    s.add_collider(b, half_extents=(0.4, 0.3, 0.4), friction=0.0)
    w = OracleWorld(s); w.step(30)
    v0 = w.read()[1][b, :3].copy()
    w.step(60)
    v1 = w.read()[1][b, :3]
    a = (v1 - v0) / 1.0                       # 60 steps of 1/60 s
    along = np.array([0.8, 0.6, 0.0])         # down-slope direction is -along
    assert abs(np.dot(a, -along) - 9.81 * 0.6) < 0.05 and abs(np.dot(a, [-0.6, 0.8, 0.0])) < 0.05


def test_a_ball_below_the_plane_is_pushed_back_out():
    """HalfSpace::project_local_point(.., false): a centre on the solid side still projects onto the plane and the contact normal
    stays the plane's (the `proj.is_inside` branch of contact_manifold_convex_ball)"""
    s = S.Scene(name="hs_deep", gravity=(0.0, 0.0, 0.0))
    s.add_collider(-1, shape=S.SHAPE_HALFSPACE, half_extents=(0.0, 1.0, 0.0))
    b = s.add_body(translation=(0.0, -0.2, 0.0)); s.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0))
    w = OracleWorld(s)
    w.step(1)
    meta, normals, _ = w.manifolds()
    assert len(meta) == 1 and tuple(normals[0]) == (0.0, 1.0, 0.0)
    w.step(120)
    pos, _ = w.read()
    assert pos[b, 1] > 0.45 and abs(pos[b, 0]) < 1e-6


def test_a_ball_centre_inside_a_cuboid_is_pushed_through_the_nearest_face():
    s = S.Scene(name="deep_ball", gravity=(0.0, 0.0, 0.0))
    g = s.add_body(body_type=S.BODY_FIXED); s.add_collider(g, half_extents=(2.0, 1.0, 2.0))
    b = s.add_body(translation=(0.3, 0.9, 0.1)); s.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.25, 0.0, 0.0))
    w = OracleWorld(s); w.step(180)
    pos, _ = w.read()
    assert pos[b, 1] > 1.2 and abs(pos[b, 0] - 0.3) < 1e-5 and abs(pos[b, 2] - 0.1) < 1e-5   # straight up through the +Y face


def test_half_space_sensor_reports_the_side_a_body_is_on():
    s = S.Scene(name="hs_sensor", gravity=(0.0, -9.81, 0.0))
    hs = s.add_collider(-1, shape=S.SHAPE_HALFSPACE, half_extents=(0.0, 1.0, 0.0), sensor=1, active_events=S.ACTIVE_EVENTS_COLLISION)
    cols = []
    for k, (shape, he) in enumerate([(S.SHAPE_BALL, (0.3, 0, 0)), (S.SHAPE_CUBOID, (0.3, 0.2, 0.1)), (S.SHAPE_CAPSULE, (0.4, 0.2, 0.0))]):
        b = s.add_body(translation=(2.0 * k, 1.0 + 0.5 * k, 0.0), rotation=(0.2, 0.1, 0.3, 0.9273618))
        cols.append(s.add_collider(b, shape=shape, half_extents=he))
    w = OracleWorld(s)
    w.step(1)
    assert [w.intersection_pair(hs, c) for c in cols] == [False, False, False]
    started = []
    for _ in range(90):
        w.step(1)
        started += [int(e[1]) for e in w.collision_events() if int(e[2]) == 1 and int(e[3]) & 1]
    assert started == cols                                         # they cross the plane in drop-height order
    assert [w.intersection_pair(hs, c) for c in cols] == [True, True, True]


def test_half_space_weighs_nothing_and_keeps_the_broad_phase_quiet():
    s = S.halfspace_scene()
    w = OracleWorld(s)
    w.step(5)
    assert w.stats()["bp_rebuilt"] in (0, 1)
    mp = w.mass_props(0)                                           # the kinematic lift (body 0): only a half-space attached
    assert mp[0] == 0.0 and not mp[4:7].any()                      # inverse mass and inverse inertia of a massless body
