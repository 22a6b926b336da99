"""Composite shapes as one collider + contact clustering in the oracle (oracle/ro_composite.h): ColliderBuilder::compound / trimesh /
heightfield (collider.rs:711, :944, :1089), cluster_manifolds_for_solver / carry_warmstart_data (contact_clustering.rs:33, :129), the
`manifolds.len() > 1` gate (pair_update.rs:350) and the overflow colour of a pair's second solver manifold (solver_graph.rs:534-547).
Outcome-level checks against code-independent references (closed forms, the same scene built from separate colliders)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rapier_amd import scenes as S  # noqa: E402
from oracle_ffi import OracleWorld  # noqa: E402


def _grid_mesh(n=4, size=8.0, y=0.0):
    """a flat n x n grid of quads (2 triangles each) spanning [-size/2, size/2]^2 at height y"""
    xs = np.linspace(-size / 2, size / 2, n + 1)
    v = np.array([[x, y, z] for z in xs for x in xs], np.float32)
    t = []
    for r in range(n):
        for c in range(n):
            a = r * (n + 1) + c
            t += [[a, a + n + 1, a + n + 2], [a, a + n + 2, a + 1]]
    return v, np.array(t, np.uint32)


def _box_on_mesh(x=0.3, z=0.2):
    s = S.Scene(name="box_on_mesh", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=S.BODY_FIXED)
    v, t = _grid_mesh()
    mid = s.add_trimesh(v, t)
    s.add_collider(g, shape=S.SHAPE_TRIMESH, half_extents=(mid, 0, 0))
    b = s.add_body(translation=(x, 0.6, z))
    s.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    return s, b


def test_a_cuboid_rests_on_a_triangle_mesh_through_one_cluster():
    s, b = _box_on_mesh()
    o = OracleWorld(s)
    o.step(240)
    pos, vel = o.read()
    assert abs(pos[b, 1] - 0.5) < 5e-3 and np.abs(vel[b]).max() < 2e-2          # at rest on the plane y = 0, like on a slab
    ncl, nsc = o.pair_clusters(0, 1)
    assert ncl == 1 and nsc[0] >= 3                                              # several triangles, ONE cluster (same normal), reduced to <= 4 points
    assert o.stats()["num_active_manifolds"] == 1


def test_clusters_carry_warm_start_like_a_slab_does():
    """the same box on a slab (one cuboid-cuboid manifold with feature-tracked warm start) and on the mesh (clusters, position-matched
    warm start) settle to the same height and stay there: the carried impulses hold the box from the first resting step on"""
    s, b = _box_on_mesh()
    slab = S.Scene(name="box_on_slab", gravity=(0.0, -9.81, 0.0))
    g = slab.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0)); slab.add_collider(g, half_extents=(4.0, 0.5, 4.0))
    bb = slab.add_body(translation=(0.3, 0.6, 0.2)); slab.add_collider(bb, half_extents=(0.5, 0.5, 0.5))
    om, osl = OracleWorld(s), OracleWorld(slab)
    hist = []
    for _ in range(200):
        om.step(1); osl.step(1)
        hist.append((om.read()[0][b, 1], osl.read()[0][bb, 1]))
    hist = np.array(hist)
    assert np.abs(hist[-50:, 0] - hist[-50:, 1]).max() < 2e-3
    assert np.abs(np.diff(hist[-50:, 0])).max() < 1e-4                            # no jitter once warm-started


def test_a_box_in_a_mesh_corner_gets_two_clusters_and_the_second_goes_to_the_overflow_colour():
    s = S.Scene(name="corner", gravity=(-4.0, -9.81, 0.0))                           # gravity leans into the wall
    g = s.add_body(body_type=S.BODY_FIXED)
    # floor (y = 0) + wall (x = 0), two triangles each, as ONE mesh collider
    v = np.array([[0, 0, -3], [6, 0, -3], [6, 0, 3], [0, 0, 3], [0, 4, -3], [0, 4, 3]], np.float32)
    t = np.array([[0, 2, 1], [0, 3, 2], [0, 4, 5], [0, 5, 3]], np.uint32)
    mid = s.add_trimesh(v, t)
    s.add_collider(g, shape=S.SHAPE_TRIMESH, half_extents=(mid, 0, 0), friction=0.0)
    b = s.add_body(translation=(0.6, 0.55, 0.0))
    s.add_collider(b, half_extents=(0.5, 0.5, 0.5), friction=0.0)
    o = OracleWorld(s)
    o.step(120)
    pos, vel = o.read()
    assert abs(pos[b, 0] - 0.5) < 1e-2 and abs(pos[b, 1] - 0.5) < 1e-2 and np.abs(vel[b]).max() < 5e-2   # pushed into the corner: held by floor AND wall
    ncl, nsc = o.pair_clusters(0, 1)
    assert ncl == 2 and nsc[0] > 0 and nsc[1] > 0
    st = o.stats()
    assert st["num_active_manifolds"] == 2                                       # two solver manifolds of one pair
    meta, _, imp = o.manifolds()
    assert sorted(int(x) for x in meta[:, 2])[-1] == 128 and (imp.sum(axis=1) > 0).all()   # the second one sits in the overflow colour; both carry load


def test_compound_mass_properties_are_the_sum_of_the_parts_and_it_moves_like_the_multi_collider_body():
    """an L of two cuboids as ONE compound collider against the same L as two colliders of one body: same mass properties, same free
    flight (bit for bit: no contact involved), and both come to rest on the ground at the same pose within contact tolerance"""
    parts = [S.collider_desc(half_extents=(0.5, 0.25, 0.25), translation=(0.0, 0.0, 0.0)), S.collider_desc(half_extents=(0.25, 0.5, 0.25), translation=(0.75, 0.25, 0.0))]

    def scene(as_compound):
        s = S.Scene(name="ell", gravity=(0.0, -9.81, 0.0))
        g = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0)); s.add_collider(g, half_extents=(5.0, 0.5, 5.0))
        b = s.add_body(translation=(0.0, 2.0, 0.0), angvel=(0.3, 0.2, 0.1))
        if as_compound:
            cid = s.add_compound(parts)
            s.add_collider(b, shape=S.SHAPE_COMPOUND, half_extents=(cid, 0, 0))
        else:
            for p in parts:
                s.add_collider(b, half_extents=tuple(p["half_extents"]), translation=tuple(p["translation"]))
        return s, b
    (sa, ba), (sb, bb) = scene(True), scene(False)
    oa, ob = OracleWorld(sa), OracleWorld(sb)
    ma, mb = oa.mass_props(ba), ob.mass_props(bb)
    np.testing.assert_allclose(ma, mb, rtol=2e-5, atol=2e-6)
    oa.step(20); ob.step(20)                                                      # free flight
    np.testing.assert_allclose(oa.read()[0][ba], ob.read()[0][bb], rtol=0, atol=2e-5)
    oa.step(400); ob.step(400)
    pa, pb = oa.read()[0][ba], ob.read()[0][bb]
    assert abs(pa[1] - pb[1]) < 5e-3 and np.abs(oa.read()[1][ba]).max() < 5e-2


def test_heightfield_is_served_as_a_mesh_and_a_ball_rolls_to_the_valley():
    s = S.Scene(name="hf", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=S.BODY_FIXED)
    n = 9
    r = np.linspace(-1, 1, n)
    h = (r[None, :] ** 2 + 0 * r[:, None]).astype(np.float32)                      # a parabolic trough along z: height = x^2
    hid = s.add_heightfield(h, (8.0, 1.0, 8.0))
    s.add_collider(g, shape=S.SHAPE_TRIMESH, half_extents=(hid, 0, 0), friction=0.8)
    b = s.add_body(translation=(2.5, 1.5, 0.3))
    s.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.4, 0, 0), friction=0.8)
    o = OracleWorld(s)
    o.step(900)
    pos, vel = o.read()
    assert abs(pos[b, 0]) < 0.6 and pos[b, 1] < 0.6 and np.isfinite(pos).all()     # rolled down into the valley (x ~ 0) and stayed on the surface


def test_sensor_with_a_composite_side_reports_intersection():
    s = S.Scene(name="sensor", gravity=(0.0, 0.0, 0.0))
    g = s.add_body(body_type=S.BODY_FIXED)
    cid = s.add_compound([S.collider_desc(half_extents=(0.5, 0.5, 0.5), translation=(-2.0, 0, 0)), S.collider_desc(shape=S.SHAPE_BALL, half_extents=(0.5, 0, 0), translation=(2.0, 0, 0))])
    s.add_collider(g, shape=S.SHAPE_COMPOUND, half_extents=(cid, 0, 0), sensor=1, active_events=1)
    b = s.add_body(translation=(2.0, 3.0, 0.0), linvel=(0.0, -2.0, 0.0))
    s.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.3, 0, 0), active_events=1)
    o = OracleWorld(s)
    started = stopped = 0
    for _ in range(240):
        o.step(1)
        for e in o.collision_events():
            started += int(e[2]) & 1; stopped += 1 - (int(e[2]) & 1)
    assert started == 1 and stopped == 1                                           # the ball passes through the compound's ball part once


def test_invalid_composites_are_refused():
    s = S.Scene(name="bad")
    o = OracleWorld(s)
    assert o.add_composite(("compound", np.array([S.collider_desc(shape=S.SHAPE_HALFSPACE, half_extents=(0, 1, 0))], S.COLLIDER_DTYPE))) < 0
    assert o.add_composite(("trimesh", np.zeros((3, 3), np.float32), np.array([[0, 1, 7]], np.uint32))) < 0
    assert o.add_composite(("heightfield", np.zeros((1, 4), np.float32), np.ones(3, np.float32))) < 0


def test_the_references_composite_demos_run_sanely():
    """examples3d/compound3.rs, heightfield3.rs, trimesh3.rs restated (rapier_amd/scenes.py) at a reduced count: everything stays
    finite and on top of its ground; the terrain as a height field and as the mesh HeightField::to_trimesh gives are the same world"""
    s = S.compound3(4, 6)
    w = OracleWorld(s); w.step(240)
    pos, vel = w.read()
    assert np.isfinite(pos).all() and np.isfinite(vel).all() and pos[1:, 1].min() > 0.15 and np.abs(vel[:, :3]).max() < 40.0
    a, b = OracleWorld(S.heightfield3(4, 12)), OracleWorld(S.heightfield3(4, 12, mesh=True))
    a.step(240); b.step(240)
    pa, pb = a.read()[0], b.read()[0]
    np.testing.assert_array_equal(pa, pb)
    assert np.isfinite(pa).all() and pa[1:, 1].min() > -2.5 and np.abs(pa[1:, [0, 2]]).max() < 50.0
