"""Convex polyhedra on the device path (RP_SHAPE_CONVEX_POLYHEDRON = ColliderBuilder::convex_hull / convex_mesh, collider.rs:1039, :1070)
against the oracle, bit for bit: the library's own hull and canonical form (rp_polyhedron.h) against the oracle's (fed Qhull's
triangles), support scans and support faces from the cv_* device tables through GJK / EPA (rp_convex.h), against every other shape,
with sensors, CCD, sleeping, compound bodies, shared polyhedra and registration in a running world."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S
from rapier_amd.world import RapierHipError
from oracle_ffi import OracleWorld, lib, hull_triangles

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


def test_the_library_holds_the_polyhedra_the_oracle_holds():
    sc = S.polyhedra_clutter(8, 2)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)      # the library takes the hull itself, the oracle is handed Qhull's triangles
    for pid in range(len(sc.polyhedra)):
        a, b = g.read_convex_polyhedron(pid), o.read_convex_polyhedron(pid)
        assert a["n_edges"] == b["n_edges"]
        for k in ("points", "face_normals", "face_first", "face_count", "loop_vertex", "loop_edge", "props"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"polyhedron {pid}: {k}")


@pytest.mark.parametrize("seed", [2, 3])
def test_clutter_of_polyhedra_bit_exact(seed):
    sc = S.polyhedra_clutter(28, seed)
    for step, g, o, ev in _lockstep(sc, 400, every=4):
        pass
    c = g.counters()
    assert c["num_manifolds"] == o.stats()["num_active_manifolds"] and c["num_manifolds"] > 30
    gm, gn, gi = g.contacts()
    om, on, oi = o.manifolds()
    gk = {(a, b): (c_, n, tuple(i), tuple(nn)) for (a, b, c_, n), i, nn in zip(gm.tolist(), gi.tolist(), gn.tolist())}
    ok = {(a, b): (c_, n, tuple(i), tuple(nn)) for (a, b, c_, n), i, nn in zip(om.tolist(), oi.tolist(), on.tolist())}
    assert gk == ok
    pos, _ = g.read_bodies()
    dyn = [i for i, b in enumerate(sc.bodies) if int(b["body_type"]) == S.BODY_DYNAMIC]
    assert pos[dyn, 1].min() > 0.0


def test_convex_mesh_and_convex_hull_give_the_same_world():
    """the same scene with every polyhedron handed over as an explicit triangle list (convex_mesh, Qhull's triangulation)"""
    sc = S.polyhedra_clutter(16, 4)
    sc2 = S.polyhedra_clutter(16, 4)
    sc2.polyhedra = [(p, hull_triangles(p)) for p, _ in sc2.polyhedra]
    a, b = PhysicsWorld.from_scene(sc), PhysicsWorld.from_scene(sc2)
    a.step(200); b.step(200)
    pa, va = a.read_bodies(); pb, vb = b.read_bodies()
    np.testing.assert_array_equal(pa, pb); np.testing.assert_array_equal(va, vb)


def test_sleeping_polyhedra_bit_exact():
    sc = S.polyhedra_clutter(16, 5).enable_sleep()
    for step, g, o, ev in _lockstep(sc, 900, every=30, sleeping=True):
        pass
    assert g.sleeping().any()


def test_sensor_polyhedron_and_fast_polyhedra():
    rng = np.random.default_rng(1)
    sc = S.Scene(name="poly_sensor_ccd", gravity=(0.0, -9.81, 0.0))
    fl = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.25, 0.0)); sc.add_collider(fl, half_extents=(100.0, 0.25, 100.0))
    zone = sc.add_convex_polyhedron((rng.standard_normal((30, 3)) * 1.2).astype(np.float32))
    chip = sc.add_convex_polyhedron((rng.standard_normal((12, 3)) * 0.06).astype(np.float32))
    gem = sc.add_convex_polyhedron((rng.standard_normal((20, 3)) * 0.3).astype(np.float32))
    z = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, 2.5, 0.0))
    sc.add_collider(z, shape=S.SHAPE_CONVEX, half_extents=(zone, 0, 0), sensor=1, active_events=S.ACTIVE_EVENTS_COLLISION)
    for k, (shape, he) in enumerate([(S.SHAPE_BALL, (0.3, 0, 0)), (S.SHAPE_CUBOID, (0.3, 0.2, 0.25)), (S.SHAPE_CONVEX, (gem, 0, 0)), (S.SHAPE_CONE, (0.3, 0.25, 0.0))]):
        b = sc.add_body(translation=(-1.2 + 0.8 * k, 6.0 + 0.8 * k, 0.2 * k), rotation=(0.2, 0.1, 0.3, 0.9273618), angvel=(1.0, 0.0, 2.0))
        sc.add_collider(b, shape=shape, half_extents=he, active_events=S.ACTIVE_EVENTS_COLLISION)
    fast = []
    for k in range(3):                                             # small fast polyhedra: the CCD pass runs GJK against the thin floor and the gem
        b = sc.add_body(translation=(6.0 + 1.5 * k, 7.0 + k, 0.1 * k), linvel=(0.0, -50.0 - 10.0 * k, 0.0), rotation=(0.3, 0.0, 0.2, 0.9327379), angvel=(3.0, 0.0, 1.0),
                        ccd_enabled=1 if k == 2 else 0)
        sc.add_collider(b, shape=S.SHAPE_CONVEX, half_extents=(chip, 0, 0), density=4.0)
        fast.append(b)
    tgt = sc.add_body(translation=(9.0, 0.5, 0.2)); sc.add_collider(tgt, shape=S.SHAPE_CONVEX, half_extents=(gem, 0, 0))
    n_sensor = 0
    for step, g, o, ev in _lockstep(sc, 200):
        n_sensor += sum(1 for e in ev if e[3] & 1)
    assert n_sensor >= 6
    pos, _ = g.read_bodies()
    assert pos[fast, 1].min() > -0.01 and g.counters()["ccd_clamp_count"] >= 2      # on the floor, not in it


def test_polyhedron_registered_in_a_running_world():
    sc = S.box_stack(4)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(30); o.step(30)
    rng = np.random.default_rng(3)
    for k in range(2):
        pts = (rng.standard_normal((16, 3)) * 0.35).astype(np.float32)
        gid, oid = g.add_convex_polyhedron(pts), o.add_convex_polyhedron(pts)
        assert gid == oid == k
        body = S.body_desc(translation=(0.1 + 1.4 * k, 6.0, 0.05), rotation=(0.1, 0.2, 0.3, 0.9273618))
        col = S.collider_desc(shape=S.SHAPE_CONVEX, half_extents=(gid, 0, 0), density=2.0)
        hb = g.insert_body(body); g.insert_collider(col, hb)
        ob = lib().ro_add_body(o._w, np.array([body], S.BODY_DTYPE).ctypes.data)
        lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
        for n in (1, 20, 100):
            g.step(n); o.step(n)
            gp, gv = g.read_bodies(); op, ov = o.read()
            np.testing.assert_array_equal(gp, op); np.testing.assert_array_equal(gv, ov)


def test_bad_polyhedra_are_refused():
    g = PhysicsWorld.from_scene(S.box_stack(1))
    with pytest.raises(RapierHipError):
        g.add_convex_polyhedron(np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0.3, 0.3, 0]]))          # flat: convex_hull returns None
    tet = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    with pytest.raises(RapierHipError):
        g.add_convex_polyhedron(tet, np.uint32([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]]))                     # wound inwards
    pid = g.add_convex_polyhedron(tet)
    hb = g.insert_body(S.body_desc(translation=(0, 5, 0)))
    g.insert_collider(S.collider_desc(shape=S.SHAPE_CONVEX, half_extents=(pid, 0, 0)), hb)
    with pytest.raises(RapierHipError):
        g.insert_collider(S.collider_desc(shape=S.SHAPE_CONVEX, half_extents=(pid + 1, 0, 0)), hb)                # no such polyhedron


def test_a_polyhedron_of_256_vertices_and_one_too_many():
    """the vertex limit of a registered polyhedron (RP_POLY_MAX_VERTS): a 256-vertex geodesic-like hull works (support scans over all of
    them), 257 hull vertices are refused"""
    def sphere(n):
        i = np.arange(n) + 0.5
        phi, th = np.arccos(1 - 2 * i / n), np.pi * (1 + 5 ** 0.5) * i
        return np.stack([np.cos(th) * np.sin(phi), np.cos(phi), np.sin(th) * np.sin(phi)], 1).astype(np.float32) * np.float32(0.5)
    sc = S.Scene(name="poly256", gravity=(0.0, -9.81, 0.0))
    gnd = sc.add_body(body_type=S.BODY_FIXED, translation=(0, -0.5, 0)); sc.add_collider(gnd, half_extents=(10, 0.5, 10))
    pid = sc.add_convex_polyhedron(sphere(256))
    for k in range(3):
        b = sc.add_body(translation=(0.2 * k, 1.0 + 1.2 * k, 0.1 * k), angvel=(1.0, 0.0, 0.5))
        sc.add_collider(b, shape=S.SHAPE_CONVEX, half_extents=(pid, 0, 0))
    for step, g, o, ev in _lockstep(sc, 240, every=8):
        pass
    assert len(g.read_convex_polyhedron(pid)["points"]) == 256
    pos, _ = g.read_bodies()
    assert pos[1:4, 1].min() > 0.45                                       # three 256-vertex "balls" of radius 0.5: on the ground (or on each other), not in it
    with pytest.raises(RapierHipError):
        g.add_convex_polyhedron(sphere(257))
