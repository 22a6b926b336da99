"""The oracle's support-mapped shapes (oracle/ro_convex.h: cylinders and cones through GJK / EPA + polygonal feature maps) against
independent checks.  parry3d — where these algorithms live for the reference — is not under /root/reference, so there is no golden
manifold to pin; what CAN be pinned:

  * distances and penetration depths against a brute-force minimisation of the support function of the configuration-space
    obstacle (numpy, 200,000 directions + local refinement: no code shared with the oracle);
  * point projections against dense samplings of the surfaces;
  * mass properties against the closed forms (response to impulses);
  * rest heights of every shape on every ground;
  * the reference's own regression test for these shapes, crates/rapier3d/tests/issue_810_cubes_thin_cylinder_tunnel.rs;
  * the shape of the manifolds the feature maps produce (a cap is a square: four points; the curved part a segment: two)."""
import ctypes as C

import numpy as np
import pytest

from rapier_amd import scenes as S
from oracle_ffi import OracleWorld
import oracle_ffi

F3, F7, F10 = C.c_float * 3, C.c_float * 7, C.c_float * 10


def _contact(sh1, he1, sh2, he2, pos, pred=10.0):
    out = F10()
    oracle_ffi.lib().ro_kat_convex_contact(C.c_int32(sh1), F3(*he1), C.c_int32(sh2), F3(*he2), F7(*pos), C.c_float(pred), out)
    o = np.array(out[:], np.float64)
    return int(o[0]), o[1:4], o[4:7], o[7:10]


def _manifold(sh1, he1, sh2, he2, pos, pred=0.002):
    L = oracle_ffi.lib()
    L.ro_kat_convex_manifold.restype = C.c_int32
    pts, n1 = (C.c_float * 72)(), F3()
    n = L.ro_kat_convex_manifold(C.c_int32(sh1), F3(*he1), C.c_int32(sh2), F3(*he2), F7(*pos), C.c_float(pred), pts, n1)
    return np.array(pts[: 9 * n], np.float64).reshape(n, 9), np.array(n1[:], np.float64)


# ---- support functions of the core shapes, written from the shapes' definitions ----
def _h(sh, he):
    if sh == S.SHAPE_CUBOID:
        return lambda d: np.abs(d) @ np.array(he)
    if sh == S.SHAPE_CYLINDER:
        return lambda d: he[1] * np.hypot(d[:, 0], d[:, 2]) + he[0] * np.abs(d[:, 1])
    if sh == S.SHAPE_CONE:
        return lambda d: np.maximum(he[0] * d[:, 1], he[1] * np.hypot(d[:, 0], d[:, 2]) - he[0] * d[:, 1])
    return lambda d: he[0] * np.abs(d[:, int(he[2])])     # a capsule's segment


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _fibonacci(n):
    i = np.arange(n) + 0.5
    phi, th = np.arccos(1 - 2 * i / n), np.pi * (1 + 5 ** 0.5) * i
    return np.stack([np.cos(th) * np.sin(phi), np.cos(phi), np.sin(th) * np.sin(phi)], 1)


_DIRS = _fibonacci(200000)


def _cso_width(sh1, he1, sh2, he2, pos):
    t, q = np.array(pos[:3]), np.array(pos[3:]) / np.linalg.norm(pos[3:])
    R, h1, h2 = _rot(q), _h(sh1, he1), _h(sh2, he2)
    return lambda d: h1(d) + h2((-d) @ R) - d @ t   # support function of shape1 - shape2 along unit d


def _brute(sh1, he1, sh2, he2, pos, seed=0):
    """min over unit d of the CSO's support function: > 0 = penetration depth, < 0 = minus the distance; and the minimiser"""
    f = _cso_width(sh1, he1, sh2, he2, pos)
    v = f(_DIRS)
    k = int(np.argmin(v))
    best, bv, rng, rad = _DIRS[k], v[k], np.random.default_rng(seed), 0.02
    for _ in range(40):
        c = best + rad * rng.standard_normal((4000, 3))
        c /= np.linalg.norm(c, axis=1)[:, None]
        v = f(c)
        k = int(np.argmin(v))
        if v[k] < bv:
            bv, best = v[k], c[k]
        rad *= 0.7
    return bv, best


def _random_pair(rng):
    kinds = [(S.SHAPE_CUBOID, lambda: tuple(rng.uniform(.2, 1, 3))), (S.SHAPE_CYLINDER, lambda: (rng.uniform(.1, 1), rng.uniform(.1, 1), 0.0)),
             (S.SHAPE_CONE, lambda: (rng.uniform(.2, 1), rng.uniform(.2, 1), 0.0)), (S.SHAPE_CAPSULE, lambda: (rng.uniform(.2, 1), 0.0, float(rng.integers(0, 3))))]
    while True:
        a, b = kinds[rng.integers(0, 4)], kinds[rng.integers(0, 4)]
        if a[0] in (S.SHAPE_CYLINDER, S.SHAPE_CONE) or b[0] in (S.SHAPE_CYLINDER, S.SHAPE_CONE):
            return a[0], a[1](), b[0], b[1]()


def _check(sh1, he1, sh2, he2, pos, f, d, tag):
    hit, p1, p2, n = _contact(sh1, he1, sh2, he2, pos)
    assert hit, (tag, "no answer", f)
    dist = (p2 - p1) @ n
    assert abs(np.linalg.norm(n) - 1.0) < 1e-4, tag
    if f < -1e-4:     # separated: the distance, and the direction where it is well defined
        assert abs(dist - (-f)) < 2e-3 * max(1.0, -f) + 3e-4, (tag, dist, -f)
        if -f > 0.02:
            assert n @ d > np.cos(np.radians(3.0)), (tag, n, d)
    elif f > 1e-4:    # overlapping: the polytope's answer is a depth along ITS normal that the CSO really has (a face on the boundary) ...
        w = _cso_width(sh1, he1, sh2, he2, pos)(n[None, :])[0]
        assert abs(w - (-dist)) < 2e-4 + 2e-2 * f, (tag, "the face is not on the boundary", w, dist)
        assert -dist < f + 2e-4 + 1e-2 * f, (tag, "a smaller depth exists", dist, f)   # ... and no sampled direction does better


def test_gjk_epa_against_support_function_minimisation():
    rng = np.random.default_rng(5)
    for case in range(60):
        sh1, he1, sh2, he2 = _random_pair(rng)
        q = rng.standard_normal(4)
        t = rng.standard_normal(3)
        t *= rng.uniform(0, 2.2) / np.linalg.norm(t)
        pos = list(t) + list(q / np.linalg.norm(q))
        f, d = _brute(sh1, he1, sh2, he2, pos)
        _check(sh1, he1, sh2, he2, pos, f, d, ("random", case, sh1, sh2))


def test_gjk_epa_on_resting_contact_depths():
    """the configurations a simulation lives in: shapes that overlap by a fraction of a millimetre or sit that far apart"""
    rng = np.random.default_rng(6)
    for case in range(60):
        sh1, he1, sh2, he2 = _random_pair(rng)
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        t = rng.standard_normal(3)
        t *= rng.uniform(0, 2.2) / np.linalg.norm(t)
        f, d = _brute(sh1, he1, sh2, he2, list(t) + list(q))
        target = rng.choice([0.001, 0.0005, -0.001, 0.003])
        for _ in range(3):      # slide shape 2 along the minimising direction until the width is `target`
            t = t + (f - target) * d
            f, d = _brute(sh1, he1, sh2, he2, list(t) + list(q))
        _check(sh1, he1, sh2, he2, list(t) + list(q), f, d, ("shallow", case, sh1, sh2, target))


@pytest.mark.parametrize("shape", [S.SHAPE_CYLINDER, S.SHAPE_CONE])
def test_point_projection_against_a_sampled_surface(shape):
    hh, r = 0.7, 0.4
    u = np.linspace(0, 2 * np.pi, 720, endpoint=False)
    ring = np.stack([np.cos(u), np.zeros_like(u), np.sin(u)], 1)
    surf = []
    for rad in np.linspace(0, r, 60):            # base (and top cap of the cylinder)
        surf.append(ring * rad + [0, -hh, 0])
        if shape == S.SHAPE_CYLINDER:
            surf.append(ring * rad + [0, hh, 0])
    for y in np.linspace(-hh, hh, 200):          # the curved part
        rad = r if shape == S.SHAPE_CYLINDER else r * (hh - y) / (2 * hh)
        surf.append(ring * rad + [0, y, 0])
    surf = np.concatenate(surf)
    rng = np.random.default_rng(1)
    out = (C.c_float * 4)()
    for _ in range(200):
        pt = rng.uniform(-1.2, 1.2, 3)
        oracle_ffi.lib().ro_kat_convex_project(C.c_int32(shape), F3(hh, r, 0), F3(*pt), out)
        proj, inside = np.array(out[:3]), bool(out[3])
        d_sampled = np.sqrt(((surf - pt) ** 2).sum(1).min())
        assert abs(np.linalg.norm(proj - pt) - d_sampled) < 6e-3, (pt, proj, d_sampled)
        rad_at = r if shape == S.SHAPE_CYLINDER else r * (hh - pt[1]) / (2 * hh)
        assert inside == (abs(pt[1]) <= hh and np.hypot(pt[0], pt[2]) <= rad_at)


def _lone_body(shape, he, density=1.0):
    sc = S.Scene(name="mass", gravity=(0.0, 0.0, 0.0))
    b = sc.add_body(translation=(0, 0, 0))
    sc.add_collider(b, shape=shape, half_extents=he, density=density)
    return sc


def test_mass_properties_of_cylinder_and_cone_closed_forms():
    hh, r, rho = 0.6, 0.35, 1.7
    for shape, vol, iy, ixz, com_y in (
            (S.SHAPE_CYLINDER, np.pi * r * r * 2 * hh, r * r / 2, (3 * r * r + 4 * hh * hh) / 12, 0.0),
            (S.SHAPE_CONE, np.pi * r * r * 2 * hh / 3, 3 * r * r / 10, 3 * r * r / 20 + 3 * (2 * hh) ** 2 / 80, -hh / 2)):
        m = vol * rho
        o = OracleWorld(_lone_body(shape, (hh, r, 0), rho))
        o.apply_impulse(0, impulse=(1.0, 0.0, 0.0))
        _, v = o.read()
        assert abs(v[0][0] - 1.0 / m) < 1e-5 / m
        # the centre of mass (the cone's: a quarter of the height above the base): a spinning body's origin circles it
        sc = _lone_body(shape, (hh, r, 0), rho)
        sc.bodies[0]["angvel"] = (0.0, 0.0, 1.0)
        o = OracleWorld(sc)
        o.step(1)
        p, _ = o.read()
        np.testing.assert_allclose(p[0][0], com_y * float(sc.params["dt"]), atol=2e-5)
        o = OracleWorld(_lone_body(shape, (hh, r, 0), rho))
        o.apply_impulse(0, torque_impulse=(0.0, 1.0, 0.0))
        _, v = o.read()
        np.testing.assert_allclose(v[0][4], 1.0 / (iy * m), rtol=2e-5)
        o = OracleWorld(_lone_body(shape, (hh, r, 0), rho))
        o.apply_impulse(0, torque_impulse=(1.0, 0.0, 0.0))
        _, v = o.read()
        np.testing.assert_allclose(v[0][3], 1.0 / (ixz * m), rtol=2e-5)


_S2 = float(np.sin(np.pi / 4))


@pytest.mark.parametrize("ground", ["cuboid", "halfspace", "cylinder"])
@pytest.mark.parametrize("shape,rot,rest", [(S.SHAPE_CYLINDER, (0, 0, 0, 1), 0.5), (S.SHAPE_CYLINDER, (0, 0, _S2, _S2), 0.3), (S.SHAPE_CONE, (0, 0, 0, 1), 0.5)])
def test_rest_heights(ground, shape, rot, rest):
    sc = S.Scene(name="rest", gravity=(0.0, -9.81, 0.0))
    g = sc.add_body(body_type=S.BODY_FIXED, translation=(0, -0.5, 0))
    if ground == "cuboid":
        sc.add_collider(g, half_extents=(10, 0.5, 10))
    elif ground == "cylinder":
        sc.add_collider(g, shape=S.SHAPE_CYLINDER, half_extents=(0.5, 10, 0))
    else:
        sc.add_collider(g, shape=S.SHAPE_HALFSPACE, half_extents=(0, 1, 0), translation=(0, 0.5, 0))
    b = sc.add_body(translation=(0.3, 1.2, 0.2), rotation=rot)
    sc.add_collider(b, shape=shape, half_extents=(0.5, 0.3, 0))
    o = OracleWorld(sc)
    o.step(300)
    p, v = o.read()
    assert abs(p[b][1] - rest) < 2.5e-3, p[b]          # allowed_linear_error = 1 mm of penetration
    assert abs(v[b][1]) < 1e-2


def test_issue_810_cubes_do_not_fall_through_a_thin_cylinder_disc():
    """crates/rapier3d/tests/issue_810_cubes_thin_cylinder_tunnel.rs:61-119, asserts restated"""
    sc = S.issue_810_disc()
    o = OracleWorld(sc)
    radius, disc_top = 10.0, -1.95
    for i in range(600):
        o.step(1)
        p, _ = o.read()
        for k in range(20):
            pos = p[1 + k]
            if pos[1] < disc_top - 0.5:
                assert np.hypot(pos[0], pos[2]) > radius - 0.2, f"cube {k} tunneled through the disc at step {i}: {pos}"
    on_disc = sum(abs(p[1 + k][1] - disc_top) < 0.2 for k in range(20))
    assert on_disc >= 15, on_disc
    assert o.stats()["num_solver_contacts"] == 4 * on_disc        # the point of the fix: a full face manifold on the cap, not one point


def test_feature_maps_give_square_caps_and_segment_sides():
    box, cyl, cone = (2, .5, 2), (.5, .3, 0), (.5, .3, 0)
    pts, n1 = _manifold(S.SHAPE_CUBOID, box, S.SHAPE_CYLINDER, cyl, (0.3, 0.999, 0.2, 0, 0, 0, 1))
    assert len(pts) == 4 and np.allclose(n1, (0, 1, 0), atol=1e-5) and np.allclose(pts[:, 6], -0.001, atol=2e-5)
    assert sorted(pts[:, 8]) == [1, 3, 5, 7] and np.allclose(np.hypot(pts[:, 3], pts[:, 5]), 0.3, atol=1e-5)       # bottom cap's square, on the rim
    pts, n1 = _manifold(S.SHAPE_CUBOID, box, S.SHAPE_CYLINDER, cyl, (0.3, 0.799, 0.2, 0, 0, _S2, _S2))
    assert len(pts) == 2 and sorted(pts[:, 8]) == [1, 11] and np.allclose(pts[:, 6], -0.001, atol=2e-5)            # lying: the curved part's segment
    pts, n1 = _manifold(S.SHAPE_CUBOID, box, S.SHAPE_CONE, cone, (0.3, 0.999, 0.2, 0, 0, 0, 1))
    assert len(pts) == 4 and sorted(pts[:, 8]) == [1, 3, 5, 7]
    pts, n1 = _manifold(S.SHAPE_CUBOID, box, S.SHAPE_CONE, cone, (0.3, 0.999, 0.2, 1, 0, 0, 0))                       # apex down: the side segment, apex touching
    assert len(pts) == 2 and abs(pts[:, 6].min() + 0.001) < 2e-5
    # a small box anywhere on a wide cap gets its whole face (the cap's square turns toward the contact)
    pts, n1 = _manifold(S.SHAPE_CYLINDER, (.05, 10, 0), S.SHAPE_CUBOID, (.05, .05, .05), (6.0, 0.099, -5.0, 0, 0, 0, 1))
    assert len(pts) == 4 and np.allclose(pts[:, 6], -0.001, atol=2e-5) and set(pts[:, 7]) == {19.0}
    # two cylinders side by side, axes parallel: the two segments clip to two points
    pts, n1 = _manifold(S.SHAPE_CYLINDER, cyl, S.SHAPE_CYLINDER, cyl, (0.599, 0.2, 0.0, 0, 0, 0, 1))
    assert len(pts) == 2 and abs(n1[0]) > 0.999 and np.allclose(pts[:, 6], -0.001, atol=5e-5)
    # crossed cylinders: one point, normal along the line of centres
    pts, n1 = _manifold(S.SHAPE_CYLINDER, cyl, S.SHAPE_CYLINDER, cyl, (0.599, 0.0, 0.0, _S2, 0, 0, _S2))
    assert len(pts) == 1 and n1[0] > 0.99 and abs(pts[0, 6] + 0.001) < 1e-4


@pytest.mark.parametrize("ground", ["cuboid", "cylinder", "halfspace"])
def test_clutter_settles_and_is_reproducible(ground):
    sc = S.convex_clutter(40, 3, ground)
    a, b = OracleWorld(sc), OracleWorld(sc)
    a.step(400); b.step(400)
    pa, va = a.read(); pb, vb = b.read()
    np.testing.assert_array_equal(pa, pb); np.testing.assert_array_equal(va, vb)
    dyn = [i for i, d in enumerate(sc.bodies) if int(d["body_type"]) == S.BODY_DYNAMIC]
    assert np.isfinite(pa).all() and pa[dyn, 1].min() > 0.1 and pa[dyn, 1].max() < 3.0       # nothing fell through, nothing was launched
    assert np.abs(pa[dyn][:, [0, 2]]).max() < 4.5                                             # everything is still inside the walls


@pytest.mark.parametrize("ground", ["cuboid", "halfspace", "round_cuboid"])
def test_round_shapes_rest_on_their_border(ground):
    """parry RoundShape<S>: the inner shape dilated by border_radius — ColliderBuilder::round_cuboid / round_cylinder / round_cone"""
    sc = S.Scene(name="round_rest", gravity=(0.0, -9.81, 0.0))
    g = sc.add_body(body_type=S.BODY_FIXED, translation=(0, -0.5, 0))
    if ground == "cuboid":
        sc.add_collider(g, half_extents=(20, 0.5, 5))
    elif ground == "round_cuboid":
        sc.add_collider(g, shape=S.SHAPE_ROUND_CUBOID, half_extents=(20, 0.4, 5), border_radius=0.1)
    else:
        sc.add_collider(g, shape=S.SHAPE_HALFSPACE, half_extents=(0, 1, 0), translation=(0, 0.5, 0))
    want = []
    for k, (sh, he, rot, rest) in enumerate([(S.SHAPE_ROUND_CUBOID, (0.3, 0.2, 0.25), (0, 0, 0, 1), 0.25), (S.SHAPE_ROUND_CYLINDER, (0.4, 0.3, 0), (0, 0, 0, 1), 0.45),
                                             (S.SHAPE_ROUND_CYLINDER, (0.4, 0.3, 0), (0, 0, _S2, _S2), 0.35), (S.SHAPE_ROUND_CONE, (0.4, 0.3, 0), (0, 0, 0, 1), 0.45),
                                             (S.SHAPE_BALL, (0.3, 0, 0), (0, 0, 0, 1), 0.3)]):
        b = sc.add_body(translation=(3.0 * k - 6, 1.2, 0), rotation=rot)
        sc.add_collider(b, shape=sh, half_extents=he, border_radius=0.05 if sh >= S.SHAPE_ROUND_CUBOID else 0.0)
        want.append((b, rest))
    o = OracleWorld(sc)
    o.step(1)
    _, v = o.read()
    o2 = OracleWorld(_lone_body(S.SHAPE_CUBOID, (0.3, 0.2, 0.25)))        # RoundShape::mass_properties = the inner shape's
    o2.apply_impulse(0, impulse=(1.0, 0.0, 0.0)); o.apply_impulse(want[0][0], impulse=(1.0, 0.0, 0.0))
    assert abs(o2.read()[1][0][0] - (o.read()[1][want[0][0]][0] - v[want[0][0]][0])) < 1e-5
    o.step(300)
    p, v = o.read()
    for b, rest in want:
        assert abs(p[b][1] - rest) < 2.5e-3 and abs(v[b][1]) < 1e-2, (b, p[b], rest)


def test_round_clutter_settles_and_is_reproducible():
    sc = S.round_clutter(30, 6)
    a, b = OracleWorld(sc), OracleWorld(sc)
    a.step(400); b.step(400)
    pa, _ = a.read(); pb, _ = b.read()
    np.testing.assert_array_equal(pa, pb)
    dyn = [i for i, d in enumerate(sc.bodies) if int(d["body_type"]) == S.BODY_DYNAMIC]
    assert np.isfinite(pa).all() and pa[dyn, 1].min() > 0.1 and pa[dyn, 1].max() < 3.0 and np.abs(pa[dyn][:, [0, 2]]).max() < 4.5
