"""The reference's box3d ports that need convex hulls — examples3d/b3d_junkyard.rs (rocks sharing one hull, stirred by an orbiting
kinematic position-based cylinder hull) and b3d_washer.rs (a kinematic velocity-based ring of 40 hulls tumbling cubes) — at reduced
size on the device, in lockstep with the oracle, bit for bit; the generators against the example files' closed forms (CPU)."""
import numpy as np
import pytest

from rapier_amd import scenes as S
from oracle_ffi import OracleWorld


def test_junkyard_generator_follows_the_example_file():
    sc = S.junkyard()
    assert len(sc.bodies) == 1 + 24 * 21 * 21 + 1 and len(sc.colliders) == 5 + 24 * 21 * 21 + 1 and len(sc.polyhedra) == 2
    np.testing.assert_allclose(sc.bodies[1]["translation"], (-40.0, 25.0, -40.0)); np.testing.assert_allclose(sc.bodies[1 + 21 * 21 + 21 + 1]["translation"], (-36.0, 29.0, -36.0))
    rock, drum = sc.polyhedra[0][0], sc.polyhedra[1][0]
    assert rock.shape == (10, 3) and np.allclose(np.linalg.norm(rock, axis=1), 1.5, atol=1e-5) and abs(rock[0, 2] - 1.5 * 0.9) < 1e-6   # b3CreateRock: z = 1 - (2 i + 1) / 10
    assert drum.shape == (32, 3) and np.allclose(np.hypot(drum[:, 0], drum[:, 2]), 4.0, atol=1e-5) and set(np.round(drum[:, 1], 4)) == {0.0, 24.0}
    o = OracleWorld(S.junkyard(1, 2, 2))
    a, b = o.read_convex_polyhedron(0), o.read_convex_polyhedron(1)
    assert len(a["points"]) == 10 and len(a["face_normals"]) == 16 and len(b["points"]) == 32 and len(b["face_normals"]) == 18   # 16 triangles; 16 sides + 2 caps
    t = S.junkyard_pusher_target(30)
    np.testing.assert_allclose(t[:3], (35.0 * np.cos(np.radians(-3.0)), 0.0, 35.0 * np.sin(np.radians(-3.0))), atol=1e-4)


def test_washer_generator_follows_the_example_file():
    sc = S.washer()
    assert len(sc.bodies) == 2 + 8000 and len(sc.polyhedra) == 40 and len(sc.colliders) == 1 + 40 + 8000
    assert int(sc.bodies[1]["body_type"]) == S.BODY_KINEMATIC_VELOCITY and abs(float(sc.bodies[1]["angvel"][2]) - np.radians(25.0)) < 1e-6
    for pts, _ in sc.polyhedra:
        r = np.hypot(pts[:, 0], pts[:, 1])
        assert pts.shape == (8, 3) and set(np.round(pts[:, 2], 4)) == {-10.0, 10.0} and (np.allclose(sorted(set(np.round(r, 3))), [16.0, 18.0]) or np.allclose(sorted(set(np.round(r, 3))), [14.0, 16.0]))
    np.testing.assert_allclose(sc.bodies[2]["translation"], (-8.0, 13.0, -8.0), atol=1e-6); np.testing.assert_allclose(sc.bodies[-1]["translation"], (7.2, 28.2, 7.2), atol=1e-4)
    assert float(sc.colliders[41]["density"]) == 1000.0


@pytest.mark.gpu
def test_junkyard_reduced_bit_exact():
    from rapier_amd import PhysicsWorld
    sc = S.junkyard(4, 9, 9)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for k in range(1, 361):
        t = S.junkyard_pusher_target(k)
        g.set_next_kinematic_position([sc.pusher], t); o.set_next_kinematic_position(sc.pusher, t)
        g.step(1); o.step(1)
        if k % 20 == 0:
            gp, gv = g.read_bodies(); op, ov = o.read()
            np.testing.assert_array_equal(gp, op, err_msg=f"step {k}"); np.testing.assert_array_equal(gv, ov, err_msg=f"step {k}")
    c = g.counters()
    assert c["overflow_flags"] == 0 and c["num_manifolds"] == o.stats()["num_active_manifolds"] > 300
    assert gp[1:-1, 1].min() > 0.5                                   # every rock above the floor (its smallest extent is ~1 m)


@pytest.mark.gpu
def test_washer_reduced_bit_exact():
    from rapier_amd import PhysicsWorld
    sc = S.washer(8)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for cp in range(20, 321, 20):
        g.step(20); o.step(20)
        gp, gv = g.read_bodies(); op, ov = o.read()
        np.testing.assert_array_equal(gp, op, err_msg=f"step {cp}"); np.testing.assert_array_equal(gv, ov, err_msg=f"step {cp}")
    c = g.counters()
    assert c["overflow_flags"] == 0 and c["num_manifolds"] == o.stats()["num_active_manifolds"] > 300
    assert np.hypot(gp[2:, 0], gp[2:, 1] - gp[1, 1]).max() < 18.5      # the cubes are still inside the ring
