"""Batches of small worlds (rp_world_begin_subworld / ro_begin_subworld): sub-worlds of one world that overlap in space and never pair.
Oracle side (CPU): n copies of a scene as one batch evolve, each, bit for bit like the scene stepped alone; a mixed batch keeps its
sub-worlds apart (pair count = the sum of the singles) — the device twin is tests/test_gpu_subworlds.py."""
import numpy as np

from rapier_amd import scenes as S
from oracle_ffi import OracleWorld


def _single(sc, steps):
    w = OracleWorld(sc); w.step(steps)
    return w.read(), w.stats()


def test_copies_in_one_batch_equal_the_scene_stepped_alone():
    sc = S.capsules(6)
    (pos, vel), st = _single(sc, 90)
    n = 5
    b = S.batch([S.capsules(6) for _ in range(n)])
    assert len(b.bodies) == n * len(sc.bodies) and b.subworlds[2] == (2 * len(sc.bodies), 2 * len(sc.colliders), 0)
    w = OracleWorld(b); w.step(90)
    bp, bv = w.read()
    nb = len(sc.bodies)
    for k in range(n):
        np.testing.assert_array_equal(bp[k * nb:(k + 1) * nb], pos, err_msg=f"sub-world {k} poses")
        np.testing.assert_array_equal(bv[k * nb:(k + 1) * nb], vel, err_msg=f"sub-world {k} velocities")
    bst = w.stats()
    assert bst["num_pairs"] == n * st["num_pairs"] and bst["num_active_manifolds"] == n * st["num_active_manifolds"]


def test_a_mixed_batch_keeps_its_sub_worlds_apart():
    parts = [S.box_stack(3), S.pyramid10(), S.joint_chain(4, with_boxes=True), S.box_stack(2, gap=0.5)]
    for p in parts:                                  # one parameter set and one gravity for the whole batch
        p.gravity, p.params = parts[0].gravity, parts[0].params.copy()
    singles = [_single(p, 60) for p in parts]
    w = OracleWorld(S.batch(parts)); w.step(60)
    bp, bv = w.read()
    assert w.stats()["num_pairs"] == sum(st["num_pairs"] for _, st in singles)
    assert np.isfinite(bp).all()
    off = 0
    for (pos, vel), _ in singles:                     # same physics (the sweep order of a batch is the batch's: bits may differ)
        assert np.abs(bp[off:off + len(pos), :3] - pos[:, :3]).max() < 1e-2
        off += len(pos)


def test_batch_refuses_mismatched_parameters():
    a, b = S.box_stack(2), S.box_stack(2)
    b.params = b.params.copy(); b.params["dt"] = 0.01
    try:
        S.batch([a, b])
    except ValueError:
        return
    raise AssertionError("a batch of scenes with different dt must be refused")


def test_a_batch_carries_registered_shapes_with_their_ids_moved():
    """convex polyhedra and composite shapes are world-wide tables: in a batch the ids a collider names move behind the ones the
    earlier sub-worlds registered — every sub-world still evolves like its scene alone (4 copies + a mesh world: no colour crosses the
    >= 32-chunk line)"""
    from test_composite_oracle import _box_on_mesh
    mesh_world, _ = _box_on_mesh()
    clutter = S.polyhedra_clutter(6, 2)
    parts = [clutter, mesh_world, S.polyhedra_clutter(6, 2)]
    for p in parts:
        p.gravity, p.params = parts[0].gravity, parts[0].params.copy()
    b = S.batch(parts)
    assert len(b.polyhedra) == 2 * len(clutter.polyhedra) and len(b.composites) == 1
    w = OracleWorld(b); w.step(120)
    bp, bv = w.read()
    off = 0
    for p in parts:
        o = OracleWorld(p); o.step(120)
        pos, vel = o.read()
        np.testing.assert_array_equal(bp[off:off + len(pos)], pos, err_msg=p.name); np.testing.assert_array_equal(bv[off:off + len(pos)], vel)
        off += len(pos)
