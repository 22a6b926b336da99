"""Host-side sharding logic (rapier_amd/sharding.py) on CPU: what a shard carries and how bodies are boxed — the cases ADVICE r4 named."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rapier_amd import scenes as S, sharding  # noqa: E402


def _two_piles():
    """two far-apart piles of support-mapped shapes over a slab + a jointed pair in the second pile"""
    s = S.Scene(name="two_piles", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, shape=S.SHAPE_CUBOID, half_extents=(40.0, 0.5, 4.0))
    pts = np.array([[-0.3, -0.3, -0.3], [0.3, -0.3, -0.3], [0.0, 0.4, 0.0], [0.0, -0.3, 0.5]], np.float32)
    pid = s.add_convex_polyhedron(pts)
    shapes = [dict(shape=S.SHAPE_CYLINDER, half_extents=(0.4, 0.3, 0.0)), dict(shape=S.SHAPE_CONE, half_extents=(0.4, 0.3, 0.0)),
              dict(shape=S.SHAPE_CONVEX_POLYHEDRON, half_extents=(pid, 0, 0)), dict(shape=S.SHAPE_ROUND_CUBOID, half_extents=(0.2, 0.2, 0.2), border_radius=0.05)]
    ids = []
    for pile, x0 in enumerate((-20.0, 20.0)):
        for k, kw in enumerate(shapes):
            b = s.add_body(translation=(x0 + 0.9 * k, 0.6, 0.0))
            s.add_collider(b, **kw)
            ids.append(b)
    return s, ids


def test_every_shape_gets_a_finite_box_and_a_rank():
    s, ids = _two_piles()
    lo, hi = sharding.body_boxes(s)
    assert np.all(np.isfinite(lo[ids])) and np.all(np.isfinite(hi[ids]))
    # the round cuboid's box holds its border: half diagonal + border radius
    rc = ids[3]
    assert hi[rc][0] - lo[rc][0] >= 2 * (np.sqrt(3 * 0.2 ** 2) + 0.05) - 1e-6
    groups = sharding.proximity_groups_from_scene(s)
    body_rank, n_groups = sharding.shards_from_groups(groups, 2)
    assert n_groups == 2 and all(body_rank[i] >= 0 for i in ids)
    assert len({int(body_rank[i]) for i in ids[:4]}) == 1 and len({int(body_rank[i]) for i in ids[4:]}) == 1
    assert body_rank[ids[0]] != body_rank[ids[4]]


def test_a_shard_carries_the_registered_polyhedra_and_steps_like_the_whole_world():
    from oracle_ffi import OracleWorld
    s, ids = _two_piles()
    groups = sharding.proximity_groups_from_scene(s)
    body_rank, _ = sharding.shards_from_groups(groups, 2)
    whole = OracleWorld(s); whole.step(40)
    wp, wv = whole.read()
    for rank in (0, 1):
        sub, gids = sharding.partition_scene(s, body_rank, rank)
        assert len(sub.polyhedra) == len(s.polyhedra)
        w = OracleWorld(sub); w.step(40)
        p, v = w.read()
        assert np.array_equal(p, wp[gids]) and np.array_equal(v, wv[gids])


def test_a_dynamic_body_without_a_shard_is_an_error():
    s, ids = _two_piles()
    groups = sharding.proximity_groups_from_scene(s)
    body_rank, _ = sharding.shards_from_groups(groups, 2)
    body_rank[ids[2]] = -1
    with pytest.raises(ValueError):
        sharding.partition_scene(s, body_rank, 0)


def test_shards_carry_composite_shapes_and_step_like_the_whole_world():
    """compounds, a triangle mesh and a height field (one collider each, round 5): every shard keeps the registered composites under
    their ids, compound bodies get finite boxes (the farthest part), and each shard evolves bit for bit like its bodies in the whole world"""
    from oracle_ffi import OracleWorld
    s = S.Scene(name="two_piles_composite", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=S.BODY_FIXED)
    v = np.array([[-30, 0, -4], [0, 0, -4], [0, 0, 4], [-30, 0, 4]], np.float32)
    s.add_collider(g, shape=S.SHAPE_TRIMESH, half_extents=(s.add_trimesh(v, np.array([[0, 2, 1], [0, 3, 2]], np.uint32)), 0, 0))
    g2 = s.add_body(body_type=S.BODY_FIXED, translation=(15.0, 0.0, 0.0))
    s.add_collider(g2, shape=S.SHAPE_TRIMESH, half_extents=(s.add_heightfield(np.zeros((5, 5), np.float32), (30.0, 1.0, 8.0)), 0, 0))
    ids = []
    for x0 in (-20.0, 20.0):
        for k in range(3):
            cid = s.add_compound([S.collider_desc(half_extents=(0.4, 0.1, 0.1)), S.collider_desc(half_extents=(0.1, 0.4, 0.1), translation=(0.4, 0.4, 0.0)),
                                  S.collider_desc(shape=S.SHAPE_BALL, half_extents=(0.15, 0, 0), translation=(-0.4, 0.3, 0.0))])
            b = s.add_body(translation=(x0 + 1.2 * k, 0.8 + 0.3 * k, 0.1 * k), can_sleep=1)
            s.add_collider(b, shape=S.SHAPE_COMPOUND, half_extents=(cid, 0, 0))
            ids.append(b)
    lo, hi = sharding.body_boxes(s)
    assert np.all(np.isfinite(lo[ids])) and np.all(np.isfinite(hi[ids]))
    assert np.all(hi[ids, 0] - lo[ids, 0] >= 2 * (np.hypot(0.4, 0.4) + np.sqrt(0.1 ** 2 + 0.4 ** 2 + 0.1 ** 2)) - 1e-6)   # the farthest part, its own radius included
    assert hi[g][0] - lo[g][0] >= 60.0 and hi[g2][0] - lo[g2][0] >= 30.0                                                       # the mesh and the height field
    groups = sharding.proximity_groups_from_scene(s)
    body_rank, n_groups = sharding.shards_from_groups(groups, 2)
    assert body_rank[ids[0]] != body_rank[ids[3]]
    whole = OracleWorld(s); whole.step(60)
    wp, wv = whole.read()
    for rank in (0, 1):
        sub, gids = sharding.partition_scene(s, body_rank, rank)
        assert len(sub.composites) == len(s.composites)
        w = OracleWorld(sub); w.step(60)
        p, v6 = w.read()
        assert np.array_equal(p, wp[gids]) and np.array_equal(v6, wv[gids])


def test_a_batch_of_sub_worlds_is_not_cut_by_body_rank():
    b = S.batch([S.capsules(2), S.capsules(2)])
    with pytest.raises(ValueError, match="sub-worlds"):
        sharding.partition_scene(b, np.zeros(len(b.bodies), np.int32), 0)
