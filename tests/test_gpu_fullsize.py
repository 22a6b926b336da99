"""Full-size parity on the very paths the bench numbers are quoted on: BASELINE configs C2, C5 and one
C4 shard at their stated sizes, long enough to run on LDS tiles + lean step graphs (C2, C5) or on replayed fused fast steps (the C4
shard), compared with the oracle bit for bit.  The reference's outcome model for the staged solver is
src/pipeline/physics_pipeline/test_staged.rs:86-148 (the same world stepped by two solver back ends must agree); here the second back
end is the CPU restatement.  Round 6: C2 and C5 run the north star's 1000 steps, all of C4 runs 120 steps on one
GPU (dense island kernel, replays, full steps).  Slow tests: the oracle does ~26 steps/s on C2, ~90 on C5 and ~6 on all of C4 with 16
threads, so C2 takes ~45 s, C5 ~15 s, the whole C4 world ~40 s incl. its build."""
import os

import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S, sharding
from oracle_ffi import OracleWorld
import oracle_ffi

pytestmark = pytest.mark.gpu


def _threads():
    oracle_ffi.set_threads(max(1, min(os.cpu_count() or 1, 32)))


def _equal(g, o, msg):
    gp, gv = g.read_bodies(); op, ov = o.read()
    assert np.isfinite(gp).all() and np.isfinite(gv).all()
    np.testing.assert_array_equal(gp, op, err_msg=msg + ": poses")
    np.testing.assert_array_equal(gv, ov, err_msg=msg + ": velocities")


def test_c2_large_pyramid_full_size_1000_steps_on_tiles_and_lean_graphs():
    """b3d_large_pyramid base 200 (20,100 cuboids, one island) at the north star's horizon: 1000 steps, six checkpoints.  The settled pile runs one launch per sweep over ~240 LDS
    tiles, on lean step graphs between layout changes; lean steps that die behind their collision stage are resumed by the full graph."""
    _threads()
    try:
        sc = S.large_pyramid(200)
        g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
        done, tiled, lean_seen = 0, 0, 0
        for cp in (10, 60, 150, 300, 600, 1000):
            g.step(cp - done); o.step(cp - done); done = cp
            _equal(g, o, f"C2 base 200 @ step {cp}")
            c = g.counters()
            tiled += 1 if (c["num_tiles"] > 0 and c["tile_sweeps"] == 1) else 0
            lean_seen = max(lean_seen, c["lean_steps"])
    finally:
        oracle_ffi.set_threads(1)
    assert c["overflow_flags"] == 0 and c["quarantined"] == 0 and c["num_dynamic_bodies"] == 20100, c
    assert tiled == 6 and c["num_tiles"] >= 200, c                      # every checkpoint saw the sweeps on tiles
    assert lean_seen > 0, c                                             # lean graphs were enqueued ...
    assert c["tile_step_steps"] > 900 and c["tile_step_steps"] <= c["lean_steps"] + c["full_steps"] and c["joint_net_disabled"] == 0, c   # ... lean and full graphs in the one-launch form (k_tile_step)
    assert c["num_manifolds"] == o.stats()["num_active_manifolds"]
    print("C2 counters:", {k: c[k] for k in ("num_tiles", "lean_steps", "tile_step_steps", "replayed_steps", "full_steps", "fast_steps")})


def test_c5_joint_grid_full_size_1000_steps_on_tiles_and_lean_graphs():
    """b3d_joint_grid 100 x 100 (10,000 balls, 19,800 spherical joints) at the north star's horizon: 1000 steps, five checkpoints — joint stages on tiles, the rows rebuilt inside the
    first biased sweep of every substep (DevWorld::joints_spherical), lean graphs once the tiling stands; joint impulses included."""
    _threads()
    try:
        sc = S.joint_grid(100)
        g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
        done = 0
        for cp in (5, 60, 300, 600, 1000):
            g.step(cp - done); o.step(cp - done); done = cp
            _equal(g, o, f"C5 100 x 100 @ step {cp}")
            gc, gi = g.read_joints(); oc, oi = o.read_joints()
            np.testing.assert_array_equal(gc, oc, err_msg=f"joint colours @ {cp}")
            np.testing.assert_array_equal(gi, oi, err_msg=f"joint impulses @ {cp}")
        c = g.counters()
    finally:
        oracle_ffi.set_threads(1)
    assert c["overflow_flags"] == 0 and c["num_tiles"] > 100 and c["tile_sweeps"] == 1, c
    assert c["lean_steps"] > 700 and c["joint_net_steps"] > 700, c  # (round 6: the lean graphs of this world are the joint-net launch, k_joint_net_step)


def test_c4_one_shard_of_eight_guarded_120_steps():
    """BASELINE config C4 (54 x 54 pyramids) as ONE of its eight shards sees it: 365 islands (more than the 240 co-resident workgroups
    of the fused step: two passes), the shard guard armed with the other shards' boxes, 120 steps incl. fused fast steps, against the
    oracle stepping the same sub-scene."""
    _threads()
    try:
        full = S.many_pyramids(54, 54)
        body_rank = sharding.many_pyramids_body_ranks(54, 54, 10, 8)
        groups = np.full(len(full.bodies), -1, np.int64)
        groups[1:] = np.repeat(np.arange(54 * 54), 55)
        sub, gids = sharding.partition_scene(full, body_rank, 0)
        g, o = PhysicsWorld.from_scene(sub), OracleWorld(sub)
        g.set_shard_guard(*sharding.guard_boxes(full, groups, body_rank, 0))
        done = 0
        for cp in (1, 30, 120):
            g.step(cp - done); o.step(cp - done); done = cp
            _equal(g, o, f"C4 shard 0 of 8 @ step {cp}")
        c = g.counters()
    finally:
        oracle_ffi.set_threads(1)
    n_isl = int((body_rank[1::55] == 0).sum())
    assert n_isl in (364, 365) and c["num_dynamic_bodies"] == n_isl * 55 and c["num_manifolds"] == n_isl * 145, c
    assert c["overflow_flags"] == 0 and c["fast_steps"] > 60, c


def test_c4_whole_world_on_one_gpu_120_steps_dense_kernel_replays_and_full_steps():
    """BASELINE config C4 whole (54 x 54 pyramids = 160,380 cuboids, 2,916 islands) on ONE GPU for 120 steps, four checkpoints: the
    register-lean two-islands-per-CU kernel in several passes, fused fast steps that give up and are replayed, and the full graph in
    between — the step mix `bench.py --workload c4` runs."""
    _threads()
    try:
        sc = S.many_pyramids(54, 54)
        g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
        done = 0
        for cp in (1, 30, 60, 120):
            g.step(cp - done); o.step(cp - done); done = cp
            _equal(g, o, f"C4 whole @ step {cp}")
        c = g.counters()
    finally:
        oracle_ffi.set_threads(1)
    assert c["num_dynamic_bodies"] == 160380 and c["num_manifolds"] == 2916 * 145 and c["overflow_flags"] == 0, c
    assert c["fast_steps"] > 30, c
    print("C4 counters:", {k: c[k] for k in ("fast_steps", "fused_steps", "replayed_steps", "full_steps")})
