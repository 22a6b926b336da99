"""Island sharding over GPUs from the device's own data: proximity groups through rp_bodies_proximity_group, and the
shard guard (rp_world_set_shard_guard) that turns a body wandering towards another shard's bodies into an error instead of a
silently wrong answer.  SURVEY section 8e; the reference's islands: island_manager/persistent.rs."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S, sharding

pytestmark = pytest.mark.gpu


def test_device_proximity_groups_are_the_pyramids():
    sc = S.many_pyramids(3, 4)
    w = PhysicsWorld.from_scene(sc)
    w.step(1)
    g = w.proximity_groups()
    assert g[0] == -1 and len(np.unique(g[g >= 0])) == 12
    assert (np.bincount(g[g >= 0])[np.unique(g[g >= 0])] == 55).all()
    ref = sharding.proximity_groups_from_scene(sc)            # same partition as the CPU stand-in (ids may differ)
    for a in np.unique(g[g >= 0]):
        assert len(set(ref[g == a].tolist())) == 1
    # joints link groups too
    jc = S.joint_chain(8)
    wj = PhysicsWorld.from_scene(jc)
    wj.step(1)
    gj = wj.proximity_groups()
    assert len(np.unique(gj[gj >= 0])) == 1


def test_sharded_worlds_equal_the_whole_world_and_the_guard_stays_quiet():
    sc = S.many_pyramids(2, 3)
    whole = PhysicsWorld.from_scene(sc)
    whole.step(1)
    groups = whole.proximity_groups()
    body_rank, ng = sharding.shards_from_groups(groups, 2)
    assert ng == 6
    whole.step(29)
    wp, wv = whole.read_bodies()
    for r in range(2):
        sub, gids = sharding.partition_scene(sc, body_rank, r)
        w = PhysicsWorld.from_scene(sub)
        w.set_shard_guard(*sharding.guard_boxes(sc, groups, body_rank, r))
        w.step(30)
        p, v = w.read_bodies()                                  # (raises if the guard fired)
        np.testing.assert_array_equal(p, wp[gids]); np.testing.assert_array_equal(v, wv[gids])


def test_guard_fires_when_a_body_reaches_another_shards_box():
    sc = S.many_pyramids(1, 2)
    full = PhysicsWorld.from_scene(sc)
    full.step(1)
    groups = full.proximity_groups()
    body_rank, _ = sharding.shards_from_groups(groups, 2)
    sub, gids = sharding.partition_scene(sc, body_rank, 0)
    w = PhysicsWorld.from_scene(sub)
    bmin, bmax = sharding.guard_boxes(sc, groups, body_rank, 0)
    w.set_shard_guard(bmin, bmax)
    w.step(5)
    w.read_bodies()
    # throw the top cube of this shard's pyramid towards the other pyramid
    local_top = len(sub.bodies) - 1
    toward = np.sign((bmin[0, 0] + bmax[0, 0]) / 2 - float(sub.bodies[local_top]["translation"][0]))
    w.write_bodies([local_top], vel6=[[toward * 40.0, 5.0, 0.0, 0.0, 0.0, 0.0]])
    from rapier_amd.world import RapierHipError
    with pytest.raises(RapierHipError, match="shard guard"):
        w.step(40)
        w.read_bodies()
    # without the guard the same motion is legal
    w2 = PhysicsWorld.from_scene(sub)
    w2.step(5)
    w2.write_bodies([local_top], vel6=[[toward * 40.0, 5.0, 0.0, 0.0, 0.0, 0.0]])
    w2.step(40)
    w2.read_bodies()


def test_native_collective_packs_on_the_device_and_gathers_through_rccl():
    """SURVEY 8(e)'s one collective through the library itself (rp_world_pack_bodies / rp_shard_all_gather): a one-rank RCCL communicator
    (ncclCommInitRank bound at run time from librccl.so.1), the shard's bodies packed by a kernel, ncclAllGather on the world's stream,
    scatter by global id.  The gathered rows equal rp_bodies_read of the shard, the rows of the other shard stay untouched."""
    from rapier_amd import ShardComm
    full = S.many_pyramids(rows=2, cols=4)
    body_rank = sharding.many_pyramids_body_ranks(2, 4, 10, 2)
    sub, gids = sharding.partition_scene(full, body_rank, 0)
    g = PhysicsWorld.from_scene(sub)
    g.step(7)
    pos, vel = g.read_bodies()
    dyn = np.array([int(b["body_type"]) == S.BODY_DYNAMIC for b in sub.bodies])
    ptr, n = g.pack_bodies()
    assert ptr and n == int(dyn.sum()) == 4 * 55
    comm = ShardComm(ShardComm.unique_id(), 1, 0, 0)
    gp, gv, per = sharding.all_gather_bodies_native(g, comm, pos, vel, gids, len(full.bodies), dyn, rows_per_rank=n + 13)
    assert per.tolist() == [n]
    np.testing.assert_array_equal(gp[gids], pos); np.testing.assert_array_equal(gv[gids], vel)
    others = np.setdiff1d(np.arange(len(full.bodies)), gids)
    assert len(others) == 4 * 55 and not gp[others].any() and not gv[others].any()
    with pytest.raises(Exception):   # a row budget below the shard's size is refused, not truncated
        sharding.all_gather_bodies_native(g, comm, pos, vel, gids, len(full.bodies), dyn, rows_per_rank=n - 1)
    comm.close()
