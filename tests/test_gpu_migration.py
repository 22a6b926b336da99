"""Island migration between shards ("when an AddPair links bodies on different shards, migrate the smaller island"):
two shards of one scene as two device worlds; a cube thrown from a pyramid of shard 0 towards a pyramid of shard 1 trips shard 0's
guard, is taken (rp_world_shard_guard_take_hits), removed from shard 0 and inserted into shard 1 with its pose and velocities, both
guards are refreshed and the run goes on — where round 3 ended it with RP_ERR_INVALID."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S, sharding

pytestmark = pytest.mark.gpu


def test_a_thrown_cube_migrates_to_the_shard_it_reaches():
    sc = S.many_pyramids(1, 2)
    whole = PhysicsWorld.from_scene(sc)
    groups = sharding.proximity_groups_from_scene(sc)
    body_rank, ng = sharding.shards_from_groups(groups, 2)
    assert ng == 2
    shards = sharding.ShardSet(sc, 2, lambda sub, r: PhysicsWorld.from_scene(sub, index_addressing=False), body_rank=body_rank, groups=groups)  # strict handles: the product default
    whole.step(5); shards.step(5)
    # the top cube of rank 0's pyramid, thrown towards the other pyramid (11 m away along x)
    top = max((i for i in range(len(sc.bodies)) if body_rank[i] == 0), key=lambda i: float(sc.bodies[i]["translation"][1]))
    other_x = np.mean([float(sc.bodies[i]["translation"][0]) for i in range(len(sc.bodies)) if body_rank[i] == 1])
    toward = float(np.sign(other_x - float(sc.bodies[top]["translation"][0])))
    kick = np.array([[toward * 9.0, 6.0, 0.0, 0.0, 0.0, 0.0]], np.float32)
    whole.write_bodies([top], vel6=kick)
    shards.worlds[0].write_bodies([shards.handle[0][top]], vel6=kick)
    moved_at = None
    for step in range(1, 140):
        whole.step(1); shards.step(1)
        if moved_at is None and shards.migrations > 0:
            moved_at = step
            assert shards.owner[top] == 1 and top in shards.handle[1] and top not in shards.handle[0]
            # in flight nothing touches the cube and the pyramids do not feel each other: the assembled shards ARE the whole world
            gp, gv = shards.read_bodies(); wp, wv = whole.read_bodies()
            np.testing.assert_array_equal(gp, wp); np.testing.assert_array_equal(gv, wv)
    assert moved_at is not None and shards.migrations == 1, "the guard never fired: the cube did not reach the other shard"
    gp, gv = shards.read_bodies(); wp, wv = whole.read_bodies()
    assert np.isfinite(gp).all() and np.isfinite(gv).all()
    # after the landing the cube sits in another row of another world (its index — hence the colour order of its contacts — differs from the
    # whole world's): same physics, not the same bits
    assert np.abs(gp[:, :3] - wp[:, :3]).max() < 0.05, float(np.abs(gp[:, :3] - wp[:, :3]).max())
    assert abs(float(gp[top, 0]) - other_x) < 6.0                     # it is over there
    for r in (0, 1):
        assert shards.worlds[r].counters()["overflow_flags"] == 0


def test_hits_looked_at_every_fourth_step_still_catch_the_cube_in_time():
    """check_every = 4: the device tests every body box inflated by |linvel| x 4 dt (rp_world_set_shard_guard_horizon), so the cube
    is handed over while it is still in flight; the job-wide top speed (rp_world_max_linear_speed) drives the refresh of the boxes"""
    sc = S.many_pyramids(1, 2)
    whole = PhysicsWorld.from_scene(sc)
    groups = sharding.proximity_groups_from_scene(sc)
    body_rank, _ = sharding.shards_from_groups(groups, 2)
    shards = sharding.ShardSet(sc, 2, lambda sub, r: PhysicsWorld.from_scene(sub, index_addressing=False), body_rank=body_rank, groups=groups, check_every=4)
    whole.step(4); shards.step(4)
    assert shards.guard_refreshes == 0                                   # a pyramid that settles does not move its box
    top = max((i for i in range(len(sc.bodies)) if body_rank[i] == 0), key=lambda i: float(sc.bodies[i]["translation"][1]))
    other_x = np.mean([float(sc.bodies[i]["translation"][0]) for i in range(len(sc.bodies)) if body_rank[i] == 1])
    toward = float(np.sign(other_x - float(sc.bodies[top]["translation"][0])))
    kick = np.array([[toward * 9.0, 6.0, 0.0, 0.0, 0.0, 0.0]], np.float32)
    whole.write_bodies([top], vel6=kick)
    shards.worlds[0].write_bodies([shards.handle[0][top]], vel6=kick)
    assert abs(shards.worlds[0].max_linear_speed() - float(np.linalg.norm(kick[0, :3]))) < 1e-5
    moved_at = None
    for look in range(1, 17):                                            # 64 steps: the cube is in the air all the time
        whole.step(4); shards.step(4)
        if moved_at is None and shards.migrations > 0:
            moved_at = look
        # before AND after the hand-over nothing was missed: the assembled shards are the whole world, bit for bit
        gp, gv = shards.read_bodies(); wp, wv = whole.read_bodies()
        np.testing.assert_array_equal(gp, wp); np.testing.assert_array_equal(gv, wv)
    assert moved_at is not None and shards.migrations == 1 and shards.owner[top] == 1
    assert shards.guard_refreshes > 0                                    # the flying cube dragged its box along
    whole.step(12); shards.step(12)                                      # the landing: in shard 1 the cube sits in another row (colour order differs)
    gp, _ = shards.read_bodies(); wp, _ = whole.read_bodies()
    assert np.isfinite(gp).all() and np.abs(gp[:, :3] - wp[:, :3]).max() < 0.1
    assert abs(float(gp[top, 0]) - other_x) < 6.0                        # it is over there
    for r in (0, 1):
        assert shards.worlds[r].counters()["overflow_flags"] == 0


def _pile_and_jointed_pair():
    """a slab, a pile of five cubes and — 8 m away — two cubes linked by a revolute joint (free axis X, no motor in the scene)"""
    s = S.Scene(name="pile_and_pair", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, shape=S.SHAPE_CUBOID, half_extents=(40.0, 0.5, 4.0))
    for k in range(5):
        b = s.add_body(translation=(-4.0 + 0.05 * (k % 2), 0.3 + 0.6 * k, 0.0))
        s.add_collider(b, shape=S.SHAPE_CUBOID, half_extents=(0.3, 0.3, 0.3))
    p = s.add_body(translation=(4.0, 0.3, 0.0)); s.add_collider(p, shape=S.SHAPE_CUBOID, half_extents=(0.3, 0.3, 0.3))
    q = s.add_body(translation=(4.0, 1.2, 0.0)); s.add_collider(q, shape=S.SHAPE_CUBOID, half_extents=(0.3, 0.3, 0.3))
    s.add_joint(p, q, anchor1=(0.0, 0.45, 0.0), anchor2=(0.0, -0.45, 0.0), locked_axes=S.LOCK_REVOLUTE)
    return s, p, q


def test_a_migrating_joint_carries_its_live_descriptor():
    """a joint of a migrating group used to be re-created from the scene's descriptor — a motor set at run time
    (rp_impulse_joints_set_motor) was lost.  The source world hands out the live descriptor (rp_impulse_joints_get =
    ImpulseJointSet::get) and the destination gets that one; a joint removed at run time stays removed."""
    sc, p, q = _pile_and_jointed_pair()
    whole = PhysicsWorld.from_scene(sc)
    groups = sharding.proximity_groups_from_scene(sc)
    body_rank, ng = sharding.shards_from_groups(groups, 2)
    assert ng == 2 and body_rank[p] == body_rank[q] != body_rank[1]
    src, dst = int(body_rank[p]), int(body_rank[1])
    shards = sharding.ShardSet(sc, 2, lambda sub, r: PhysicsWorld.from_scene(sub, index_addressing=False), body_rank=body_rank, groups=groups)
    whole.step(4); shards.step(4)
    # the run-time edit: a velocity motor on the free axis (GenericJoint::set_motor_velocity), in the whole world and in the shard
    ws = shards.worlds[src]
    hj = ws.joint_handles(); assert len(hj) == 1
    d0 = ws.impulse_joint_descs(hj)[0]
    assert int(d0["motor_axes"]) == 0 and int(d0["locked_axes"]) == S.LOCK_REVOLUTE
    assert {int(d0["body1"]), int(d0["body2"])} == {shards.handle[src][p], shards.handle[src][q]}   # RigidBodyHandles come back
    whole.set_joint_motor(whole.joint_handles(), [3], target_vel=2.0, damping=0.5)
    ws.set_joint_motor(hj, [3], target_vel=2.0, damping=0.5)
    d1 = ws.impulse_joint_descs(hj)[0]
    assert int(d1["motor_axes"]) == 1 << 3 and float(d1["motors"][3]["target_vel"]) == 2.0 and float(d1["motors"][3]["damping"]) == 0.5
    kick = np.array([[-9.0, 5.0, 0.0, 0.0, 0.0, 0.0]] * 2, np.float32)
    whole.write_bodies([p, q], vel6=kick)
    ws.write_bodies([shards.handle[src][p], shards.handle[src][q]], vel6=kick)
    moved_at = None
    for step in range(1, 120):
        whole.step(1); shards.step(1)
        gp, gv = shards.read_bodies(); wp, wv = whole.read_bodies()
        if shards.migrations == 0:
            np.testing.assert_array_equal(gp, wp); np.testing.assert_array_equal(gv, wv)
        elif moved_at is None:
            moved_at = step
            assert shards.owner[p] == dst and shards.owner[q] == dst
            wd = shards.worlds[dst]
            hd = wd.joint_handles(); hd = hd[hd != np.uint64(0xFFFFFFFFFFFFFFFF)]
            assert len(hd) == 1
            d2 = wd.impulse_joint_descs(hd)[0]
            assert int(d2["motor_axes"]) == 1 << 3 and float(d2["motors"][3]["target_vel"]) == 2.0 and float(d2["motors"][3]["damping"]) == 0.5
            for f in ("local_anchor1", "local_anchor2", "local_basis1", "local_basis2", "locked_axes", "limit_axes", "limits", "contacts_enabled", "coupled_axes"):
                np.testing.assert_array_equal(d2[f], d1[f])
            assert {int(d2["body1"]), int(d2["body2"])} == {shards.handle[dst][p], shards.handle[dst][q]}
            assert len(shards.worlds[src].joint_handles()[shards.worlds[src].joint_handles() != np.uint64(0xFFFFFFFFFFFFFFFF)]) == 0
            assert int(shards.joints[0]["motor_axes"]) == 1 << 3                 # the set's own table follows
        elif step == moved_at + 3:
            # in flight the motor keeps driving the pair at the destination as in the whole world (cold joint impulses for one step: close)
            rel_g, rel_w = gv[q, 3:] - gv[p, 3:], wv[q, 3:] - wv[p, 3:]
            assert np.abs(rel_g - rel_w).max() < 0.05, (rel_g, rel_w)
            assert np.abs(shards.worlds[dst].joint_motor_impulses()).max() > 0.0
    assert moved_at is not None and shards.migrations == 1, "the guard never fired"
    gp, gv = shards.read_bodies(); wp, wv = whole.read_bodies()
    assert np.isfinite(gp).all() and np.abs(gp[:, :3] - wp[:, :3]).max() < 0.1, float(np.abs(gp[:, :3] - wp[:, :3]).max())
    for r in (0, 1):
        assert shards.worlds[r].counters()["overflow_flags"] == 0
