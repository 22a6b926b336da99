"""How far the sharded run drifts from the whole world after a migrated cube lands (tests/test_gpu_migration.py), per look."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S, sharding

for every in (1, 4):
    sc = S.many_pyramids(1, 2)
    whole = PhysicsWorld.from_scene(sc)
    groups = sharding.proximity_groups_from_scene(sc)
    body_rank, _ = sharding.shards_from_groups(groups, 2)
    shards = sharding.ShardSet(sc, 2, lambda sub, r: PhysicsWorld.from_scene(sub), body_rank=body_rank, groups=groups, check_every=every)
    whole.step(4); shards.step(4)
    top = max((i for i in range(len(sc.bodies)) if body_rank[i] == 0), key=lambda i: float(sc.bodies[i]["translation"][1]))
    other_x = np.mean([float(sc.bodies[i]["translation"][0]) for i in range(len(sc.bodies)) if body_rank[i] == 1])
    toward = float(np.sign(other_x - float(sc.bodies[top]["translation"][0])))
    kick = np.array([[toward * 9.0, 6.0, 0.0, 0.0, 0.0, 0.0]], np.float32)
    whole.write_bodies([top], vel6=kick)
    shards.worlds[0].write_bodies([shards.handle[0][top]], vel6=kick)
    print(f"check_every={every} top={top}")
    for look in range(1, 36):
        whole.step(4); shards.step(4)
        gp, gv = shards.read_bodies(); wp, wv = whole.read_bodies()
        d = np.abs(gp[:, :3] - wp[:, :3]).max(1)
        print(f"  step {4 * look:3d} migrations {shards.migrations} refreshes {shards.guard_refreshes} owner {shards.owner[top]} max diff {d.max():.3e} at body {int(d.argmax())} cube at {gp[top, :3].round(2).tolist()} whole {wp[top, :3].round(2).tolist()}")
