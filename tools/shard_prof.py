"""rocprofv3 target: one C4 shard (365 pyramids far from the origin: a third of its steps abort and are replayed), 60 + 400 steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapier_amd import PhysicsWorld, scenes as S, sharding
full = S.many_pyramids(54, 54)
wf = PhysicsWorld.from_scene(full, 0); wf.step(1); groups = wf.proximity_groups(); wf.close()
body_rank, n_groups = sharding.shards_from_groups(groups, 8)
scene, gids = sharding.partition_scene(full, body_rank, 0)
w = PhysicsWorld.from_scene(scene, 0)
w.step(60); w.sync()
w.step(int(sys.argv[1]) if len(sys.argv) > 1 else 400); w.sync()
c = w.counters()
print({k: c[k] for k in ("fast_steps", "full_steps", "replayed_steps", "fused_launches")})
