"""Where do a C4 shard's replays come from?  The same 365 pyramids (a) as a world of their own, (b) as the shard sub-world without the
guard, (c) with the guard armed — fast / full / replayed steps of 60 + 1000 steps each.   python tools/replay_diag.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rapier_amd import PhysicsWorld, scenes as S, sharding
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
def run(name, scene, guard=None):
    w = PhysicsWorld.from_scene(scene, 0)
    if guard is not None: w.set_shard_guard(*guard)
    w.step(60); w.sync()
    c0 = w.counters()
    t = time.perf_counter(); w.step(steps); w.sync(); dt = time.perf_counter() - t
    c = w.counters()
    print(f"{name}: {steps / dt:,.0f} steps/s; fast {c['fast_steps'] - c0['fast_steps']} full {c['full_steps'] - c0['full_steps']} replayed {c['replayed_steps'] - c0['replayed_steps']} fused launches {c['fused_launches'] - c0['fused_launches']} islands {c['num_islands']}", flush=True)
    w.close()
run("19 x 19 pyramids, a world of their own", S.many_pyramids(19, 19))
full = S.many_pyramids(54, 54)
wf = PhysicsWorld.from_scene(full, 0); wf.step(1); groups = wf.proximity_groups(); wf.close()
body_rank, n_groups = sharding.shards_from_groups(groups, 8)
scene, gids = sharding.partition_scene(full, body_rank, 0)
run("shard 0 of 8 (365 pyramids), no guard", scene)
run("shard 0 of 8, guard armed", scene, sharding.guard_boxes(full, groups, body_rank, 0))
