"""Cost of live-world edits: a settled b3d_many_pyramids 6 x 6 world, then 200 ticks of
insert body + collider / remove the oldest inserted body / step — milliseconds per call, host wall clock."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S

w = PhysicsWorld.from_scene(S.many_pyramids(6, 6))
w.step(200); w.sync()
t0 = time.perf_counter(); w.step(200); w.sync(); base = (time.perf_counter() - t0) / 200
alive, t_ins, t_rem, t_step = [], 0.0, 0.0, 0.0
for tick in range(200):
    body = S.body_desc(translation=(float(-30 + 0.3 * tick), 14.0, float(-30 + (tick % 7))))
    col = S.collider_desc(half_extents=(0.5, 0.5, 0.5), density=100.0)
    t = time.perf_counter(); hb = w.insert_body(body); w.insert_collider(col, hb); t_ins += time.perf_counter() - t
    alive.append(hb)
    if len(alive) > 10:
        t = time.perf_counter(); w.remove_body([alive.pop(0)]); t_rem += time.perf_counter() - t
    t = time.perf_counter(); w.step(1); w.sync(); t_step += time.perf_counter() - t
c = w.counters()
print(f"settled step {base * 1e3:.3f} ms | per tick: insert body + collider {t_ins / 200 * 1e3:.3f} ms, remove body {t_rem / 190 * 1e3:.3f} ms, step after the edits {t_step / 200 * 1e3:.3f} ms | rows {w._lib.rp_num_bodies(w._ptr)} bp_rebuilds {c['bp_rebuilds']} overflow {c['overflow_flags']}")
