"""Batched small worlds (rp_world_begin_subworld): n copies of capsules(6) — 19 bodies each — as the sub-worlds of ONE device world,
against one such world stepped alone (a small world alone is latency-bound, ~60-110 us per step whatever its size).
usage: subworld_rate.py [n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402


def rate(w, steps):
    w.step(300); w.sync()                     # the capsules have landed
    t = time.perf_counter(); w.step(steps); w.sync(); dt = time.perf_counter() - t
    return steps / dt, w.counters()


ns = [int(a) for a in sys.argv[1:]] or [1, 16, 64, 256, 1024]
print(f"{'worlds':>7} {'bodies':>8} {'batch steps/s':>14} {'world-steps/s':>14} {'us/batch step':>14}  paths (fused / fast / full / replayed)")
for n in ns:
    sc = S.capsules(6) if n == 1 else S.batch([S.capsules(6) for _ in range(n)])
    w = PhysicsWorld.from_scene(sc)
    r, c = rate(w, 600)
    print(f"{n:7d} {len(sc.bodies):8d} {r:14.0f} {r * n:14.0f} {1e6 / r:14.1f}  {c['fused_steps']} / {c['fast_steps']} / {c['full_steps']} / {c['replayed_steps']}  islands {c['num_islands']} global bodies {c['num_global_bodies']} manifolds {c['num_manifolds']}")
    w.close() if hasattr(w, "close") else None
