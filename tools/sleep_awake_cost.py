"""Cost of a step of a sleep-enabled world while its bodies are still awake (b3d_many_pyramids with can_sleep(true): every
pyramid falls asleep after ~36 steps, so the steps 4..30 are timed, over several fresh worlds)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402

tot, n = 0.0, 0
for rep in range(8):
    w = PhysicsWorld.from_scene(S.many_pyramids().enable_sleep())
    w.step(4); w.sync()
    t = time.perf_counter(); w.step(26); w.sync(); tot += time.perf_counter() - t; n += 26
    assert w.sleeping().sum() == 0
print(f"many_pyramids, sleeping allowed, all awake: {tot / n * 1e6:.1f} us/step ({n / tot:.0f} steps/s)")
