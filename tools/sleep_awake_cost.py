"""Cost of a step of a sleep-enabled world while its bodies are still awake (b3d_many_pyramids with can_sleep(true): every
pyramid falls asleep after ~36 steps, so 20 steps between step 10 and step 30 are timed, over several fresh worlds; the counters
say which way those steps went — fused single-kernel steps since round 5)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402

tot, n, paths = 0.0, 0, {}
for rep in range(8):
    w = PhysicsWorld.from_scene(S.many_pyramids().enable_sleep())
    w.step(6); w.sync(); w.step(3); w.sync(); w.step(1); w.sync()
    c0 = w.counters()
    t = time.perf_counter(); w.step(20); w.sync(); tot += time.perf_counter() - t; n += 20
    assert w.sleeping().sum() == 0
    c1 = w.counters()
    for k in ("fused_steps", "fast_steps", "full_steps", "replayed_steps"):
        paths[k] = paths.get(k, 0) + c1[k] - c0[k]
print(f"many_pyramids, sleeping allowed, all awake: {tot / n * 1e6:.1f} us/step ({n / tot:.0f} steps/s); the timed steps: {paths}")
