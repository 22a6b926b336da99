"""Step a scene N times on the GPU (profiling driver for rocprofv3)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "many_pyramids"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
scene = {"many_pyramids": S.many_pyramids, "large_pyramid": S.large_pyramid, "pyramid10": S.pyramid10}[name]()
w = PhysicsWorld.from_scene(scene)
w.step(60); w.sync()
t = time.time(); w.step(steps); w.sync(); dt = time.time() - t
print(f"{name}: {steps / dt:.1f} steps/s  {dt / steps * 1e3:.3f} ms/step", w.counters())
