"""Is the tile sweep of one giant island bound by HBM or by the Infinity Cache?  (first question.)  b3d_large_pyramid
at base 200 / 300 / 400: the constraint rows a step streams (816 B x manifolds + the six shadow planes) grow from ~55 MB — inside the
256 MB Infinity Cache — to ~220 MB, past what it keeps between sweeps.  If the solver loop's time PER MANIFOLD stays put, the sweeps do
not live off the cache.  Prints steps/s, solver-loop ms (hipEvent timers) and ns per manifold per step after a settling run."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402

print("base  cuboids  manifolds  tiles  steps/s  solver-loop ms  ns per manifold-step  rows MB")
for base in [int(a) for a in sys.argv[1:]] or (200, 300, 400):
    w = PhysicsWorld.from_scene(S.large_pyramid(base))
    w.step(400); w.sync()
    t = time.perf_counter(); w.step(200); w.sync(); dt = (time.perf_counter() - t) / 200
    w.enable_timers(True); w.step(60); w.sync()
    c = w.counters()
    w.enable_timers(False)
    M = c["num_manifolds"]
    print(f"{base:4d} {c['num_dynamic_bodies']:8d} {M:10d} {c['num_tiles']:6d} {1 / dt:8.1f} {c['velocity_update_ms']:12.4f} {c['velocity_update_ms'] * 1e6 / max(M, 1):14.2f} {M * (816 + 96) / 1e6:10.1f}", flush=True)
    del w
