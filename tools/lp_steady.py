import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S
w = PhysicsWorld.from_scene(S.large_pyramid())
for warm in (60, 500, 1500):
    w.step(warm); w.sync()
    t = time.time(); w.step(200); w.sync(); dt = time.time() - t
    w.enable_timers(True); w.step(50); w.sync(); c = w.counters(); w.enable_timers(False)
    print(f"after +{warm}: {200/dt:.0f} steps/s {dt/200*1e3:.3f} ms/step | bp {c['broad_phase_ms']:.3f} np {c['narrow_phase_ms']:.3f} isl {c['island_construction_ms']:.3f} coll {c['collision_detection_ms']:.3f} loop {c['velocity_resolution_ms']:.3f} glob {c['velocity_update_ms']:.3f} full_updates {c['full_updates']} bp_rebuilds {c['bp_rebuilds']} colors {c['num_colors']} stages {c['num_parallel_stages']}")
