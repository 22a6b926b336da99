"""Step a scene N times on the GPU (profiling driver for rocprofv3); prints steps/s and per-stage ms."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "many_pyramids"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
variants = {
    "many_pyramids": S.many_pyramids, "large_pyramid": S.large_pyramid, "pyramid10": S.pyramid10, "joint_grid": S.joint_grid,
    # feature variants of the headline scene (full step path): sleeping allowed, Coulomb friction, all events on
    "many_pyramids_sleep": lambda: S.many_pyramids().enable_sleep(),
    "many_pyramids_coulomb": lambda: _with(S.many_pyramids(), "friction_model", S.FRICTION_COULOMB),
    "many_pyramids_events": lambda: S.many_pyramids().enable_events(3, 100.0),
    "washer": S.washer, "junkyard": S.junkyard,   # the reference's box3d ports (tools/b3d_ports.py; the junkyard's pusher stands still here)
    "batch_capsules": lambda: S.batch([S.capsules(6) for _ in range(int(os.environ.get("RP_BATCH_N", "256")))]),   # 256 small worlds as sub-worlds of one (tools/subworld_rate.py)
    "shapes_rain": lambda: S.shapes_rain(8000),   # all ten shape kinds landing on a slab (tools/shapes_bench.py)
}


def _with(scene, key, value):
    scene.params[key] = value
    return scene


scene = variants[name]()
w = PhysicsWorld.from_scene(scene)
w.step(300 if name == "batch_capsules" else 60); w.sync()
t = time.time(); w.step(steps); w.sync(); dt = time.time() - t
print(f"{name}: {steps / dt:.1f} steps/s  {dt / steps * 1e3:.3f} ms/step", w.counters())
if os.environ.get("RP_PROF_COUNTERS"):  # (tools/gpu_profile.sh: pmc_summary.py turns bytes per LAUNCH into bytes per STEP with the fused steps / launches of this run)
    import json
    with open(os.environ["RP_PROF_COUNTERS"], "w") as f:
        json.dump({k: int(v) if isinstance(v, (int,)) else float(v) for k, v in w.counters().items()}, f)
try:  # hand-off statistics of the dataflow launch (rp_flow.hip), accumulated since world creation
    import ctypes as C
    import numpy as np
    from rapier_amd import _ffi
    buf = np.zeros(64, np.int64)
    L = _ffi.lib()
    L.rp_debug_cycles.argtypes = [C.c_void_p, C.c_void_p]
    if L.rp_debug_cycles(w._ptr, buf.ctypes.data) == 0 and buf[20]:
        print(f"{name} flow: wave-items {buf[20]}  polls/item {buf[21] / buf[20]:.2f}  applies/item {buf[22] / buf[20]:.2f}  waves*launches {buf[23]}"
              f"  per wave-launch: kernel {buf[26] / buf[23] / 100:.1f} us, in wait loops {buf[25] / buf[23] / 100:.1f} us"
              )
except Exception as e:  # noqa: BLE001
    print("no flow statistics:", e)
if os.environ.get("RP_PROF_TIMERS", "1") == "1":
    w.enable_timers(True); w.step(50); w.sync()
    c = w.counters()
    print(f"{name} stage ms: collision {c['collision_detection_ms']:.4f} solver-loop {c['velocity_resolution_ms']:.4f} finish {c['velocity_update_ms']:.4f}")
