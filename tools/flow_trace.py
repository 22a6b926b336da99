"""Event timeline of ONE body under the dataflow solver (library built with -DRP_FLOW_TRACE, see tools/gpu_trace.sh): for every
ticket of body n_bodies/2 the wall-clock time (10 ns ticks) at which the event was published; prints the gaps between events."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapier_amd import PhysicsWorld, scenes as S, _ffi  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "large_pyramid"
scene = {"large_pyramid": S.large_pyramid, "joint_grid": S.joint_grid}[name]()
w = PhysicsWorld.from_scene(scene)
w.step(int(sys.argv[2]) if len(sys.argv) > 2 else 120); w.sync()
L = _ffi.lib()
L.rp_debug_read.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
buf = np.zeros(960, np.int64)
assert L.rp_debug_read(w._ptr, 64, 960, buf.ctypes.data) == 0
t = buf[buf > 0]
n = int((buf > 0).sum())
idx = np.flatnonzero(buf > 0)
print(f"{name}: traced body has {n} published events, tickets {idx.min()}..{idx.max()}, span {(t.max() - t.min()) / 100:.1f} us")
ts = buf[idx.min():idx.max() + 1]
gaps = np.diff(ts) / 100.0
print("gaps between consecutive tickets (us):")
for k in range(0, len(gaps), 10):
    print(f"  ticket {idx.min() + k + 1:4d}: " + " ".join(f"{g:6.2f}" for g in gaps[k:k + 10]))
