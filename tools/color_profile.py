"""Where the first-step colouring of a scene spends its time (library built with -DRP_COLOR_PROFILE as rapier_amd/librapier_hip_colprof.so,
selected with RP_HIP_LIB): pass times, number of wavefront rounds, mean frontier size."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RP_HIP_LIB", os.path.join(ROOT, "rapier_amd", "librapier_hip_colprof.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from rapier_amd import PhysicsWorld, scenes as S, _ffi  # noqa: E402

for name, make in (("large_pyramid", S.large_pyramid), ("many_pyramids", S.many_pyramids)):
    w = PhysicsWorld.from_scene(make())
    w.step(1); w.sync()
    buf = np.zeros(64, np.int64)
    L = _ffi.lib()
    L.rp_debug_cycles.argtypes = [C.c_void_p, C.c_void_p]
    assert L.rp_debug_cycles(w._ptr, buf.ctypes.data) == 0
    t = buf[40:47]
    names = ["count", "reserve", "fill", "rank", "succ+frontier", "rounds"]
    print(name, "pairs", buf[50], "rounds", buf[48], "items/round %.1f" % (buf[49] / max(buf[48], 1)),
          " | ".join(f"{n} {(t[k + 1] - t[k]) / 100:.0f} us" for k, n in enumerate(names)), f"| total {(t[6] - t[0]) / 100:.0f} us")
