"""Cycle breakdown of k_island_solve (island 0); needs the library built with -DRP_ISL_PROFILE."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapier_amd import PhysicsWorld, scenes as S, _ffi  # noqa: E402

rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (14, 14)
base = int(sys.argv[3]) if len(sys.argv) > 3 else 10   # cubes in a pyramid's bottom row (10: b3d_many_pyramids; 9: 117 manifolds = 4 wavefronts of lanes)
w = PhysicsWorld.from_scene(S.many_pyramids(rows=rows, cols=cols, base_count=base))
print(f"{rows * cols} islands of base {base}, RP_ISL_DENSE={os.environ.get('RP_ISL_DENSE')}")
w.step(200); w.sync()
buf = np.zeros(64, np.int64)
L = _ffi.lib()
L.rp_debug_cycles.argtypes = [C.c_void_p, C.c_void_p]
L.rp_debug_cycles(w._ptr, buf.ctypes.data)
n = max(int(buf[63]), 1)
names = ["load", "generate+pose0", "ws terms + increment (substeps 1..)", "ws accumulate", "biased sweep", "integrate", "pose stage", "relax sweep", "writeback", "extra empty sweep", "fused: validate+arrive", "fused: wait for arrivals",
         "ws terms + increment (substep 0: incl. the wait for the validating wavefronts)"]
print(f"validation as its own wavefronts see it: first validating lane {buf[13] / n:.0f} cycles, last lane {buf[14] / n:.0f} cycles (both start at the barrier behind the body loads)")
tot = buf[:13].sum() / n
for k, nm in enumerate(names):
    print(f"{nm:38s} {buf[k] / n:10.0f} cycles/step  {100.0 * buf[k] / n / tot:5.1f}%")
print(f"total {tot:.0f} cycles/step over {n} steps")
