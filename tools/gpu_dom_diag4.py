import sys, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from rapier_amd import PhysicsWorld, scenes as S
sc = S.box_stack(3); sc.bodies[2]["dominance"] = 1
g = PhysicsWorld.from_scene(sc)
g.step(7); g.sync()
out = np.zeros(64, np.int64)
g._lib.rp_debug_cycles.argtypes = [C.c_void_p, C.c_void_p]
g._lib.rp_debug_cycles(g._ptr, out.ctypes.data)
for i in range(20, 28):
    v = int(out[i])
    print(i - 20, "static", v & 0xff, "lid", ((v >> 8) & 0xff) - 1, "gid", ((v >> 16) & 0xff) - 1, "dppODD", ((v >> 24) & 0xff) - 1, "dppEVEN", ((v >> 32) & 0xff) - 1, "slot", v >> 40)
