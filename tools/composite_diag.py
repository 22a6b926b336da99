"""First differing step of a composite scene between device and oracle, with both sides' solver manifolds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RP_INDEX_ADDRESSING", "1")
from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld

def scene(n_compound=1, n_loose=0):
    s = S.Scene(name="compounds", gravity=(0.0, -9.81, 0.0))
    gb = s.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0)); s.add_collider(gb, half_extents=(8.0, 0.5, 8.0))
    ell = s.add_compound([S.collider_desc(half_extents=(0.5, 0.25, 0.25)), S.collider_desc(half_extents=(0.25, 0.5, 0.25), translation=(0.75, 0.25, 0.0))])
    for k in range(n_compound):
        b = s.add_body(translation=(-3.0 + 1.2 * k, 1.0 + 0.7 * k, 0.2 * (k % 3)), angvel=(0.4 * k, 0.2, -0.3), linvel=(0.3, 0.0, 0.1 * k))
        s.add_collider(b, shape=S.SHAPE_COMPOUND, half_extents=(ell, 0, 0), friction=0.4)
    return s

s = scene(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
g, o = PhysicsWorld.from_scene(s), OracleWorld(s)
print("mass props oracle", o.mass_props(1))
for step in range(1, 60):
    g.step(1); o.step(1)
    gp, gv = g.read_bodies(); op, ov = o.read()
    same = np.array_equal(gp, op) and np.array_equal(gv, ov)
    gm = g.contacts(); om = o.manifolds()
    print(step, "same" if same else "DIFF", "dev manifolds", len(gm[0]) if gm is not None else None, "oracle", len(om[0]), "pairs", g.counters()["num_pairs"], o.stats()["num_pairs"])
    import ctypes as C
    from oracle_ffi import lib as olib
    from rapier_amd import _ffi
    db, ob = np.zeros(256, np.float32), np.zeros(256, np.float32)
    L = _ffi.lib(); L.rp_debug_pair_points.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]; L.rp_debug_pair_points.restype = C.c_int32
    nd = L.rp_debug_pair_points(g._ptr, 0, 1, 256, db.ctypes.data)
    OL = olib(); OL.ro_debug_pair_points.restype = C.c_int32
    no = OL.ro_debug_pair_points(C.c_void_p(o._w), C.c_int32(0), C.c_int32(1), C.c_int32(256), C.c_void_p(ob.ctypes.data))
    if step >= 22:
        print("   dev pts", nd, np.round(db[:max(nd, 0)], 5).tolist()); print("   ora pts", no, np.round(ob[:max(no, 0)], 5).tolist())
    print("   oracle clusters", o.pair_clusters(0, 1), "dev imp", gm[2].tolist(), "ora imp", om[2].tolist())
    if not same or len(gm[0]) != len(om[0]):
        print(" dev meta", gm[0].tolist()); print(" dev nrm", gm[1].tolist()); print(" dev imp", gm[2].tolist())
        print(" ora meta", om[0].tolist()); print(" ora nrm", om[1].tolist()); print(" ora imp", om[2].tolist())
        print(" dpos", np.abs(gp - op).max(), "dvel", np.abs(gv - ov).max())
    if not same:
        break
