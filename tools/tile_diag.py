"""Counters of a scene step by step (which path the global solver took; tiles built?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S
sc = {"lp": lambda: S.large_pyramid(200), "jg": lambda: S.joint_grid(100)}.get(sys.argv[1] if len(sys.argv) > 1 else "", lambda: S.reference_pile(14, 5, 14, chain=False, sleep=False))()
w = PhysicsWorld.from_scene(sc)
done = 0
for cp in (1, 5, 20, 40, 60, 80, 120) + ((int(sys.argv[2]),) if len(sys.argv) > 2 else ()):
    w.step(cp - done); done = cp
    c = w.counters()
    print(cp, {k: c[k] for k in ("num_manifolds", "num_colors", "num_parallel_stages", "num_tiles", "tile_sweeps", "overflow_flags", "num_pairs", "full_updates", "lean_steps", "tile_step_steps")})
import ctypes as C, numpy as np
from rapier_amd import _ffi
L = _ffi.lib()
L.rp_debug_read.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
buf = np.zeros(16, np.int64)
print("rp_debug_read", L.rp_debug_read(w._ptr, 900, 16, buf.ctypes.data))
print("tilings", buf[0], "reason", buf[1], "NG", buf[2], "NT", buf[3], "T", buf[4], "max cone bodies", buf[5], "max cone cons", buf[6], "sum cone bodies", buf[7], "sum cone cons", buf[8])

prof = np.zeros(48, np.int64)
L.rp_debug_read(w._ptr, 920 + 24, 48, prof.ctypes.data)
for mode, name in ((0, "biased"), (1, "relaxed")):
    r = prof[24 * mode: 24 * mode + 24]
    if r[2]:
        n = float(r[2])
        print(f"tile 0 {name} sweep ({int(n)} launches): load {r[0] / n / 100:.2f} us, joint prepare {r[3] / n / 100:.2f} us, store {r[1] / n / 100:.2f} us, stages " + " ".join(f"{r[4 + k] / n / 100:.2f}" for k in range(17) if r[4 + k]))

# every tile's whole sweep (thread 0 of the tile, kernel entry to its last store)
per = np.zeros(520, np.int64)
L.rp_debug_read(w._ptr, 300, 520, per.ctypes.data)
nj = np.zeros(70, np.int64)
L.rp_debug_read(w._ptr, 830, 70, nj.ctypes.data)
for name, off, n in (("biased", 0, float(prof[2])), ("relaxed", 260, float(prof[26]))):
    v = per[off: off + 256].astype(float)
    v = v[v > 0]
    if n and len(v):
        v = v / n / 100
        print(f"{name}: {len(v)} tiles, whole sweep per tile: min {v.min():.2f} median {np.median(v):.2f} mean {v.mean():.2f} p90 {np.percentile(v, 90):.2f} max {v.max():.2f} us (argmax {int(v.argmax())})")
        if off == 0: print("  first tiles (us, joints):", " ".join(f"{v[k]:.1f}/{int(nj[k])}" for k in range(min(70, len(v)))))

# the joint-net launch (k_joint_net_step), workgroup 0: ticks per phase, summed over the substeps of a launch
jn = np.zeros(12, np.int64)
L.rp_debug_read(w._ptr, 260, 12, jn.ctypes.data)
if jn[11]:
    n = float(jn[11]) * 100
    names = ("prologue (lists, first entry)", "increment into LDS", "rows from poses", "biased stages", "integrate + publish", "wait for the neighbouring tiles 1", "halo reload", "relaxed stages", "publish", "wait for the neighbouring tiles 2", "impulses out")
    print(f"k_joint_net_step, workgroup 0, {int(jn[11])} launches, us per launch: " + "; ".join(f"{names[k]} {jn[k] / n:.2f}" for k in range(11)) + f"; total {jn[:11].sum() / n:.1f}")

# the one-launch TGS loop of a tiled contact world (k_tile_step), workgroup 100: ticks per phase, summed over the substeps of a launch
ts = np.zeros(13, np.int64)
L.rp_debug_read(w._ptr, 272, 13, ts.ctypes.data)
if ts[12]:
    n = float(ts[12]) * 100
    names = ("prologue (lists)", "A update + terms of the owned manifolds", "flag A", "B increment + warm start of the owned bodies", "flag B", "halo in (C)", "biased stages", "integrate + publish", "flag C",
             "halo in (D)", "relaxed stages", "publish")
    print(f"k_tile_step, workgroup 100, {int(ts[12])} launches, us per launch: " + "; ".join(f"{names[k]} {ts[k] / n:.2f}" for k in range(12)) + f"; total {ts[:12].sum() / n:.1f}")
