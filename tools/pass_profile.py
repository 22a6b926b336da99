"""Per-pass times of the fused rebuild kernels (k_layout_rebuild, k_bp_rebuild full pass) on a scene: builds
rapier_amd/librapier_hip_passprof.so with -DRP_PASS_PROFILE (on the machine that has hipcc; the .so travels to the GPU box) when run
with `build`, otherwise steps the scene and prints the averages.   python tools/pass_profile.py build | <scene> <steps>"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rapier_amd", "librapier_hip_passprof.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    src = os.path.join(ROOT, "rapier_amd", "csrc")
    subprocess.run(["make", "-s", "-C", src], check=True)
    flags = "-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DRP_PASS_PROFILE".split()
    objs = []
    for f in ("rp_api", "rp_broadphase", "rp_narrowphase", "rp_solver", "rp_islands", "rp_islands_lean", "rp_joints", "rp_sleep", "rp_flow", "rp_tiles"):
        if f in ("rp_islands", "rp_broadphase"):
            o = f"/tmp/{f}_pp.o"
            subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(src, f + ".hip"), "-o", o], check=True)
        else:
            o = os.path.join(src, f + ".o")
        objs.append(o)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    print("built", LIB)
    sys.exit(0)
os.environ.setdefault("RP_HIP_LIB", LIB)
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from rapier_amd import PhysicsWorld, scenes as S, _ffi  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "large_pyramid"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 160
scene = {"large_pyramid": S.large_pyramid, "many_pyramids": S.many_pyramids, "joint_grid": S.joint_grid}[name]()
w = PhysicsWorld.from_scene(scene)
w.step(steps); w.sync()
L = _ffi.lib()
L.rp_debug_read.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
buf = np.zeros(64, np.int64)
assert L.rp_debug_read(w._ptr, 200, 64, buf.ctypes.data) == 0
lay = ["init+clear", "bucket count", "union", "isl_count", "isl_number", "isl_fill", "owner prefix + stage order", "scatter", "rank overflow"]
bp = ["build (bucket slots)", "pairs", "finish + rest state"]
for title, base, names in (("k_layout_rebuild", 0, lay), ("k_bp_rebuild (full pass)", 20, bp)):
    n = max(int(buf[base + 15]), 1)
    tot = sum(int(buf[base + k]) for k in range(len(names)))
    print(f"{name} {title}: {n} dirty launches, {tot / n / 100:.1f} us each:", " | ".join(f"{nm} {int(buf[base + k]) / n / 100:.1f}" for k, nm in enumerate(names)))
print(w.counters())

why = np.zeros(9, np.int64)
assert L.rp_debug_read(w._ptr, 240, 9, why.ctypes.data) == 0
if why[0]:
    print(f"{name} broad-phase passes {why[0]}: incremental {why[1]}, not incremental because: grid not ok {why[2]}, too many changed {why[3]}, stale list full {why[4]}, tombstones {why[5]}; "
          f"changed colliders per pass {why[6] / why[0]:.0f}, stale colliders {why[7] / why[0]:.0f}; large list {why[8]} (colliders spanning > 3 cells + those that met a full bucket)")
