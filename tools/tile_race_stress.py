"""Replay of the round-3 intermittent tile-path mismatch and the check that it is gone.

Root cause: finalize() uploaded DevWorld::b_order (iota) and tl_bbox (rest state) with a synchronous hipMemcpy — legacy stream — right
behind the hipMemsetAsync zero fill of the same arrays on the world's NON-BLOCKING stream.  Nothing orders the two: when the fill ran
late it wiped the upload.  b_order == 0 everywhere ranks every manifold of a colour to the same constraint position (k_layout_rebuild)
=> a gross mismatch from the first step on.  The fill runs late when the device is busy, so this tool keeps it busy: a second world
(launched eagerly, RP_NO_GRAPH=1: no host throttle) has tens of milliseconds of kernels queued while the world under test is built.

  python tools/tile_race_stress.py [iterations] [target]          (RP_HIP_LIB=<old build> to replay the failure)

Prints one line per failing iteration and a JSON summary; exit code 1 when any iteration differs from the oracle."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
# the --replay hook (RP_TEST_LATE_FILL) only exists in the testing build of the library (make -C rapier_amd/csrc testing)
os.environ.setdefault("RP_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rapier_amd", "librapier_hip_testing.so"))

from rapier_amd import PhysicsWorld, scenes as S   # noqa: E402
from oracle_ffi import OracleWorld                 # noqa: E402
import oracle_ffi                                  # noqa: E402


def replay():
    """the fingerprint of the failure: the late fill replayed on purpose (RP_TEST_LATE_FILL, a hook in finalize()) must give the very
    numbers of the round-3 log (gpurun_out/r03z1_pytest_1.log: 9478 of 12817 pose elements, max abs 0.44956553 after rp_step(2))"""
    oracle_ffi.set_threads(max(1, min(os.cpu_count() or 1, 16)))
    scene = S.large_pyramid(60)
    os.environ["RP_TILE_TARGET"] = "3"
    out = []
    for mode in (1, 2, 3, 0):
        os.environ["RP_TEST_LATE_FILL"] = str(mode)
        for first in (2, 1):
            g = PhysicsWorld.from_scene(scene); o = OracleWorld(scene)
            g.step(0)
            g.step(first); o.step(first)
            gp, _ = g.read_bodies(); op, _ = o.read()
            rec = {"late_fill": mode, "steps": first, "mismatched": int((gp != op).sum()), "of": int(gp.size),
                   "max_abs": float(np.nanmax(np.abs(gp - op)))}
            print(json.dumps(rec), flush=True)
            out.append(rec)
            g.close()
    return 0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--replay":
        return replay()
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    target = sys.argv[2] if len(sys.argv) > 2 else "3"
    oracle_ffi.set_threads(max(1, min(os.cpu_count() or 1, 16)))
    scene = S.large_pyramid(60)
    o = OracleWorld(scene)
    want = {}
    done = 0
    for cp in (1, 2, 6):
        o.step(cp - done); done = cp
        want[cp] = o.read()
    os.environ["RP_NO_GRAPH"] = "1"
    busy = PhysicsWorld.from_scene(S.large_pyramid(60))
    busy.step(1); busy.sync()
    del os.environ["RP_NO_GRAPH"]
    os.environ["RP_TILE_TARGET"] = target
    bad = []
    for it in range(iters):
        g = PhysicsWorld.from_scene(scene)           # host mirrors only: the device world is built by the first rp_step
        busy.step(40 + (it % 5) * 10)                # ~20-40 ms of kernels queued on another stream
        g.step(0)                                    # finalize() under a busy device
        form = it % 3                                # [2, 6] (two steps back to back), [1, 2, 6], [1, 6]
        cps = ((2, 6), (1, 2, 6), (1, 6))[form]
        done = 0
        for cp in cps:
            g.step(cp - done); done = cp
            gp, gv = g.read_bodies()
            op, ov = want[cp]
            if not (np.array_equal(gp, op) and np.array_equal(gv, ov)):
                nbad = int((gp != op).sum())
                err = float(np.nanmax(np.abs(gp - op))) if np.isfinite(gp).all() else float("nan")
                print(f"iteration {it} form {cps}: step {cp} differs: {nbad} of {gp.size} pose elements, max abs error {err:.4g}", flush=True)
                bad.append(it)
                break
        g.close()
    busy.sync()
    print(json.dumps({"lib": os.environ.get("RP_HIP_LIB", "rapier_amd/librapier_hip.so"), "iterations": iters, "tile_target": target,
                      "failing_iterations": bad}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
