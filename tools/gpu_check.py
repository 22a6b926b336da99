"""Ad-hoc GPU-vs-oracle comparison used during bring-up (run through gpurun)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402
from oracle_ffi import OracleWorld  # noqa: E402


def compare(scene, checkpoints):
    g = PhysicsWorld.from_scene(scene)
    o = OracleWorld(scene)
    done = 0
    for cp in checkpoints:
        g.step(cp - done)
        o.step(cp - done)
        done = cp
        gp, gv = g.read_bodies()
        op, ov = o.read()
        dp = np.abs(gp[:, :3] - op[:, :3]).max()
        dq = np.minimum(np.abs(gp[:, 3:] - op[:, 3:]).max(axis=1), np.abs(gp[:, 3:] + op[:, 3:]).max(axis=1)).max()
        dv = np.abs(gv - ov).max()
        c = g.counters()
        print(f"[{scene.name}] step {cp}: dpos {dp:.3e} dquat {dq:.3e} dvel {dv:.3e} | gpu M={c['num_manifolds']} pairs={c['num_pairs']} "
              f"colors={c['num_colors']} par={c['num_parallel_stages']} rebuilds={c['bp_rebuilds']} ovf={c['overflow_flags']} q={c['quarantined']} "
              f"| oracle {o.stats()['num_active_manifolds']} {o.stats()['num_pairs']}", flush=True)
    return g, o


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "small"):
        compare(S.box_stack(3), [1, 2, 5, 20, 60])
        compare(S.pyramid10(), [1, 2, 5, 20, 100, 300])
    if which in ("all", "many"):
        g, o = compare(S.many_pyramids(), [1, 2, 10, 50])
        t = time.time(); g.step(200); g.sync(); dt = time.time() - t
        print(f"many_pyramids GPU: {200 / dt:.1f} steps/s ({dt / 200 * 1e3:.3f} ms/step)")
        g.enable_timers(True); g.step(50); g.sync(); print(g.counters()); g.enable_timers(False)
