"""Steps/s of every BASELINE.json config that fits one GPU (and of the feature variants of the headline scene), each next to the
C oracle timed on the host cores over a bounded sample.  Writes one JSON document (profiles/r01_configs_1gpu.json when run by
tools/gpu_round.sh).  Debug / evidence aid — bench.py stays the contract for the headline metric.

    python tools/bench_configs.py [out.json] [--quick]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402
import oracle_ffi  # noqa: E402


def _with(scene, key, value):
    scene.params[key] = value
    return scene


CONFIGS = [
    # name, scene factory, warm-up steps, timed GPU steps, oracle sample steps
    ("C1 pyramid10 (55 cuboids; BASELINE 'pyramid3' plumbing case)", S.pyramid10, 60, 2000, 2000),
    ("C2 b3d_large_pyramid (20,100 cuboids, one island)", S.large_pyramid, 60, 300, 30),
    ("C3 b3d_many_pyramids (10,780 cuboids, 196 islands) — the bench.py workload", S.many_pyramids, 60, 2000, 300),
    ("C5 b3d_joint_grid (9,900 balls, 19,800 spherical joints)", S.joint_grid, 60, 500, 100),
    ("C3 + sleeping allowed (awake while settling, then idle steps)", lambda: S.many_pyramids().enable_sleep(), 60, 2000, 300),
    ("C3 + FrictionModel::Coulomb (islands on k_island_generic)", lambda: _with(S.many_pyramids(), "friction_model", S.FRICTION_COULOMB), 60, 300, 100),
    ("C3 + collision and contact-force events on every collider", lambda: S.many_pyramids().enable_events(3, 100.0), 60, 1000, 100),
    ("capsules(6) feature scene (full updates every step)", lambda: S.capsules(6), 30, 1000, 1000),
]


def main():
    out_path = next((a for a in sys.argv[1:] if not a.startswith("--")), None)
    quick = "--quick" in sys.argv
    cores = max(1, min(os.cpu_count() or 1, 32))
    rows = []
    for name, make, warm, steps, osteps in CONFIGS:
        if quick:
            steps, osteps = max(50, steps // 10), max(10, osteps // 10)
        w = PhysicsWorld.from_scene(make())
        w.step(warm); w.sync()
        t = time.perf_counter(); w.step(steps); w.sync(); dt = time.perf_counter() - t
        c = w.counters()
        row = {"config": name, "gpu_steps_per_s": steps / dt, "gpu_ms_per_step": dt / steps * 1e3, "gpu_steps": steps, "warmup": warm,
               "bodies": c["num_dynamic_bodies"], "manifolds": c["num_manifolds"], "colors": c["num_colors"],
               "fast_steps": c["fast_steps"], "full_steps": c["full_steps"], "replayed_steps": c["replayed_steps"], "sleeping_bodies": c["num_sleeping_bodies"]}
        del w
        best = None
        for threads in sorted({1, cores}):      # small scenes run faster on one thread than across an OpenMP team
            oracle_ffi.set_threads(threads)
            o = oracle_ffi.OracleWorld(make())
            o.step(warm)
            n = osteps if threads > 1 else max(10, osteps // 4)
            t = time.perf_counter(); o.step(n); odt = time.perf_counter() - t
            rate = n / odt
            row[f"oracle_steps_per_s_{threads}t"] = rate
            if best is None or rate > best[0]:
                best = (rate, threads, n)
            del o
        oracle_ffi.set_threads(1)
        row.update({"oracle_steps_per_s": best[0], "oracle_threads": best[1], "oracle_steps": best[2], "ratio": (steps / dt) / best[0]})
        rows.append(row)
        print(f"{name}: GPU {row['gpu_steps_per_s']:.0f} steps/s ({row['gpu_ms_per_step']:.3f} ms)  oracle[{row['oracle_threads']}t] {row['oracle_steps_per_s']:.1f} steps/s", flush=True)
    doc = {"what": "steps/s per config on one MI355X (inputs resident, all stages, host sync at the end) vs the C oracle (kind: port) on the host cores",
           "host_threads": cores, "rows": rows}
    if out_path:
        with open(out_path, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
