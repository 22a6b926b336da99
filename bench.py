#!/usr/bin/env python
"""bench.py — physics steps/sec on b3d_many_pyramids (BASELINE.json metric), MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one PhysicsPipeline::step() of the resident world (inputs already in HBM).  N = 1: the
workload is BASELINE config C3 = b3d_many_pyramids (14x14 pyramids, 10,780 cuboids, M = 28,420
solver manifolds).  N > 1: weak scaling over independent contact islands: the node
holds a 14 x 14N pyramid world, rank r owns pyramid columns [14r, 14r+14) (ground replicated) and
steps them with NO data-path collective; `value` = N * K / t_max = many_pyramids-sized world-steps
per second across the node.  One all-gather of packed body state (RCCL) after the timed region
assembles the world on every rank (readback, not timed).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_step(M: int, N: int, substeps: int = 4) -> float:
    """B_solve(step) = S * [ M * (1100 + 476 + 1008) + N * 224 ] bytes."""
    return substeps * (M * (1100.0 + 476.0 + 1008.0) + N * 224.0)


def cpu_baseline(steps: int = 400, warmup: int = 60):
    """The C oracle (a scalar port of the reference algorithm, OpenMP over the same body-disjoint colour
    stages the reference hands to its rayon pool) timed on the host cores on a bounded sample of the
    same workload.  `value` = the multi-threaded rate; the single-thread rate is reported beside it."""
    import oracle_ffi
    from rapier_amd import scenes as S
    cores = max(1, min(os.cpu_count() or 1, 32))
    out = {}
    for threads in (1, cores):
        oracle_ffi.set_threads(threads)
        w = oracle_ffi.OracleWorld(S.many_pyramids())
        w.step(warmup)
        n = steps if threads > 1 else steps // 4
        t = time.perf_counter()
        w.step(n)
        out[threads] = n / (time.perf_counter() - t)
    oracle_ffi.set_threads(1)
    return {"value": out[cores], "unit": "steps/s", "cores": cores, "kind": "port", "single_thread_value": out[1],
            "sample": f"{steps} steps of b3d_many_pyramids (10,780 cuboids) after {warmup} warm-up steps, oracle/librapier_oracle.so "
                      f"(C restatement, OpenMP, {cores} threads; {steps // 4} steps on 1 thread)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=200)
    args = ap.parse_args()

    import torch  # first: the HIP runtime that torch loads is the one the library binds to
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    import numpy as np
    from rapier_amd import PhysicsWorld, scenes as S, sharding

    if world > 1:
        scene = S.many_pyramids(rows=14, cols=14 * world, col_range=(14 * rank, 14 * rank + 14))
        workload = f"b3d_many_pyramids weak-scaled: 14x{14 * world} pyramids, rank owns 14x14 (10,780 cuboids/GPU)"
    else:
        scene = S.many_pyramids()
        workload = "b3d_many_pyramids (14x14 pyramids, 10,780 cuboids, f32, dt=1/60, 4 substeps)"
    w = PhysicsWorld.from_scene(scene, device=local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    w.step(args.warmup)
    w.sync()
    barrier()
    t0 = time.perf_counter()
    w.step(args.steps)
    w.sync()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    counters = w.counters()
    M, Nd = counters["num_manifolds"], counters["num_dynamic_bodies"]

    # roofline leg: k_island_solve (one launch per step = the whole TGS velocity-solve loop of every
    # island) timed with hipEvents on the world's own stream, right after the timed region
    w.enable_timers(True)
    w.step(args.roofline_steps)
    w.sync()
    loop_ms, nmeas = w.solver_loop_time_ms()
    tc = w.counters()
    w.enable_timers(False)
    bytes_step = algorithmic_bytes_per_step(M, Nd, int(scene.params["num_solver_iterations"]))
    achieved = bytes_step / (loop_ms * 1e-3) / 1e9 if loop_ms > 0 else 0.0
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "r01_many_pyramids_hbm_traffic.json")
    if world == 1 and os.path.exists(tfile):  # PMC passes are separate rocprofv3 runs (tools/gpu_profile.sh)
        with open(tfile) as f:
            traffic = json.load(f).get("k_island_solve_hbm_bytes_per_launch")

    # readback (not timed): assemble the world state with one all-gather over RCCL/xGMI
    pos, vel = w.read_bodies()
    finite = bool(np.isfinite(pos).all())
    if dist is not None:
        dyn = np.array([int(b["body_type"]) == S.BODY_DYNAMIC for b in scene.bodies])
        per = 55
        gids = sharding.column_shard_global_ids(14, 14, 10, world, rank)
        gpos, _ = sharding.all_gather_bodies(pos, vel, gids, 1 + 14 * 14 * world * per, dyn, device="cuda")
        finite = finite and bool(np.isfinite(gpos).all())

    if rank == 0:
        out = {
            "metric": "physics steps/sec (whole node), b3d_many_pyramids 3D f32",
            "value": world * args.steps / dt,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (closed-form scene generator, no RNG)",
            "config": {"workload": workload, "bodies_per_gpu": Nd, "solver_manifolds_per_gpu": M,
                       "colors": counters["num_colors"], "value_definition": "n_gpus * steps / max-over-ranks time"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_island_solve (TGS velocity-solve loop: 4 substeps x [warmstart, biased, relaxed sweeps] of all 196 islands, 1 launch/step)",
                         "algorithmic_bytes_per_launch": bytes_step, "kernel_ms_per_launch": loop_ms, "measured_launches": nmeas,
                         "traffic_note": "HBM bytes/launch from rocprofv3 PMC passes (profiles/); constraints live in VGPRs/LDS, so traffic << algorithmic bytes",
                         "stage_ms": {k: tc[k] for k in ("collision_detection_ms", "velocity_resolution_ms", "velocity_update_ms")},
                         "path": {k: tc[k] for k in ("fast_steps", "full_steps", "replayed_steps")}},
            "finite": finite,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
