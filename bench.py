#!/usr/bin/env python
"""bench.py — physics steps/sec on b3d_many_pyramids (BASELINE.json metric), MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload auto|c3|c4|large_pyramid|joint_grid]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one PhysicsPipeline::step() of the resident world (inputs already in HBM).

N = 1 (default workload): BASELINE config C3 = b3d_many_pyramids (14x14 pyramids, 10,780 cuboids, M = 28,420 solver
manifolds); `value` = steps/s.

N > 1: independent contact islands sharded over the ranks, NO data-path collective.  The world is the
many_pyramids generator at BASELINE config C4's density: N = 8 is exactly C4 (54x54 = 2,916 pyramids = 160,380 cuboids,
364-365 islands per rank by `sharding.bin_pack`), N = 2 / 4 are 27x27 / 27x54 (the same ~364.5 islands per rank), so the
per-GPU work is fixed as N grows ("weak").  `value` keeps the metric's unit and workload: C3-equivalent steps/s =
(cuboids stepped by all ranks / 10,780) * K / t_max, i.e. whole-job cuboid-steps/s normalised to the b3d_many_pyramids
world of N = 1; the absolute rate of the sharded world (C4 steps/s at N = 8) is reported beside it as
`config.sharded_world_steps_per_s`.  `--workload c4 --gpus 1` steps all of C4 on one GPU (the strong-scaling reference
for the N = 8 line).  One all-gather of packed body state (RCCL) after the timed region assembles the world (readback, not timed).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
C3_CUBOIDS = 10780
import glob
# the PMC record of a workload = the latest round's profiles/rNN_<scene>_hbm_traffic.json (tools/gpu_profile.sh + tools/pmc_summary.py)
TRAFFIC_SCENE = {"c3": "many_pyramids", "large_pyramid": "large_pyramid", "joint_grid": "joint_grid"}
# sources whose change invalidates a recorded HBM-traffic measurement: the island kernel's for the metric workload; the whole
# global solver path (tiles, per-stage launches, joints) for the single-island / jointed scenes
_ISLAND_SOURCES = ["rapier_amd/csrc/rp_islands.hip", "rapier_amd/csrc/rp_island_stages.h", "rapier_amd/csrc/rp_sleep_observe.h", "rapier_amd/csrc/rp_lanepair.h", "rapier_amd/csrc/rp_constraint.h", "rapier_amd/csrc/rp_pairs.h", "rapier_amd/csrc/rp_world.h"]
_GLOBAL_SOURCES = _ISLAND_SOURCES[4:] + ["rapier_amd/csrc/rp_tiles.hip", "rapier_amd/csrc/rp_solver.hip", "rapier_amd/csrc/rp_global.h", "rapier_amd/csrc/rp_lanepair.h",
                                       "rapier_amd/csrc/rp_joints.h", "rapier_amd/csrc/rp_joints.hip", "rapier_amd/csrc/rp_flow.hip"]
KERNEL_SOURCES = {"c3": _ISLAND_SOURCES, "large_pyramid": _GLOBAL_SOURCES, "joint_grid": _GLOBAL_SOURCES}
# kernels of the TGS loop on the global path (what `velocity_update_ms` brackets minus assembly / write-back): their PMC bytes per
# step are summed into `solver_loop_hbm_bytes_per_step` by tools/pmc_summary.py
SOLVER_LOOP_KERNELS = ("k_joint_net_step", "k_tile_step", "k_tile_sweep", "k_ws_prepare", "k_increment_ws", "k_increment", "k_integrate", "k_stage", "k_tail", "k_global_flow",
                       "k_joint_update", "k_joint_sweep", "k_joint_tail")


def algorithmic_bytes_per_step(M: int, N: int, substeps: int = 4, joint_rows: int = 0) -> float:
    """B_solve(step) = S * [ M * (1100 + 476 + 1008) + R * 136 * 3 + N * 224 ] bytes — M solver manifolds, N solver bodies,
    R joint constraint rows (one 136-byte JointConstraint per row: written by the rebuild, read by the biased and by the relaxed
    sweep: b3d_joint_grid = 19,800 joints x 3 rows -> 24 MB per substep)."""
    return substeps * (M * (1100.0 + 476.0 + 1008.0) + joint_rows * 136.0 * 3.0 + N * 224.0)


def joint_rows_of(scene) -> int:
    """constraint rows the scene's impulse joints build per substep: one per locked axis, per limited and per motorised free axis"""
    pop = lambda x: bin(int(x) & 0x3f).count("1")  # noqa: E731
    rows = 0
    for j in scene.joints:
        locked = int(j["locked_axes"])
        rows += pop(locked) + pop(int(j["limit_axes"]) & ~locked) + pop(int(j["motor_axes"]) & ~locked)
    return rows


def kernel_code_sha(workload: str = "c3") -> str:
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES.get(workload, _ISLAND_SOURCES):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def recorded_traffic(workload: str = "c3"):
    """HBM bytes of the dominant kernel (per launch; c3) / of the solver loop's kernels (per step; the global-path scenes) from the
    separate rocprofv3 --pmc passes (tools/gpu_profile.sh writes the file with the hash of the kernel sources it measured).  A record
    taken on different kernel code is refused: traffic = null."""
    scene = TRAFFIC_SCENE.get(workload)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{scene}_hbm_traffic.json"))) if scene else []
    if not files:
        return None, f"no PMC record (profiles/rNN_{scene}_hbm_traffic.json)"
    with open(files[-1]) as f:
        rec = json.load(f)
    if rec.get("kernel_code_sha") != kernel_code_sha(workload):
        return None, f"stale PMC record refused ({os.path.basename(files[-1])}: kernel sources changed since {rec.get('kernel_code_sha')})"
    key = "k_island_solve_hbm_bytes_per_step" if workload == "c3" else "solver_loop_hbm_bytes_per_step"
    val = rec.get(key)
    if val is None and workload == "c3":
        val = rec.get("k_island_solve_hbm_bytes_per_launch")  # (records of rounds 1-5: one step per launch)
    return val, f"rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE per the guide), {os.path.basename(files[-1])}, same kernel sources"


def recorded_c4_anchor():
    """`--workload c4 --gpus 1` (all of C4 on ONE GPU) as last recorded under profiles/: the strong-scaling anchor quoted in the
    config of an N > 1 line (an N > 1 run cannot measure it itself)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_c4_160380_cuboids_1gpu.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            rec = json.loads(f.read().strip().splitlines()[-1])
        return {"file": os.path.basename(files[-1]), "c4_world_steps_per_s_on_1_gpu": rec["config"]["sharded_world_steps_per_s"], "ms_per_step": rec["ms_per_step"]}
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(steps: int = 400, warmup: int = 60):
    """The C oracle (a scalar port of the reference algorithm, OpenMP over the same body-disjoint colour
    stages the reference hands to its rayon pool) timed on the host cores on a bounded sample of the
    same workload.  `value` = the best multi-threaded rate of a short sweep over thread counts (threads pinned to cores, one per
    core: OMP_PLACES=cores, OMP_PROC_BIND=close — set before the OpenMP runtime starts); the single-thread rate is reported beside it."""
    os.environ.setdefault("OMP_PLACES", "cores"); os.environ.setdefault("OMP_PROC_BIND", "close"); os.environ.setdefault("OMP_WAIT_POLICY", "active")
    import oracle_ffi
    from rapier_amd import scenes as S
    ncpu = os.cpu_count() or 1
    counts = sorted({c for c in (8, 16, 32, 64) if c <= ncpu} | {min(ncpu, 32)})
    w = oracle_ffi.OracleWorld(S.many_pyramids())
    oracle_ffi.set_threads(min(ncpu, 32))
    w.step(warmup)
    out = {}
    for threads in [1] + counts:
        oracle_ffi.set_threads(threads)
        n = steps // 4
        w.step(4)   # (the team of this size is up and its pages are touched before the clock starts)
        t = time.perf_counter()
        w.step(n)
        out[threads] = n / (time.perf_counter() - t)
    oracle_ffi.set_threads(1)
    best = max(counts, key=lambda c: out[c])
    return {"value": out[best], "unit": "steps/s", "cores": best, "kind": "port", "single_thread_value": out[1],
            "thread_sweep": {str(c): round(out[c], 2) for c in counts},
            "sample": f"{steps // 4} steps of b3d_many_pyramids (10,780 cuboids) per thread count after {warmup} warm-up steps, oracle/librapier_oracle.so "
                      f"(C restatement, gcc -O3 -march=x86-64-v3 -ffp-contract=off, OpenMP, threads pinned one per core; best of {counts} threads = {best}; {steps // 4} steps on 1 thread)"}


def build_workload(name: str, world: int, rank: int):
    """(scene of this rank, description, global ids of its bodies or None, global body count, pyramid grid or None)"""
    from rapier_amd import scenes as S, sharding
    if name == "auto":
        name = "c3" if world == 1 else "c4"
    if name == "c3":
        if world != 1:
            raise SystemExit("--workload c3 is the single-GPU metric configuration")
        return S.many_pyramids(), "b3d_many_pyramids (14x14 pyramids, 10,780 cuboids, f32, dt=1/60, 4 substeps)", None, None, (14, 14)
    if name == "c4":
        rows, cols = sharding.C4_GRIDS.get(world, (54, 54))
        mask, gids, n_global = sharding.island_shard(rows, cols, 10, world, rank)
        scene = S.many_pyramids(rows, cols, pyramids=mask)
        desc = (f"b3d_many_pyramids at BASELINE config C4 density: {rows}x{cols} pyramids = {rows * cols * 55:,} cuboids"
                f"{' (= C4)' if (rows, cols) == (54, 54) else ''}, islands bin-packed over {world} rank(s): {int(mask.sum())} islands on rank {rank}")
        return scene, desc, gids, n_global, (rows, cols)
    if name.startswith("grid:"):  # grid:RxC — small sharded worlds for the CPU control-flow test
        rows, cols = (int(v) for v in name[5:].split("x"))
        mask, gids, n_global = sharding.island_shard(rows, cols, 10, world, rank)
        return S.many_pyramids(rows, cols, pyramids=mask), f"many_pyramids {rows}x{cols} sharded over {world}", gids, n_global, (rows, cols)
    if world != 1:
        raise SystemExit(f"--workload {name} is a single-island / single-GPU scene: replicas only (DESIGN.md §7)")
    if name == "large_pyramid":
        return S.large_pyramid(200), "b3d_large_pyramid (20,100 cuboids, one island)", None, None, None
    if name == "joint_grid":
        return S.joint_grid(100), "b3d_joint_grid (100x100 balls, 19,800 spherical joints)", None, None, None
    raise SystemExit(f"unknown workload {name}")


def run(args, make_world=None, backend: str = "nccl", use_cuda: bool = True):
    """The bench control flow.  `make_world(scene, device)` defaults to the HIP product (`PhysicsWorld.from_scene`); the CPU test of
    the N > 1 leg passes the oracle in its place with backend="gloo", so the only lines it cannot reach are the nccl / cuda ones."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch = None
    if use_cuda:
        import torch  # first: the HIP runtime that torch loads is the one the library binds to
    dist = None
    if world > 1 or args.force_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")  # (--force-dist without a launcher: a one-rank group on this host)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = getattr(args, "backend", None) or backend
        if use_cuda and getattr(args, "share_gpu", False):
            # a development box with ONE GPU: the ranks share it (the sharded path runs for real — every rank a world of its own on the
            # device — only the collective cannot be RCCL, which refuses two ranks on one device: --backend gloo)
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        if use_cuda and backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            if use_cuda:
                torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif use_cuda and torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    coll_on_device = use_cuda and backend == "nccl"   # the collectives' tensors: device memory for RCCL, host memory for gloo

    import numpy as np
    from rapier_amd import scenes as S, sharding
    if make_world is None:
        from rapier_amd import PhysicsWorld
        make_world = PhysicsWorld.from_scene

    wl = args.workload
    if args.force_dist and world == 1 and wl in ("auto", "c3"):
        wl = "grid:14x14"  # the C3 world through the sharded code path (one rank owns every island): global ids, all-gather at readback
    scene, workload, gids, n_global, grid = build_workload(wl, world, rank)
    guard, shard_source = None, None
    if gids is not None:
        shard_source = "generator (sharding.island_shard: the pyramids of the closed-form scene)"
        if args.shard_by == "device" and grid is not None:
            # the shards come from the device's own proximity groups of the WHOLE scene (built once per rank, stepped once, dropped):
            # whole groups are bin-packed over the ranks, and every rank's world is guarded against the boxes of the other ranks' groups
            # RANK 0 discovers (it builds and steps the whole world once, ~10 s for the 160,380 cuboids of C4) and BROADCASTS the group of
            # every body (4 B each); the other ranks only generate the closed-form scene on the host and cut their shard out of it.  One
            # whole-world build per job instead of one per rank; every rank cuts the world from the same array.  
            full = S.many_pyramids(grid[0], grid[1])
            groups, discovered, err = None, 0, ""
            if rank == 0 or dist is None:
                try:
                    wf = make_world(full, local_rank)
                    wf.step(1)
                    groups = np.ascontiguousarray(wf.proximity_groups() if hasattr(wf, "proximity_groups") else sharding.proximity_groups_from_scene(full), np.int32)
                    if hasattr(wf, "close"):
                        wf.close()
                    del wf
                    discovered = int(len(groups) == len(full.bodies))
                except Exception as e:  # noqa: BLE001 — the generator's shards are the fallback
                    err = f"{type(e).__name__}: {e}"
            if dist is not None:
                # every rank must cut the world the same way: a failed discovery sends ALL ranks to the generator's shards
                ok = torch.tensor([discovered], dtype=torch.int64, device="cuda" if coll_on_device else None)
                dist.broadcast(ok, src=0)
                discovered = int(ok.item())
                if discovered:
                    gt = torch.from_numpy(groups if rank == 0 else np.zeros(len(full.bodies), np.int32))
                    if coll_on_device:
                        gt = gt.cuda()
                    dist.broadcast(gt, src=0)
                    groups = gt.cpu().numpy()
            if discovered:
                body_rank, n_groups = sharding.shards_from_groups(groups, world)
                scene, gids = sharding.partition_scene(full, body_rank, rank)
                n_global = len(full.bodies)
                guard = sharding.guard_boxes(full, groups, body_rank, rank)
                shard_source = f"device proximity groups discovered on rank 0 and broadcast ({n_groups} groups of the whole scene, bin-packed; {len(guard[0])} foreign boxes guarded on rank {rank})"
                workload += f"; shards from the device's proximity groups ({int((body_rank == rank).sum()) // 55} islands on rank {rank})"
            else:
                shard_source = f"generator (device discovery failed on rank 0{': ' + err if err else ''})"
            del full
            if not discovered:
                scene, workload, gids, n_global, grid = build_workload(wl, world, rank)
                guard = None
    w = make_world(scene, local_rank)
    if guard is not None and len(guard[0]) and hasattr(w, "set_shard_guard"):
        w.set_shard_guard(*guard)
    dev = "cuda" if coll_on_device else None

    def barrier():
        if dist is not None:
            dist.barrier()
        if use_cuda:
            torch.cuda.synchronize()

    w.step(0)   # builds the device world (allocation, uploads) — construction, not stepping: it must not land in the timed region when --warmup is 0
    w.step(args.warmup)
    w.sync()
    barrier()
    t0 = time.perf_counter()
    w.step(args.steps)
    w.sync()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    counters = w.counters()
    M, Nd = counters["num_manifolds"], counters["num_dynamic_bodies"]
    total_cuboids = Nd
    if dist is not None:
        tn = torch.tensor([Nd], dtype=torch.int64, device=dev)
        dist.all_reduce(tn, op=dist.ReduceOp.SUM)
        total_cuboids = int(tn.item())

    # roofline leg: the dominant kernel timed with hipEvents on the world's own stream, right after the timed region
    roof = None
    if hasattr(w, "enable_timers"):
        w.enable_timers(True)
        c_before = w.counters()
        w.step(args.roofline_steps)
        w.sync()
        loop_ms, nmeas = w.solver_loop_time_ms()
        tc = w.counters()
        # steps one launch of the dominant kernel carries (round 6: k_island_solve_steps takes up to 32 fused steps; 1 for every other form)
        d_steps, d_launches = tc.get("fused_steps", 0) - c_before.get("fused_steps", 0), tc.get("fused_launches", 0) - c_before.get("fused_launches", 0)
        steps_per_launch = max(1, round(d_steps / d_launches)) if d_launches > 0 else 1
        w.enable_timers(False)
        wkey = {"auto": "c3" if world == 1 else "c4"}.get(args.workload, args.workload)
        jrows = joint_rows_of(scene)
        sequence_ms = None
        bytes_step = algorithmic_bytes_per_step(M, Nd, int(scene.params["num_solver_iterations"]), jrows)
        kernel_name = ("k_island_solve (TGS velocity-solve loop: 4 substeps x [warmstart, biased, relaxed sweeps] of every LDS-resident island; "
                       + (f"k_island_solve_steps: {steps_per_launch} fused steps per launch)" if steps_per_launch > 1 else "1 launch/step)"))
        if tc["velocity_update_ms"] > tc["velocity_resolution_ms"]:
            # single giant islands / jointed worlds (--workload large_pyramid, joint_grid): the TGS loop runs on the global path (one launch
            # per colour stage, or the dataflow launch), timed by the events around it — not one kernel, a launch sequence
            kernel_only_ms = loop_ms   # (steps whose TGS loop is ONE launch: the events around that launch alone — launch_step, rp_api_step.inc)
            loop_ms = tc["velocity_update_ms"]
            kernel_name = "global solver path (TGS loop as per-colour-stage launches or one dataflow launch; hipEvents around the sequence)"
            # round 6: a net of spherical joints without contacts runs its whole TGS loop as ONE launch (k_joint_net_step) on lean graphs,
            # and those graphs are what the events bracket here too (solver sequence of the lean graph: that launch, the write-back, k_ccd)
            # round 6: a tiled contact world (b3d_large_pyramid) runs its whole TGS loop as ONE launch too (k_tile_step, lean and full graphs);
            # the events bracket k_begin_generate, that launch, the write-back and k_ccd
            d_ts = tc.get("tile_step_steps", 0) - c_before.get("tile_step_steps", 0)
            if d_ts > 0 and kernel_only_ms > 0:
                # ... and since the loop IS one kernel, the roofline prices that kernel (as k_island_solve for C3: the figure rocprofv3's kernel
                # stats must agree with); the sequence the earlier rounds timed (k_begin_generate + loop + write-back + k_ccd) stays beside it
                sequence_ms = loop_ms
                loop_ms = kernel_only_ms
                kernel_name = (f"k_tile_step (the TGS loop of a step as ONE launch over the LDS tiles: prepare / increment / biased / relaxed sweeps of every substep as "
                               f"phases, neighbouring tiles' flags between them); {d_ts} of {args.roofline_steps} timed steps took it")
            d_jn = tc.get("joint_net_steps", 0) - c_before.get("joint_net_steps", 0)
            if d_jn > 0:
                kernel_name = (f"k_joint_net_step (the TGS loop of a step as one launch: every tile's joints in registers, grid barriers between sweeps) "
                               f"+ k_writeback_bodies + k_ccd; {d_jn} of {args.roofline_steps} timed steps took it")
        achieved = bytes_step / (loop_ms * 1e-3) / 1e9 if loop_ms > 0 else 0.0
        traffic, traffic_note = recorded_traffic(wkey) if (world == 1 and wkey in TRAFFIC_SCENE) else (None, "PMC records exist for the single-GPU workloads c3, large_pyramid, joint_grid")
        # `frac` prices the reference's ALGORITHMIC bytes against the HBM peak; `hbm_frac` is its twin for the bytes the
        # kernel really moved (PMC): the constraint set lives in registers / LDS, so the kernel is latency-bound, not bandwidth-bound
        hbm_frac = (traffic / (loop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and loop_ms > 0) else None
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": (traffic * steps_per_launch) if traffic else None, "traffic_per_step": traffic,
                "hbm_frac": hbm_frac,
                "kernel": kernel_name,
                # a LAUNCH of the dominant kernel = steps_per_launch steps: bytes, time and traffic per launch are the per-step figures x that
                # (achieved = algorithmic_bytes_per_step / kernel_ms_per_step either way); measured_launches = steps the events covered
                "steps_per_launch": steps_per_launch, "algorithmic_bytes_per_step": bytes_step, "kernel_ms_per_step": loop_ms,
                "algorithmic_bytes_per_launch": bytes_step * steps_per_launch, "kernel_ms_per_launch": loop_ms * steps_per_launch, "measured_launches": nmeas,
                # what `kernel_ms_per_launch` was measured on (rp_api_step.inc launch_step with timers on): C3 / C4 = the SAME one-kernel fused
                # step the timed region runs (k_island_solve validating the step itself), launched directly between two hipEvents on the
                # world's stream; every timed step is waited for before the next is enqueued, so — unlike in the timed region — no launch
                # overlaps the tail of its predecessor (ms_per_step can therefore be a little BELOW kernel_ms_per_launch)
                "timed_form": ("one-kernel fused step (k_island_solve / k_island_solve_dense), launched directly between two hipEvents, one step at a time"
                               if tc["velocity_update_ms"] <= tc["velocity_resolution_ms"] else
                               ("the solver sequence of the lean step graph the timed region runs (k_joint_net_step, write-back, k_ccd) between two hipEvents, one step at a time"
                                if (tc.get("joint_net_steps", 0) - c_before.get("joint_net_steps", 0)) > 0 else
                                "k_tile_step alone between two hipEvents on the world's stream (the launches of a timed step go out directly), one step at a time"
                                if (tc.get("tile_step_steps", 0) - c_before.get("tile_step_steps", 0)) > 0 else
                                "the solver-loop launches of a full / lean step (tile sweeps or colour-stage launches) between two hipEvents, one step at a time")),
                "sequence_ms_per_step": sequence_ms,   # k_begin_generate + the loop + write-back + k_ccd (what rounds 3-5 priced for this workload)
                "frac_sequence": ((bytes_step / (sequence_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if sequence_ms else None),
                "traffic_note": traffic_note, "kernel_code_sha": kernel_code_sha(wkey), "joint_rows": jrows,
                "stage_ms": {k: tc[k] for k in ("collision_detection_ms", "velocity_resolution_ms", "velocity_update_ms")},
                "path": {k: tc[k] for k in ("fast_steps", "full_steps", "replayed_steps", "lean_steps", "joint_net_steps", "tile_step_steps") if k in tc}}

    # readback (not timed): assemble the world state with one all-gather over RCCL/xGMI
    pos, vel = w.read_bodies()
    finite = bool(np.isfinite(pos).all())
    gathered, readback = None, None
    if dist is not None:
        dyn = np.array([int(b["body_type"]) == S.BODY_DYNAMIC for b in scene.bodies])
        readback = "torch.distributed all_gather of host-packed rows (gloo leg / worlds without the native collective)"
        if backend == "nccl" and coll_on_device and hasattr(w, "shard_all_gather"):
            # the library's own collective: k_pack_bodies -> ncclAllGather on the world's stream -> D2H of the gathered rows;
            # torch.distributed only carries the 128-byte ncclUniqueId and the agreed row count
            from rapier_amd import ShardComm
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt = torch.frombuffer(bytearray(ShardComm.unique_id()), dtype=torch.uint8).to(dev)
            dist.broadcast(idt, src=0)
            comm = ShardComm(bytes(idt.cpu().numpy().tobytes()), world, rank, local_rank)
            rows = torch.tensor([int(dyn.sum())], dtype=torch.int64, device=dev)
            dist.all_reduce(rows, op=dist.ReduceOp.MAX)
            gp, gv, per = sharding.all_gather_bodies_native(w, comm, pos, vel, gids, n_global, dyn, max(1, int(rows.item())))
            gathered = (gp, gv)
            readback = f"rp_shard_all_gather: k_pack_bodies -> ncclAllGather (librccl, world stream) -> one D2H; rows per rank {per.tolist()}"
            comm.close()
        else:
            gathered = sharding.all_gather_bodies(pos, vel, gids, n_global, dyn, device=dev)
        finite = finite and bool(np.isfinite(gathered[0]).all())

    per_rank_paths = None
    if dist is not None: # which way every rank's steps went (fused fast steps / full steps / aborted-and-replayed ones): one small all-gather
        mine = torch.tensor([counters.get("fast_steps", 0), counters.get("full_steps", 0), counters.get("replayed_steps", 0)], dtype=torch.int64, device=dev)
        allp = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        per_rank_paths = [{"fast": int(t[0]), "full": int(t[1]), "replayed": int(t[2])} for t in (x.cpu() for x in allp)]
    out = None
    if rank == 0:
        is_metric_workload = world == 1 and args.workload in ("auto", "c3")
        single_scene = {"large_pyramid": "b3d_large_pyramid", "joint_grid": "b3d_joint_grid"}.get(args.workload)  # one-GPU scenes quoted in their own unit
        value = (total_cuboids / C3_CUBOIDS) * args.steps / dt
        if is_metric_workload or single_scene:
            value_def = "steps / max-over-ranks time"
        else:
            value_def = ("C3-equivalent steps/s = (cuboids stepped by all ranks / 10,780) * steps / max-over-ranks time; sharded_world_steps_per_s = steps/s of "
                         "the whole sharded world")
        out = {
            "metric": f"physics steps/sec (whole node), {single_scene or 'b3d_many_pyramids'} 3D f32",
            "value": args.steps / dt if (is_metric_workload or single_scene) else value,
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
            "config": {"workload": workload, "bodies_per_gpu": Nd, "solver_manifolds_per_gpu": M, "total_cuboids": total_cuboids,
                       "colors": counters["num_colors"], "pyramid_grid": list(grid) if grid else None,
                       "sharded_world_steps_per_s": args.steps / dt,
                       "value_definition": value_def,
                       # N > 1 steps BASELINE config C4's density (364.5 islands per GPU), N = 1 steps C3 (196 islands): the like-for-like
                       # anchor of the N > 1 lines is all of C4 on ONE GPU (`--workload c4 --gpus 1`), quoted from its last record
                       "strong_scaling_anchor": recorded_c4_anchor() if world > 1 else None},
            "roofline": roof,
            "finite": finite,
            "dist": None if dist is None else {"backend": backend, "world_size": world, "forced": bool(args.force_dist), "shard_source": shard_source,
                                               "gathered_bodies": None if gathered is None else int(gathered[0].shape[0]), "readback": readback,
                                               "per_rank_step_paths": per_rank_paths},
        }
        if not args.no_cpu_baseline and is_metric_workload:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out, gathered


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--workload", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=200)
    ap.add_argument("--shard-by", choices=("device", "generator"), default="device",
                    help="N > 1: shards from the device's proximity groups of the whole scene (default) or from the scene generator's pyramid list")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default=None, help="collective backend of the N > 1 leg (default: nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="N ranks on a box with fewer GPUs: rank r uses device r mod (device count); needs --backend gloo")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (init_process_group('nccl'), barrier, all-reduce, all-gather of body state) even with one rank")
    return ap.parse_args(argv)


if __name__ == "__main__":
    run(parse_args())
