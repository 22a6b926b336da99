"""bench.py's N > 1 leg with the PRODUCT on the device: two ranks (two processes, one device world each, shards from the device's own
proximity groups, shard guard armed) share the one GPU of a development box; the collectives run over gloo — RCCL refuses two ranks
on one device — everything else is the path `torchrun --nproc-per-node N bench.py --gpus N` takes on an 8-GPU node.  The world
assembled by the all-gather must equal the unsharded world stepped on the same device, bit for bit."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    args = bench.parse_args(["--gpus", str(world), "--steps", "40", "--warmup", "20", "--workload", "grid:4x6", "--no-cpu-baseline", "--backend", "gloo", "--share-gpu",
                             "--roofline-steps", "8"])
    out, gathered = bench.run(args)
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), pos=gathered[0], vel=gathered[1])
        with open(os.path.join(out_dir, "line.json"), "w") as f:
            json.dump(out, f)


def test_two_ranks_share_one_gpu_and_assemble_the_whole_world(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    sys.path.insert(0, ROOT)
    from rapier_amd import PhysicsWorld, scenes as S
    line = json.load(open(os.path.join(str(tmp_path), "line.json")))
    w = PhysicsWorld.from_scene(S.many_pyramids(rows=4, cols=6))
    w.step(20 + 40 + 8)                                         # warm-up + timed steps + the roofline leg's steps
    pos, vel = w.read_bodies()
    g = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    np.testing.assert_array_equal(g["pos"], pos)
    np.testing.assert_array_equal(g["vel"], vel)
    d = line["dist"]
    assert line["n_gpus"] == 2 and d["backend"] == "gloo" and d["world_size"] == 2 and d["gathered_bodies"] == 1 + 24 * 55 and line["finite"]
    assert d["shard_source"].startswith("device proximity groups discovered on rank 0 and broadcast (24 groups")   # rank 1 never built the whole world
    assert len(d["per_rank_step_paths"]) == 2 and all(p["fast"] + p["full"] > 0 for p in d["per_rank_step_paths"])
    assert line["config"]["total_cuboids"] == 24 * 55 and line["config"]["bodies_per_gpu"] == 12 * 55
    assert line["config"]["strong_scaling_anchor"] is None or "c4_world_steps_per_s_on_1_gpu" in line["config"]["strong_scaling_anchor"]
