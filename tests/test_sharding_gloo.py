"""world_size-2 gloo test of the N>1 path: islands sharded over ranks, each rank steps
its shard independently (no data-path collective), one all-gather assembles the world; the result
must equal the unsharded world bit for bit because islands never couple.  On CPU the per-rank
stepper is the oracle (the HIP path needs a GPU; the sharding/gather logic is what is under test)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, steps, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from rapier_amd import scenes as S, sharding
    from oracle_ffi import OracleWorld
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = S.many_pyramids(rows=2, cols=3)
    br = sharding.many_pyramids_body_ranks(2, 3, 10, world)
    sub, gids = sharding.partition_scene(full, br, rank)
    w = OracleWorld(sub)
    w.step(steps)
    pos, vel = w.read()
    dyn = np.array([int(b["body_type"]) == S.BODY_DYNAMIC for b in sub.bodies])
    gpos, gvel = sharding.all_gather_bodies(pos, vel, gids, len(full.bodies), dyn)
    dist.barrier()
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), pos=gpos, vel=gvel)
    dist.destroy_process_group()


def test_sharded_islands_equal_unsharded_world(tmp_path):
    steps = 25
    port = _free_port()
    mp.spawn(_worker, args=(2, port, steps, str(tmp_path)), nprocs=2, join=True)
    from rapier_amd import scenes as S
    from oracle_ffi import OracleWorld
    w = OracleWorld(S.many_pyramids(rows=2, cols=3))
    w.step(steps)
    pos, vel = w.read()
    g = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    np.testing.assert_array_equal(g["pos"], pos)
    np.testing.assert_array_equal(g["vel"], vel)
