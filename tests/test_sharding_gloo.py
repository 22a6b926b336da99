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


# ---- migration between ranks (sharding.ShardSet): two gloo ranks, one shard each, the oracle in the device's place ----------------------
class _OracleShard:
    """The slice of PhysicsWorld that sharding.ShardSet uses, over the oracle — with the shard guard restated on the host: a dynamic body
    whose box (centre +- the half diagonal of its colliders + the fat margin) overlaps a foreign box is a hit."""

    def __init__(self, scene, rank):
        from oracle_ffi import OracleWorld
        from rapier_amd import scenes as S, sharding
        self._S, self._w = S, OracleWorld(scene)
        self._half = [np.zeros(3) for _ in scene.bodies]               # the tight box of the (unrotated) cuboids, like the device's fat AABBs
        for c, p in zip(scene.colliders, scene.collider_parents):
            if p >= 0:
                self._half[p] = np.maximum(self._half[p], np.asarray(c["half_extents"], np.float64))
        self._dyn = [int(b["body_type"]) == S.BODY_DYNAMIC for b in scene.bodies]
        self._gone = set()
        self._boxes = None

    def step(self, n): self._w.step(n)
    def read_bodies(self): return self._w.read()
    def body_handles(self): return np.arange(self._w.n, dtype=np.uint64)
    def set_shard_guard(self, lo, hi): self._boxes = None if lo is None else (np.asarray(lo, np.float64), np.asarray(hi, np.float64))

    def take_shard_guard_hits(self):
        if self._boxes is None:
            return np.zeros(0, np.uint64)
        pos, _ = self._w.read()
        lo, hi = self._boxes
        out = []
        for i in range(self._w.n):
            if i >= len(self._dyn) or not self._dyn[i] or i in self._gone:
                continue
            a, b = pos[i][:3] - self._half[i] - 0.05, pos[i][:3] + self._half[i] + 0.05
            if np.any(np.all((a <= hi) & (lo <= b), axis=1)):
                out.append(i)
        return np.asarray(out, np.uint64)

    def remove_body(self, handles):
        for h in handles:
            self._w.remove_body(int(h)); self._gone.add(int(h))

    def insert_body(self, body):
        from oracle_ffi import lib
        b = int(lib().ro_add_body(self._w._w, np.array([body], self._S.BODY_DTYPE).ctypes.data))
        while len(self._dyn) <= b:
            self._dyn.append(False); self._half.append(np.zeros(3))
        self._dyn[b] = True; self._gone.discard(b); self._half[b] = np.zeros(3)
        return b

    def insert_collider(self, col, parent):
        from oracle_ffi import lib
        c = int(lib().ro_add_collider(self._w._w, np.array([col], self._S.COLLIDER_DTYPE).ctypes.data, int(parent)))
        self._half[int(parent)] = np.maximum(self._half[int(parent)], np.asarray(col["half_extents"], np.float64))
        return c


def _migration_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from rapier_amd import scenes as S, sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def exchange(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    sc = S.many_pyramids(1, 2)
    groups = sharding.proximity_groups_from_scene(sc)
    body_rank, _ = sharding.shards_from_groups(groups, world)
    shards = sharding.ShardSet(sc, world, lambda sub, r: _OracleShard(sub, r), body_rank=body_rank, groups=groups, local_ranks=[rank], exchange=exchange)
    shards.step(5)
    top = max((i for i in range(len(sc.bodies)) if body_rank[i] == 0), key=lambda i: float(sc.bodies[i]["translation"][1]))
    other_x = np.mean([float(sc.bodies[i]["translation"][0]) for i in range(len(sc.bodies)) if body_rank[i] == 1])
    toward = float(np.sign(other_x - float(sc.bodies[top]["translation"][0])))
    if rank == 0:
        shards.worlds[0]._w.set_vel(shards.handle[0][top], (toward * 9.0, 6.0, 0.0), (0.0, 0.0, 0.0))
    shards.step(120)
    pos, vel = shards.read_bodies()
    if rank == 0:
        np.savez(os.path.join(out_dir, "migrated.npz"), pos=pos, vel=vel, owner=shards.owner, migrations=shards.migrations, top=top, toward=toward)
    dist.barrier()
    dist.destroy_process_group()


def test_a_thrown_cube_migrates_between_two_gloo_ranks(tmp_path):
    """the distributed form of tests/test_gpu_migration.py: every rank holds ONE shard, the hits / boxes / body states travel through
    all_gather_object, and both ranks keep the same ownership tables"""
    mp.spawn(_migration_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    from rapier_amd import scenes as S
    from oracle_ffi import OracleWorld
    g = np.load(os.path.join(str(tmp_path), "migrated.npz"))
    top = int(g["top"])
    assert int(g["migrations"]) == 1 and int(g["owner"][top]) == 1
    sc = S.many_pyramids(1, 2)
    w = OracleWorld(sc)
    w.step(5); w.set_vel(top, (float(g["toward"]) * 9.0, 6.0, 0.0), (0.0, 0.0, 0.0)); w.step(120)
    pos, vel = w.read()
    assert np.isfinite(g["pos"]).all() and np.abs(g["pos"][:, :3] - pos[:, :3]).max() < 0.05
