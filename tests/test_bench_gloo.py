"""bench.py's N > 1 control flow on CPU: two gloo ranks run `bench.run` itself (workload construction by island bin-packing,
barrier-bracketed timed region, max-over-ranks time, cuboid all-reduce, int64-id all-gather, the JSON line) with the CPU
oracle standing in for the HIP world — only the nccl / cuda branches stay unexecuted.  The assembled world must equal the
unsharded world bit for bit (islands never couple)."""
import json
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleAdapter:
    """The slice of PhysicsWorld's interface bench.run uses, over the oracle."""

    built = []          # body counts of the worlds this process built (the whole-world discovery build happens on rank 0 only)

    def __init__(self, scene, device):
        from oracle_ffi import OracleWorld
        from rapier_amd import scenes as S
        OracleAdapter.built.append(len(scene.bodies))
        self._w = OracleWorld(scene)
        self._nd = sum(1 for b in scene.bodies if int(b["body_type"]) == S.BODY_DYNAMIC)

    def step(self, n):
        self._w.step(n)

    def sync(self):
        pass

    def read_bodies(self):
        return self._w.read()

    def counters(self):
        st = self._w.stats()
        return {"num_manifolds": st["num_active_manifolds"], "num_dynamic_bodies": self._nd, "num_colors": st["num_colors_used"],
                "fast_steps": 0, "full_steps": st.get("steps", 0), "replayed_steps": 0}


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    args = bench.parse_args(["--gpus", str(world), "--steps", "6", "--warmup", "3", "--workload", "grid:2x3", "--no-cpu-baseline"])
    out, gathered = bench.run(args, make_world=OracleAdapter, backend="gloo", use_cuda=False)
    with open(os.path.join(out_dir, f"built{rank}.json"), "w") as f:
        json.dump(OracleAdapter.built, f)
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), pos=gathered[0], vel=gathered[1])
        with open(os.path.join(out_dir, "line.json"), "w") as f:
            json.dump(out, f)


def test_bench_control_flow_two_ranks(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    sys.path.insert(0, ROOT)
    from rapier_amd import scenes as S
    from oracle_ffi import OracleWorld
    w = OracleWorld(S.many_pyramids(rows=2, cols=3))
    w.step(9 )
    pos, vel = w.read()
    g = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    np.testing.assert_array_equal(g["pos"], pos)
    np.testing.assert_array_equal(g["vel"], vel)
    line = json.load(open(os.path.join(str(tmp_path), "line.json")))
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["warmup"] == 3 and line["scaling"] == "weak" and line["unit"] == "steps/s"
    assert line["config"]["total_cuboids"] == 6 * 55 and line["config"]["bodies_per_gpu"] == 3 * 55 and line["finite"]
    # whole-job value: C3-equivalent steps/s of all ranks, consistent with the reported time
    assert abs(line["value"] - (330 / 10780) * 6 / (line["ms_per_step"] * 6e-3)) <= 1e-6 * line["value"]
    # discovery: rank 0 alone built the whole world (1 + 6*55 bodies), the groups reached rank 1 by broadcast; both ranks then built their shard
    built = [json.load(open(os.path.join(str(tmp_path), f"built{r}.json"))) for r in range(2)]
    assert built[0] == [331, 1 + 3 * 55] and built[1] == [1 + 3 * 55]
    assert line["dist"]["shard_source"].startswith("device proximity groups discovered on rank 0 and broadcast (6 groups")
    assert [set(p) for p in line["dist"]["per_rank_step_paths"]] == [{"fast", "full", "replayed"}] * 2
    assert abs(line["config"]["sharded_world_steps_per_s"] - 1e3 / line["ms_per_step"]) <= 1e-6 * line["config"]["sharded_world_steps_per_s"]


def _worker_forced(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    args = bench.parse_args(["--gpus", "1", "--steps", "4", "--warmup", "2", "--workload", "grid:2x2", "--no-cpu-baseline", "--force-dist"])
    out, gathered = bench.run(args, make_world=OracleAdapter, backend="gloo", use_cuda=False)
    with open(os.path.join(out_dir, "forced.json"), "w") as f:
        json.dump({"line": out, "n": int(gathered[0].shape[0])}, f)


def test_force_dist_runs_the_collective_leg_on_one_rank(tmp_path):
    """`bench.py --gpus 1 --force-dist` = the N > 1 code path (process group, barrier, all-reduce, all-gather of body state) with a
    single rank: what `gpurun` runs on one MI355X with the nccl backend (the log is kept under profiles/)"""
    mp.spawn(_worker_forced, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    rec = json.load(open(os.path.join(str(tmp_path), "forced.json")))
    d = rec["line"]["dist"]
    assert d["shard_source"].startswith("device proximity groups discovered on rank 0 and broadcast (4 groups")   # shards derived from the (stand-in) device's groups
    assert {k: d[k] for k in ("backend", "world_size", "forced", "gathered_bodies")} == {"backend": "gloo", "world_size": 1, "forced": True, "gathered_bodies": 1 + 4 * 55}
    assert rec["line"]["n_gpus"] == 1 and rec["n"] == 221 and rec["line"]["finite"]


def test_c4_shards_cover_the_world():
    """BASELINE config C4 over 8 ranks: 2,916 islands -> 364 or 365 per rank, disjoint and complete, ids as int64."""
    sys.path.insert(0, ROOT)
    from rapier_amd import sharding
    rows, cols = sharding.C4_GRIDS[8]
    seen = np.zeros(1 + rows * cols * 55, np.int32)
    counts = []
    for r in range(8):
        mask, gids, n = sharding.island_shard(rows, cols, 10, 8, r)
        assert gids.dtype == np.int64 and n == 160381 and gids[0] == 0
        counts.append(int(mask.sum()))
        seen[gids[1:]] += 1
    assert sorted(set(counts)) == [364, 365] and sum(counts) == 2916
    assert (seen[1:] == 1).all()
    for n_ranks, grid in sharding.C4_GRIDS.items():
        if n_ranks > 1:
            assert abs(grid[0] * grid[1] / n_ranks - 364.5) < 1e-9


def test_shards_from_proximity_groups_and_guard_boxes():
    """the N > 1 leg no longer needs generator knowledge: groups (here from the CPU stand-in; on the GPU from
    rp_bodies_proximity_group) -> whole groups bin-packed over ranks -> per-rank sub-scenes + the boxes of the OTHER ranks' groups"""
    sys.path.insert(0, ROOT)
    from rapier_amd import scenes as S, sharding
    sc = S.many_pyramids(3, 4)
    groups = sharding.proximity_groups_from_scene(sc)
    assert groups[0] == -1 and len(set(groups[groups >= 0].tolist())) == 12 and (np.bincount(groups[groups >= 0])[np.unique(groups[groups >= 0])] == 55).all()
    body_rank, ng = sharding.shards_from_groups(groups, 3)
    assert ng == 12 and body_rank[0] == -1 and sorted(np.bincount(body_rank[body_rank >= 0]).tolist()) == [220, 220, 220]
    for g in np.unique(groups[groups >= 0]):
        assert len(set(body_rank[groups == g].tolist())) == 1           # a group never spans two ranks
    seen = np.zeros(len(sc.bodies), np.int32)
    lo, hi = sharding.body_boxes(sc)
    for r in range(3):
        sub, gids = sharding.partition_scene(sc, body_rank, r)
        seen[gids[1:]] += 1
        bmin, bmax = sharding.guard_boxes(sc, groups, body_rank, r)
        assert bmin.shape == (8, 3) and (bmin < bmax).all()
        own = np.nonzero(body_rank == r)[0]
        # no body of this rank starts inside a foreign box (the guard would otherwise fire on the first rewritten fat AABB)
        for i in own:
            assert not np.any(np.all((lo[i] <= bmax) & (bmin <= hi[i]), axis=1))
    assert (seen[1:] == 1).all()
