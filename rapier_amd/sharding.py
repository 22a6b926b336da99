"""Multi-GPU sharding of independent contact islands.

The solve does not shard inside an island, but islands that only touch fixed geometry never couple
(fixed bodies are not island members, island_manager/persistent.rs:1).  Each rank owns a subset of
the dynamic bodies (whole islands), replicates every fixed body, and runs the full step on its own
MI355X with NO data-path collective; one all-gather of packed body state (13 f32 per body) over
RCCL/xGMI assembles the world state at readback.
"""
from __future__ import annotations

import numpy as np

from . import scenes as S


def bin_pack(sizes, world_size: int) -> np.ndarray:
    """Greedy longest-first bin packing of islands (by body count) onto ranks; returns rank per island."""
    sizes = np.asarray(sizes)
    order = np.argsort(-sizes, kind="stable")
    load = np.zeros(world_size, np.int64)
    out = np.zeros(len(sizes), np.int32)
    for i in order:
        r = int(np.argmin(load))
        out[i] = r
        load[r] += sizes[i]
    return out


def partition_scene(scene: S.Scene, body_rank: np.ndarray, rank: int):
    """Sub-scene of `rank`: its dynamic bodies + every fixed body (replicated), with colliders.
    Returns (sub_scene, global_index_of_local_body)."""
    sub = S.Scene(name=f"{scene.name}@{rank}", gravity=scene.gravity, params=scene.params.copy())
    local_of = {}
    global_ids = []
    for gi, b in enumerate(scene.bodies):
        if int(b["body_type"]) != S.BODY_DYNAMIC or int(body_rank[gi]) == rank:
            local_of[gi] = len(sub.bodies)
            sub.bodies.append(b)
            global_ids.append(gi)
    for c, p in zip(scene.colliders, scene.collider_parents):
        if p < 0:
            sub.colliders.append(c)
            sub.collider_parents.append(-1)
        elif p in local_of:
            sub.colliders.append(c)
            sub.collider_parents.append(local_of[p])
    for j in scene.joints:
        b1, b2 = int(j["body1"]), int(j["body2"])
        if b1 in local_of and b2 in local_of:
            jj = j.copy()
            jj["body1"], jj["body2"] = local_of[b1], local_of[b2]
            sub.joints.append(jj)
    return sub, np.asarray(global_ids, np.int64)


def many_pyramids_body_ranks(rows: int, cols: int, base_count: int, world_size: int) -> np.ndarray:
    """Island (= pyramid) -> rank for the many_pyramids generator; body 0 is the ground."""
    per = base_count * (base_count + 1) // 2
    ranks = bin_pack([per] * (rows * cols), world_size)
    body_rank = np.zeros(1 + rows * cols * per, np.int32)
    body_rank[1:] = np.repeat(ranks, per)
    return body_rank


def all_gather_bodies(local_pos: np.ndarray, local_vel: np.ndarray, global_ids: np.ndarray, n_global: int,
                      dynamic_mask_local: np.ndarray, device=None):
    """All-gather packed body state (pos7 + vel6 = 13 f32 per body) from every rank and scatter it into
    arena order.  Uses torch.distributed (backend nccl = RCCL over xGMI on GPU, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    packed = np.concatenate([local_pos, local_vel], axis=1).astype(np.float32)
    own = dynamic_mask_local.astype(bool)
    counts = torch.tensor([int(own.sum())], dtype=torch.int64, device=device)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    maxn = int(max(int(c.item()) for c in all_counts))
    # body state travels as f32, the global ids as int64 in a gather of their own (an f32 is exact only below 2^24 bodies)
    buf = torch.zeros((maxn, 13), dtype=torch.float32, device=device)
    ids_buf = torch.full((maxn,), -1, dtype=torch.int64, device=device)
    if own.any():
        n_own = int(own.sum())
        buf[:n_own] = torch.from_numpy(packed[own]).to(buf.device)
        ids_buf[:n_own] = torch.from_numpy(np.ascontiguousarray(global_ids[own], np.int64)).to(buf.device)
    gathered = [torch.zeros_like(buf) for _ in range(world)]
    gathered_ids = [torch.zeros_like(ids_buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    dist.all_gather(gathered_ids, ids_buf)
    pos = np.zeros((n_global, 7), np.float32)
    vel = np.zeros((n_global, 6), np.float32)
    # fixed bodies are replicated: take them from the local copy
    fixed = ~own
    pos[global_ids[fixed]] = local_pos[fixed]
    vel[global_ids[fixed]] = local_vel[fixed]
    for r in range(world):
        n = int(all_counts[r].item())
        if n == 0:
            continue
        g = gathered[r][:n].cpu().numpy()
        ids = gathered_ids[r][:n].cpu().numpy()
        pos[ids] = g[:, :7]
        vel[ids] = g[:, 7:13]
    return pos, vel


def column_shard_global_ids(rows: int, cols_per_rank: int, base_count: int, world_size: int, rank: int) -> np.ndarray:
    """Global body ids of rank `rank`'s bodies when a rows x (cols_per_rank * world_size) many_pyramids world is sharded by
    pyramid columns (bench.py --gpus N): local body 0 is the replicated ground (global 0); the generator emits pyramids row by
    row, so row r of the rank's local scene maps into row r of the global one at column offset rank * cols_per_rank."""
    per = base_count * (base_count + 1) // 2
    local = np.arange(rows * cols_per_rank * per)
    row, in_row = local // (cols_per_rank * per), local % (cols_per_rank * per)
    return np.concatenate([[0], 1 + row * (cols_per_rank * world_size * per) + rank * cols_per_rank * per + in_row]).astype(np.int64)


def island_shard(rows: int, cols: int, base_count: int, world_size: int, rank: int):
    """Shard of a rows x cols many_pyramids world for `rank`: the pyramids (= independent contact islands) are bin-packed by body
    count onto the ranks (`bin_pack`: 2,916 islands over 8 ranks = 364 or 365 each for BASELINE config C4), the ground is
    replicated.  Returns (pyramid mask for scenes.many_pyramids(pyramids=...), global body ids of the shard's bodies in the
    shard's own insertion order, global body count)."""
    per = base_count * (base_count + 1) // 2
    ranks = bin_pack([per] * (rows * cols), world_size)
    mask = ranks == rank
    owned = np.flatnonzero(mask)
    gids = np.concatenate([[0], (1 + owned[:, None] * per + np.arange(per)[None, :]).ravel()]).astype(np.int64)
    return mask, gids, 1 + rows * cols * per


# bench.py --gpus N: the pyramid grid each N steps (C4-shaped: ~364.5 islands per GPU; N = 8 is exactly BASELINE config C4)
C4_GRIDS = {1: (54, 54), 2: (27, 27), 4: (27, 54), 8: (54, 54)}


# ---- shards from the device's own proximity groups (no generator knowledge) -------------------------------------------------------------
def shards_from_groups(groups: np.ndarray, world_size: int):
    """groups[i] = proximity group of body i (PhysicsWorld.proximity_groups: -1 for fixed bodies) -> (rank of every body, -1 for fixed
    ones; number of groups): whole groups are bin-packed by body count (`bin_pack`), so no pair or joint ever spans two ranks."""
    groups = np.asarray(groups)
    ids, inv, counts = np.unique(groups[groups >= 0], return_inverse=True, return_counts=True)
    rank_of_group = bin_pack(counts, world_size)
    body_rank = np.full(len(groups), -1, np.int32)
    body_rank[groups >= 0] = rank_of_group[inv]
    return body_rank, len(ids)


def body_boxes(scene: S.Scene):
    """Conservative world AABB of every body (centre of every collider +- its bounding radius): (min[n, 3], max[n, 3])."""
    nb = len(scene.bodies)
    lo = np.full((nb, 3), np.inf, np.float64); hi = np.full((nb, 3), -np.inf, np.float64)
    pos = np.array([b["translation"] for b in scene.bodies], np.float64).reshape(nb, 3)
    for c, p in zip(scene.colliders, scene.collider_parents):
        if p < 0:
            continue
        he = np.asarray(c["half_extents"], np.float64)
        shape = int(c["shape"])
        if shape == S.SHAPE_BALL:
            r = he[0]
        elif shape == S.SHAPE_CAPSULE:
            r = he[0] + he[1]
        elif shape == S.SHAPE_CUBOID:
            r = float(np.linalg.norm(he))
        else:
            continue  # (half-spaces sit on fixed bodies: never boxed)
        r += float(np.linalg.norm(np.asarray(c["translation"], np.float64)))  # the collider's offset from the body, whatever the rotation
        lo[p] = np.minimum(lo[p], pos[p] - r); hi[p] = np.maximum(hi[p], pos[p] + r)
    return lo, hi


def guard_boxes(scene: S.Scene, groups: np.ndarray, body_rank: np.ndarray, rank: int, clearance: float = 0.25):
    """One box per proximity group that lives on ANOTHER rank (the union of its bodies' boxes, inflated by `clearance`): the input of
    PhysicsWorld.set_shard_guard for the shard of `rank`."""
    lo, hi = body_boxes(scene)
    groups = np.asarray(groups); body_rank = np.asarray(body_rank)
    foreign = (groups >= 0) & (body_rank != rank) & np.isfinite(lo[:, 0])
    ids, inv = np.unique(groups[foreign], return_inverse=True)
    bmin = np.full((len(ids), 3), np.inf); bmax = np.full((len(ids), 3), -np.inf)
    np.minimum.at(bmin, inv, lo[foreign]); np.maximum.at(bmax, inv, hi[foreign])
    return (bmin - clearance).astype(np.float32), (bmax + clearance).astype(np.float32)


def proximity_groups_from_scene(scene: S.Scene, margin: float = 0.1) -> np.ndarray:
    """CPU stand-in for PhysicsWorld.proximity_groups on small scenes (tests, the oracle adapter of the gloo bench test): components of
    the non-fixed bodies whose conservative boxes (inflated by `margin`) overlap, plus joints.  O(n^2): small worlds only."""
    lo, hi = body_boxes(scene)
    nb = len(scene.bodies)
    dyn = np.array([int(b["body_type"]) != S.BODY_FIXED for b in scene.bodies]) & np.isfinite(lo[:, 0])
    parent = np.arange(nb)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]; x = parent[x]
        return x
    idx = np.nonzero(dyn)[0]
    for k, i in enumerate(idx):
        rest = idx[k + 1:]
        ov = np.all((lo[i] - margin <= hi[rest] + margin) & (lo[rest] - margin <= hi[i] + margin), axis=1)
        for j in rest[ov]:
            a, b = find(i), find(j)
            if a != b:
                parent[max(a, b)] = min(a, b)
    for j in scene.joints:
        b1, b2 = int(j["body1"]), int(j["body2"])
        if dyn[b1] and dyn[b2]:
            a, b = find(b1), find(b2)
            if a != b:
                parent[max(a, b)] = min(a, b)
    return np.array([find(i) if dyn[i] else -1 for i in range(nb)], np.int32)
