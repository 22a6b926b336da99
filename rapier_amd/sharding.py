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
    if scene.subworlds:
        raise ValueError("partition_scene: a batch of sub-worlds (scenes.batch) cannot be cut by body rank — its sub-worlds overlap in space and a shard would lose their boundaries; give whole sub-worlds to each rank instead")
    sub = S.Scene(name=f"{scene.name}@{rank}", gravity=scene.gravity, params=scene.params.copy())
    sub.polyhedra = list(scene.polyhedra)   # registered point clouds keep their ids in every shard (SHAPE_CONVEX half_extents[0])
    sub.composites = list(scene.composites)  # ... and so do compounds, triangle meshes and height fields (SHAPE_COMPOUND / SHAPE_TRIMESH half_extents[0])
    local_of = {}
    global_ids = []
    for gi, b in enumerate(scene.bodies):
        if int(b["body_type"]) == S.BODY_DYNAMIC and int(body_rank[gi]) < 0:
            raise ValueError(f"dynamic body {gi} belongs to no shard (no collider gives it a box, or its group was not ranked)")
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


def all_gather_bodies_native(world_obj, comm, local_pos: np.ndarray, local_vel: np.ndarray, global_ids: np.ndarray, n_global: int,
                             dynamic_mask_local: np.ndarray, rows_per_rank: int):
    """The same read-back through the library's own collective (include/rapier_hip.h: rp_shard_all_gather): the shard's bodies are packed
    ON THE DEVICE, one ncclAllGather on the world's stream moves them between the GPUs, one D2H of the gathered rows follows the RCCL
    kernel — no body state touches the host before the collective.  `comm` = rapier_amd.ShardComm over the job's ranks,
    rows_per_rank = a value every rank agrees on, >= the largest shard's dynamic-body count.  Returns (pos, vel, rows_per_rank_seen)."""
    world_obj.set_global_ids(np.ascontiguousarray(global_ids, np.int64))
    pos = np.zeros((n_global, 7), np.float32)
    vel = np.zeros((n_global, 6), np.float32)
    fixed = ~dynamic_mask_local.astype(bool)   # fixed bodies are replicated: take them from the local copy
    pos[global_ids[fixed]] = local_pos[fixed]
    vel[global_ids[fixed]] = local_vel[fixed]
    per = world_obj.shard_all_gather(comm, rows_per_rank, pos, vel)
    return pos, vel, per


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
    def radius(shape, he, border):
        """bounding radius of a shape about the origin of its own frame"""
        he = np.asarray(he, np.float64)
        if shape == S.SHAPE_BALL:
            r = he[0]
        elif shape == S.SHAPE_CAPSULE:
            r = he[0] + he[1]
        elif shape in (S.SHAPE_CUBOID, S.SHAPE_ROUND_CUBOID):
            r = float(np.linalg.norm(he))
        elif shape in (S.SHAPE_CYLINDER, S.SHAPE_CONE, S.SHAPE_ROUND_CYLINDER, S.SHAPE_ROUND_CONE):
            r = float(np.hypot(he[0], he[1]))   # (half height, radius): the rim is the farthest point from the centre
        elif shape in (S.SHAPE_CONVEX_POLYHEDRON, S.SHAPE_ROUND_CONVEX_POLYHEDRON):
            r = float(np.linalg.norm(np.asarray(scene.polyhedra[int(he[0])][0], np.float64), axis=1).max())
        elif shape in (S.SHAPE_COMPOUND, S.SHAPE_TRIMESH):
            comp = scene.composites[int(he[0])]
            if comp[0] == "compound":      # the farthest part: its offset in the compound's frame + its own radius
                r = max(float(np.linalg.norm(np.asarray(q["translation"], np.float64))) + radius(int(q["shape"]), q["half_extents"], float(q["border_radius"])) for q in comp[1])
            elif comp[0] == "trimesh":
                r = float(np.linalg.norm(np.asarray(comp[1], np.float64), axis=1).max())
            else:                          # height field: the unit square scaled by (sx, sy, sz), centred on the origin
                h, sc = np.asarray(comp[1], np.float64), np.asarray(comp[2], np.float64)
                r = float(np.linalg.norm([0.5 * sc[0], np.abs(h).max() * sc[1], 0.5 * sc[2]]))
        else:
            raise ValueError(f"body_boxes: unknown shape {shape}")
        return r + border
    nb = len(scene.bodies)
    lo = np.full((nb, 3), np.inf, np.float64); hi = np.full((nb, 3), -np.inf, np.float64)
    pos = np.array([b["translation"] for b in scene.bodies], np.float64).reshape(nb, 3)
    for c, p in zip(scene.colliders, scene.collider_parents):
        if p < 0:
            continue
        shape = int(c["shape"])
        if shape == S.SHAPE_HALFSPACE:
            continue  # (half-spaces sit on fixed or kinematic bodies and have no finite box)
        r = radius(shape, c["half_extents"], float(c["border_radius"]) if "border_radius" in c.dtype.names else 0.0)
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


# ---- migration: a body of one shard reaches a box of another ----------------------------------------------------------
class ShardSet:
    """The shards of ONE scene as a caller holds them when the guard of one of them fires: per rank a world, the sub-scene it was built
    from and the global id of each of its rows.  `make_world(scene, rank)` builds a world (PhysicsWorld.from_scene on the device; the
    CPU tests pass an oracle adapter).  Every rank may live in this process (one GPU, several worlds: tests, a single-process driver)
    or one per process: then `exchange` is `torch.distributed.all_gather_object` over the job and `local_ranks` names the rank(s) held
    here — the bookkeeping below is a pure function of what is exchanged, so every process ends up with the same tables.

    step(n) steps every local world; when a guard fired, the bodies it caught are taken (rp_world_shard_guard_take_hits), their whole
    proximity groups move to the rank whose group they reached — the smaller side moves, as the reference's note on cross-shard pairs
    says — and every rank gets fresh guard boxes from the current poses.  A migrated body keeps pose and velocities; the contact
    warm-start of its pairs is not carried (it has none with the shard it leaves by the time it reaches the clearance of another)."""

    def __init__(self, scene: S.Scene, n_ranks: int, make_world, body_rank=None, groups=None, local_ranks=None, exchange=None, clearance: float = 0.25,
                 check_every: int = 1):
        self.scene, self.n_ranks, self.make_world = scene, n_ranks, make_world
        # A guard hit waits at most `check_every` steps for us.  Three allowances make that sound without one world-wide clearance (which
        # would merge shards that merely stand close): the DEVICE inflates every tested body box by |linvel| x horizon
        # (rp_world_set_shard_guard_horizon, horizon = dt x check_every); every group's box carries base + ITS top speed x horizon; and the
        # boxes are refreshed once the fastest body of the job (rp_world_max_linear_speed, exchanged with the hits) may have drifted
        # half the base clearance away from where the boxes were taken.
        self.clearance = float(clearance)
        self._dt = float(scene.params["dt"])
        self.check_every, self._since_check = max(1, int(check_every)), 0   # guard hits are looked at (one exchange) every so many steps
        self.local_ranks = list(range(n_ranks)) if local_ranks is None else list(local_ranks)
        self.exchange = exchange or (lambda obj: [obj])   # one process holds everything: nothing to exchange
        if groups is None:
            groups = proximity_groups_from_scene(scene)
        if body_rank is None:
            body_rank, _ = shards_from_groups(groups, n_ranks)
        self.owner = np.asarray(body_rank, np.int32).copy()           # global body -> rank (-1: fixed, replicated)
        self.desc = [b.copy() for b in scene.bodies]                  # global body -> descriptor (pose / velocities refreshed when it moves)
        self.cols = [[] for _ in scene.bodies]                        # global body -> its collider descriptors
        for c, p in zip(scene.colliders, scene.collider_parents):
            if p >= 0:
                self.cols[p].append(c.copy())
        self.joints = [j.copy() for j in scene.joints]                # impulse joints by GLOBAL body id: re-inserted where their bodies go
        self.worlds, self.handle, self.gid_of_row = {}, {}, {}
        for r in self.local_ranks:
            sub, gids = partition_scene(scene, self.owner, r)
            w = make_world(sub, r)
            self.worlds[r] = w
            self.gid_of_row[r] = [int(g) for g in gids]               # row of rank r's world -> global body
            hs = w.body_handles() if hasattr(w, "body_handles") else np.arange(len(gids), dtype=np.uint64)
            self.handle[r] = {int(g): int(h) for g, h in zip(gids, hs)}
        self.migrations = 0
        self._groups0 = np.asarray(groups)
        self._horizon = self._dt * self.check_every
        for w in self.worlds.values():
            if hasattr(w, "set_shard_guard_horizon"):
                w.set_shard_guard_horizon(self._horizon)
        self._drift, self.guard_refreshes = 0.0, 0
        self._refresh_guards(first=True)

    def _local_max_speed(self) -> float:
        out = 0.0
        for r, w in self.worlds.items():
            if hasattr(w, "max_linear_speed"):
                out = max(out, float(w.max_linear_speed()))
            else:                                                     # (the oracle stand-in of the CPU tests)
                _, vel = w.read_bodies()
                rows = [row for row, g in enumerate(self.gid_of_row[r]) if g >= 0 and self.owner[g] == r]
                if rows:
                    out = max(out, float(np.linalg.norm(np.asarray(vel)[rows, :3].astype(np.float64), axis=1).max()))
        return out

    # -- state of the locally held bodies, boxes of every group -------------------------------------------------------------------------
    def _local_state(self):
        """{global body: (pos7, vel6)} of every dynamic body held here"""
        out = {}
        for r, w in self.worlds.items():
            pos, vel = w.read_bodies()
            for row, g in enumerate(self.gid_of_row[r]):
                if g >= 0 and self.owner[g] == r:
                    out[g] = (pos[row].copy(), vel[row].copy())
        return out

    def _groups_and_boxes(self):
        """current proximity groups of the locally held dynamic bodies and their boxes: [(rank, [global bodies], box min, box max)]"""
        lo0, hi0 = body_boxes(self.scene)
        half = np.where(np.isfinite(lo0), (hi0 - lo0) / 2.0, 0.0)
        recs = []
        for r, w in self.worlds.items():
            pos, vel = w.read_bodies()
            gr = w.proximity_groups() if hasattr(w, "proximity_groups") else None
            rows = [row for row, g in enumerate(self.gid_of_row[r]) if g >= 0 and self.owner[g] == r]
            if gr is None:                                                       # (oracle stand-in: every body its own group unless boxes overlap)
                gr = np.arange(len(self.gid_of_row[r]))
            by = {}
            for row in rows:
                by.setdefault(int(gr[row]) if int(gr[row]) >= 0 else -1 - row, []).append(row)
            for rows_g in by.values():
                gl = [self.gid_of_row[r][row] for row in rows_g]
                c = np.array([pos[row][:3] for row in rows_g], np.float64)
                h = np.array([half[g] for g in gl])
                allow = self.clearance + self._horizon * float(np.linalg.norm(np.array([vel[row][:3] for row in rows_g], np.float64), axis=1).max())
                recs.append((r, gl, (c - h).min(0) - allow, (c + h).max(0) + allow))
        return recs

    def _initial_boxes(self):
        """before the first step the device has no pair set yet: the groups the shards were cut from, boxed from the scene's poses"""
        lo, hi = body_boxes(self.scene)
        recs = []
        for r in self.worlds:
            mine = (self._groups0 >= 0) & (self.owner == r) & np.isfinite(lo[:, 0])
            for k in np.unique(self._groups0[mine]):
                m = mine & (self._groups0 == k)
                allow = self.clearance + self._horizon * max(float(np.linalg.norm(np.asarray(self.scene.bodies[g]["linvel"], np.float64))) for g in np.flatnonzero(m))
                recs.append((r, [int(g) for g in np.flatnonzero(m)], lo[m].min(0) - allow, hi[m].max(0) + allow))
        return recs

    def _refresh_guards(self, first=False):
        mine = self._initial_boxes() if first else self._groups_and_boxes()
        every = [x for part in self.exchange(mine) for x in part]
        self._boxes = every
        self._drift = 0.0
        self.guard_refreshes += 0 if first else 1
        for r, w in self.worlds.items():
            foreign = [(lo, hi) for (rr, _, lo, hi) in every if rr != r]
            if hasattr(w, "set_shard_guard"):
                if foreign:
                    w.set_shard_guard(np.array([f[0] for f in foreign], np.float32), np.array([f[1] for f in foreign], np.float32))
                else:
                    w.set_shard_guard(None, None)

    # -- stepping -----------------------------------------------------------------------------------------------------------------------
    def step(self, n: int = 1):
        for _ in range(n):
            hits = {}
            for r, w in self.worlds.items():
                w.step(1)
            self._since_check += 1
            if self._since_check < self.check_every:
                continue
            self._since_check = 0
            for r, w in self.worlds.items():
                caught = w.take_shard_guard_hits() if hasattr(w, "take_shard_guard_hits") else np.zeros(0, np.uint64)
                if len(caught):
                    rev = {h: g for g, h in self.handle[r].items()}
                    hits[r] = [rev[int(h)] for h in caught if int(h) in rev]
            parts = self.exchange((hits, self._local_max_speed()))   # ONE exchange per look: the hits and every rank's top speed
            anyone = [h for h, _ in parts if h]
            self._drift += max(v for _, v in parts) * self._horizon     # how far a body may be from where the boxes were taken
            if anyone:
                self._migrate({r: g for part in anyone for r, g in part.items()})
            elif self._drift > 0.5 * self.clearance:
                self._refresh_guards()

    def _migrate(self, hits):
        """hits = {rank: [global bodies its guard caught]}: plan from the exchanged boxes, so every process plans alike"""
        boxes = [x for part in self.exchange(self._groups_and_boxes()) for x in part]
        group_of = {}
        for k, (r, gl, lo, hi) in enumerate(boxes):
            for g in gl:
                group_of[g] = k
        moves = {}                                                              # group index -> destination rank
        for r, bodies in sorted(hits.items()):
            for g in bodies:
                k = group_of.get(g)
                if k is None or k in moves:
                    continue
                _, gl, lo, hi = boxes[k]
                # the foreign group it reached: the nearest box of another rank that overlaps this group's (inflated) box
                best = None
                for k2, (r2, gl2, lo2, hi2) in enumerate(boxes):
                    if r2 == r or k2 in moves:
                        continue
                    if np.all(lo <= hi2) and np.all(lo2 <= hi):
                        d = float(np.linalg.norm((lo + hi) / 2 - (lo2 + hi2) / 2))
                        if best is None or d < best[0]:
                            best = (d, k2)
                if best is None:
                    continue
                k2 = best[1]
                if len(boxes[k2][1]) < len(gl):                                  # the smaller group moves
                    moves[k2] = r
                else:
                    moves[k] = boxes[k2][0]
        if not moves:
            self._refresh_guards()
            return
        # the state of every moving body, from whoever holds it
        local = self._local_state()
        moving = {g for k in moves for g in boxes[k][1]}
        state = {}
        for part in self.exchange({g: local[g] for g in moving if g in local}):
            state.update(part)
        # ... and the LIVE descriptors of the impulse joints that travel with them (ImpulseJointSet::get = rp_impulse_joints_get: what was
        # inserted plus every rp_impulse_joints_set_motor edit since; a joint the caller removed is not listed and stays removed), from
        # the source world before its bodies — and with them their joints — are removed.  Keyed by the GLOBAL ids of the two ends; several
        # joints between one pair keep their insertion order.  A world type without the read (the CPU oracle in the gloo tests) is not
        # listed: its joints are re-created from the descriptors this set holds.
        mine = {}
        for r, w in self.worlds.items():
            if not self.joints or not hasattr(w, "impulse_joint_descs"):
                continue
            rev = {int(h): g for g, h in self.handle[r].items()}
            hs = w.joint_handles(); hs = hs[hs != np.uint64(0xFFFFFFFFFFFFFFFF)]
            table = {}
            for d in w.impulse_joint_descs(hs):
                g1, g2 = rev.get(int(d["body1"])), rev.get(int(d["body2"]))
                if g1 in moving or g2 in moving:
                    dd = d.copy(); dd["body1"], dd["body2"] = g1, g2
                    table.setdefault((g1, g2), []).append(dd)
            mine[r] = table
        live = {}
        for part in self.exchange(mine):
            live.update(part)
        # a destination's guard still holds the box the arriving group had on its old rank — the group sits INSIDE it — and an insertion
        # may run collision detection there (a world that is rebuilt around new joints does): its guard is off while the group is handed
        # over, _refresh_guards below sets the new boxes before the next step
        for dst in set(moves.values()):
            if dst in self.worlds and hasattr(self.worlds[dst], "set_shard_guard"):
                self.worlds[dst].set_shard_guard(None, None)
        for k, dst in sorted(moves.items()):
            src, gl = boxes[k][0], boxes[k][1]
            # replicated bodies (fixed, kinematic) already live on every rank: only the group's dynamic bodies change hands
            gl = [g for g in gl if int(self.desc[g]["body_type"]) == S.BODY_DYNAMIC]
            for g in gl:
                pos7, vel6 = state[g]
                d = self.desc[g]
                d["translation"], d["rotation"], d["linvel"], d["angvel"] = pos7[:3], pos7[3:], vel6[:3], vel6[3:]
                if src in self.worlds:
                    self.worlds[src].remove_body([self.handle[src].pop(g)])
                    self.gid_of_row[src] = [(-1 if x == g else x) for x in self.gid_of_row[src]]
                if dst in self.worlds:
                    w = self.worlds[dst]
                    hb = w.insert_body(d)
                    for c in self.cols[g]:
                        w.insert_collider(c, hb)
                    row = int(hb) & 0xFFFFFFFF
                    rows = self.gid_of_row[dst]
                    if row < len(rows):
                        rows[row] = g                                            # (a reused arena slot)
                    else:
                        rows.extend([-1] * (row - len(rows)) + [g])
                    self.handle[dst][g] = int(hb)
                self.owner[g] = dst
            # the group's impulse joints: removing a body dropped them at the source (RigidBodySet::remove removes attached joints);
            # the destination gets them back once both ends live there (the other end: a body of the group or a replicated one).
            # The descriptors are the source world's live ones (above) and replace the ones this set held, on every process alike; the
            # joints' warm-start impulses do not travel (nor do the contacts': §7) — the first step at the destination solves them cold.
            if self.joints:
                moved = set(gl)
                table = live.get(src)                                        # None: the source world cannot hand out descriptors
                taken, kept = {}, []
                for j in self.joints:
                    b1, b2 = int(j["body1"]), int(j["body2"])
                    if (b1 in moved or b2 in moved) and table is not None:
                        i = taken.get((b1, b2), 0); taken[(b1, b2)] = i + 1
                        if i >= len(table.get((b1, b2), [])):
                            continue                                         # removed at run time (ImpulseJointSet::remove): it stays removed
                        j = table[(b1, b2)][i]
                    kept.append(j)
                self.joints = kept
            if dst in self.worlds and self.joints:
                w = self.worlds[dst]
                row_of = {g: int(h) for g, h in self.handle[dst].items()}   # rp_joint_desc.body1 / body2 are RigidBodyHandles
                back = []
                for j in self.joints:
                    b1, b2 = int(j["body1"]), int(j["body2"])
                    if b1 in moved or b2 in moved:
                        if b1 not in row_of or b2 not in row_of:
                            # (a joint links the proximity groups of its bodies, so both ends travel together or are replicated: an end that
                            # is missing here means the shards and the scene disagree — do not continue with a world that lost a joint)
                            raise RuntimeError(f"migration: joint ({b1}, {b2}) of a moved body cannot be re-created on rank {dst}: body {b1 if b1 not in row_of else b2} is not there")
                        jj = j.copy(); jj["body1"], jj["body2"] = row_of[b1], row_of[b2]
                        back.append(jj)
                if back:
                    if not hasattr(w, "insert_impulse_joints"):
                        raise NotImplementedError("this world type cannot re-insert the joints of a migrating group")
                    w.insert_impulse_joints(np.array(back, dtype=S.JOINT_DTYPE))
            self.migrations += 1
        self._refresh_guards()

    # -- readback -----------------------------------------------------------------------------------------------------------------------
    def read_bodies(self):
        """(pos[n_global, 7], vel[n_global, 6]) assembled from every rank (fixed bodies from the scene)"""
        n = len(self.scene.bodies)
        pos = np.zeros((n, 7), np.float32); vel = np.zeros((n, 6), np.float32)
        for g, b in enumerate(self.scene.bodies):
            pos[g, :3], pos[g, 3:] = b["translation"], b["rotation"]
        state = {}
        for part in self.exchange(self._local_state()):
            state.update(part)
        for g, (p, v) in state.items():
            pos[g], vel[g] = p, v
        return pos, vel
