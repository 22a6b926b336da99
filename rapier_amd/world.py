"""Host-side mirror of the reference's public API for the step path, over the C ABI.

Names follow the reference: `PhysicsWorld` (physics_world.rs:61-157) owns
`RigidBodySet` / `ColliderSet` / `ImpulseJointSet` / `PhysicsPipeline` /
`IntegrationParameters`; `insert`, `insert_body`, `insert_collider`, `insert_impulse_joint`,
`step` keep the reference's argument meaning.  All state lives on the MI355X; reading a body
downloads it.  Errors from the device library raise `RapierHipError` (the reference panics).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _ffi, scenes as S


class RapierHipError(RuntimeError):
    pass


def _check(world_ptr, status: int, what: str):
    if status != 0:
        msg = _ffi.lib().rp_last_error(world_ptr)
        raise RapierHipError(f"{what} failed (status {status}): {msg.decode() if msg else ''}")


class IntegrationParameters:
    """IntegrationParameters (integration_parameters.rs:181-304): attribute access on the packed struct."""

    def __init__(self, values: np.ndarray | None = None):
        object.__setattr__(self, "_v", S.default_params() if values is None else np.array(values, dtype=S.PARAMS_DTYPE))

    def __getattr__(self, name):
        v = object.__getattribute__(self, "_v")
        if name in v.dtype.names:
            return v[name].item()
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in self._v.dtype.names:
            self._v[name] = value
        else:
            raise AttributeError(name)

    def inv_dt(self) -> float:
        return 0.0 if self.dt == 0.0 else 1.0 / self.dt

    def prediction_distance(self) -> float:
        return self.normalized_prediction_distance * self.length_unit

    def as_array(self) -> np.ndarray:
        return np.ascontiguousarray(self._v)


class RigidBodyHandle(int):
    """Index{index, generation} packed as generation<<32 | index (arena.rs:58-90)."""

    def into_raw_parts(self):
        return int(self) & 0xFFFFFFFF, int(self) >> 32


class ColliderHandle(RigidBodyHandle):
    pass


class RigidBodySet:
    def __init__(self, world: "PhysicsWorld"):
        self._w = world
        self._n = 0

    def insert(self, body: np.ndarray) -> RigidBodyHandle:
        return self._w.insert_body(body)

    def remove(self, handle):
        self._w.remove_body(handle)

    def __len__(self):
        return self._n

    def translation(self, handle) -> np.ndarray:
        pos, _ = self._w.read_bodies([handle])
        return pos[0, :3]


class ColliderSet:
    def __init__(self, world: "PhysicsWorld"):
        self._w = world
        self._n = 0

    def insert_with_parent(self, collider: np.ndarray, parent) -> ColliderHandle:
        return self._w.insert_collider(collider, parent)

    def insert(self, collider: np.ndarray) -> ColliderHandle:
        return self._w.insert_collider(collider, None)

    def remove(self, handle):
        self._w.remove_collider(handle)

    def __len__(self):
        return self._n


class ImpulseJointSet:
    def __init__(self, world: "PhysicsWorld"):
        self._w = world
        self._n = 0

    def insert(self, body1, body2, joint: np.ndarray):
        return self._w.insert_impulse_joint(body1, body2, joint)

    def remove(self, handle):
        self._w.remove_impulse_joint(handle)

    def __len__(self):
        return self._n


class PhysicsPipeline:
    """PhysicsPipeline (physics_pipeline/mod.rs:45): `step` + `counters`."""

    def __init__(self, world: "PhysicsWorld"):
        self._w = world

    def step(self, nsteps: int = 1):
        self._w.step(nsteps)

    @property
    def counters(self) -> dict:
        return self._w.counters()


class PhysicsWorld:
    """PhysicsWorld::new() on device `device` (physics_world.rs:61-157)."""

    def __init__(self, gravity=(0.0, -9.81, 0.0), integration_parameters: IntegrationParameters | None = None, device: int = 0,
                 index_addressing: bool | None = None):
        # index_addressing: values below 2^32 passed where a handle is expected name an arena SLOT (its current occupant,
        # RigidBodySet::get_unknown_gen) instead of a generation-0 handle.  Off by default — a stale handle must be refused, not
        # silently retargeted; the oracle-lockstep tests, which keep the oracle's plain indices, switch it on (RP_INDEX_ADDRESSING=1).
        self.index_addressing = (os.environ.get("RP_INDEX_ADDRESSING") == "1") if index_addressing is None else bool(index_addressing)
        self._lib = _ffi.lib()
        self.integration_parameters = integration_parameters or IntegrationParameters()
        self.gravity = tuple(float(g) for g in gravity)
        self._ptr = C.c_void_p()
        prm = self.integration_parameters.as_array()
        grav = np.asarray(self.gravity, dtype=np.float32)
        st = self._lib.rp_world_create(prm.ctypes.data, grav.ctypes.data, device, C.byref(self._ptr))
        if st != 0 or not self._ptr:
            raise RapierHipError(f"rp_world_create failed (status {st}): no usable HIP device; there is no CPU fallback")
        self.bodies = RigidBodySet(self)
        self.colliders = ColliderSet(self)
        self.impulse_joints = ImpulseJointSet(self)
        self.physics_pipeline = PhysicsPipeline(self)

    @classmethod
    def from_scene(cls, scene: S.Scene, device: int = 0, index_addressing: bool | None = None) -> "PhysicsWorld":
        w = cls(gravity=scene.gravity, integration_parameters=IntegrationParameters(scene.params), device=device, index_addressing=index_addressing)
        for pts, tris in getattr(scene, "polyhedra", []):
            w.add_convex_polyhedron(pts, tris)
        for comp in getattr(scene, "composites", []):
            if comp[0] == "compound":
                w.add_compound(comp[1])
            elif comp[0] == "trimesh":
                w.add_trimesh(comp[1], comp[2])
            else:
                w.add_heightfield(comp[1], comp[2])
        bodies, cols, joints = scene.body_array(), scene.collider_array(), scene.joint_array()
        parents = scene.parent_array().astype(np.int64) if len(cols) else np.zeros(0, np.int64)
        ph = np.where(parents < 0, np.uint64(_ffi.RP_INVALID_HANDLE), parents.astype(np.uint64)).astype(np.uint64)
        # a batch (scenes.batch): every sub-world's bodies and colliders go in behind its own rp_world_begin_subworld
        starts = list(getattr(scene, "subworlds", None) or [(0, 0, 0)]) + [(len(bodies), len(cols), len(joints))]
        for k in range(len(starts) - 1):
            (b0, c0, _), (b1, c1, _) = starts[k], starts[k + 1]
            if k:
                w.begin_subworld()
            if b1 > b0:
                w.insert_bodies(bodies[b0:b1])
            if c1 > c0:
                w.insert_colliders(cols[c0:c1], ph[c0:c1])
        w.subworlds = [(starts[k][0], starts[k + 1][0]) for k in range(len(starts) - 1)]   # body rows [first, end) of every sub-world
        if len(joints):
            w.insert_impulse_joints(joints)
        return w

    def set_integration_parameters(self, params):
        """Replace the world's IntegrationParameters (rp_params_set); switching `friction_model` on a stepped
        world rebuilds the device world from the current body states."""
        ip = params if isinstance(params, IntegrationParameters) else IntegrationParameters(params)
        _check(self._ptr, self._lib.rp_params_set(self._ptr, ip.as_array().ctypes.data), "rp_params_set")
        self.integration_parameters = ip

    # ---- convex polyhedra ----
    def add_convex_polyhedron(self, points, triangles=None) -> int:
        """SharedShape::convex_mesh(points, indices) — SharedShape::convex_hull(points) without `triangles` — registered with the world
        (rp_convex_polyhedron_create); colliders use the id: collider_desc(shape=SHAPE_CONVEX, half_extents=(id, 0, 0))."""
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        tris = None if triangles is None else np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
        out = np.zeros(1, np.int32)
        _check(self._ptr, self._lib.rp_convex_polyhedron_create(self._ptr, len(pts), pts.ctypes.data, 0 if tris is None else len(tris),
                                                                None if tris is None else tris.ctypes.data, out.ctypes.data), "rp_convex_polyhedron_create")
        return int(out[0])

    def add_compound(self, parts) -> int:
        """SharedShape::compound(parts) (rp_compound_create): `parts` = collider descriptors; colliders use the id:
        collider_desc(shape=SHAPE_COMPOUND, half_extents=(id, 0, 0))."""
        parts = np.ascontiguousarray(np.array(list(parts), dtype=S.COLLIDER_DTYPE))
        out = np.zeros(1, np.int32)
        _check(self._ptr, self._lib.rp_compound_create(self._ptr, len(parts), parts.ctypes.data, out.ctypes.data), "rp_compound_create")
        return int(out[0])

    def add_trimesh(self, vertices, triangles) -> int:
        """SharedShape::trimesh(vertices, indices) (rp_trimesh_create): collider_desc(shape=SHAPE_TRIMESH, half_extents=(id, 0, 0))."""
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3); t = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
        out = np.zeros(1, np.int32)
        _check(self._ptr, self._lib.rp_trimesh_create(self._ptr, len(v), v.ctypes.data, len(t), t.ctypes.data, out.ctypes.data), "rp_trimesh_create")
        return int(out[0])

    def add_heightfield(self, heights, scale) -> int:
        """SharedShape::heightfield(heights, scale) (rp_heightfield_create): used like a triangle mesh (shape=SHAPE_TRIMESH)."""
        h = np.ascontiguousarray(heights, np.float32); sc = np.ascontiguousarray(np.asarray(scale, np.float32).reshape(3))
        out = np.zeros(1, np.int32)
        _check(self._ptr, self._lib.rp_heightfield_create(self._ptr, h.shape[0], h.shape[1], h.ctypes.data, sc.ctypes.data, out.ctypes.data), "rp_heightfield_create")
        return int(out[0])

    def read_convex_polyhedron(self, pid: int) -> dict:
        """the polyhedron as the library holds it (rp_convex_polyhedron_read): recentred points, faces as vertex loops, mass properties"""
        cnt = np.zeros(4, np.int32)
        _check(self._ptr, self._lib.rp_convex_polyhedron_read(self._ptr, pid, cnt.ctypes.data, None, None, None, None, None, None, None), "rp_convex_polyhedron_read")
        nv, nf, nl, ne = (int(x) for x in cnt)
        pts, fn = np.zeros((nv, 3), np.float32), np.zeros((nf, 3), np.float32)
        ff, fc, lv, le, props = np.zeros(nf, np.int32), np.zeros(nf, np.int32), np.zeros(nl, np.int32), np.zeros(nl, np.int32), np.zeros(20, np.float32)
        _check(self._ptr, self._lib.rp_convex_polyhedron_read(self._ptr, pid, cnt.ctypes.data, pts.ctypes.data, fn.ctypes.data, ff.ctypes.data, fc.ctypes.data,
                                                              lv.ctypes.data, le.ctypes.data, props.ctypes.data), "rp_convex_polyhedron_read")
        return dict(points=pts, face_normals=fn, face_first=ff, face_count=fc, loop_vertex=lv, loop_edge=le, n_edges=ne, props=props)

    # ---- handles ----
    def body_handles(self) -> np.ndarray:
        """RigidBodySet::iter as handles: generation << 32 | index of every arena row (rp_bodies_handles)."""
        n = self._lib.rp_bodies_handles(self._ptr, 0, None)
        out = np.zeros(max(n, 0), np.uint64)
        if n > 0:
            self._lib.rp_bodies_handles(self._ptr, n, out.ctypes.data)
        return out

    def collider_handles(self) -> np.ndarray:
        n = self._lib.rp_colliders_handles(self._ptr, 0, None)
        out = np.zeros(max(n, 0), np.uint64)
        if n > 0:
            self._lib.rp_colliders_handles(self._ptr, n, out.ctypes.data)
        return out

    def _bh(self, handles, table=None) -> np.ndarray:
        """Body (or collider) handles as the ABI wants them: generation << 32 | index, passed through UNCHANGED.  A handle issued before
        the first removal has generation 0 and therefore equals its arena index; once the slot has been reused its occupant carries a
        higher generation and the old value is refused by the library (RP_ERR_INVALID), exactly as Arena::get returns None for a stale
        handle (data/arena.rs).  Addressing a slot's CURRENT occupant by index is a separate, explicit operation: `body_handles_at` /
        `collider_handles_at` (RigidBodySet::get_unknown_gen) — or, for callers that keep plain indices throughout, a world created
        with index_addressing=True, where every value below 2^32 is such an index."""
        h = np.ascontiguousarray(np.atleast_1d(np.asarray(handles, dtype=np.uint64))).copy()
        if self.index_addressing:
            idx = h < np.uint64(1 << 32)
            if idx.any():
                cur = self.body_handles() if table is None else table
                ok = idx & (h < np.uint64(len(cur)))
                h[ok] = cur[h[ok].astype(np.int64)]
        return h

    def _ch(self, handles) -> np.ndarray:
        return self._bh(handles, table=self.collider_handles() if self.index_addressing else None)

    def body_handles_at(self, indices) -> np.ndarray:
        """RigidBodySet::get_unknown_gen: the handles of the bodies that occupy these arena slots NOW."""
        return self._handles_at(indices, self.body_handles(), "body")

    def collider_handles_at(self, indices) -> np.ndarray:
        return self._handles_at(indices, self.collider_handles(), "collider")

    @staticmethod
    def _handles_at(indices, table, what) -> np.ndarray:
        i = np.atleast_1d(np.asarray(indices, dtype=np.int64))
        if len(i) and (i.min() < 0 or i.max() >= len(table)):
            raise IndexError(f"{what} index out of range")
        return table[i].copy()  # (a free row yields the handle of its last occupant, which every entry point but rp_bodies_read refuses)

    # ---- insertion (RigidBodySet::insert, ColliderSet::insert_with_parent, ...) ----
    def insert_bodies(self, descs: np.ndarray) -> np.ndarray:
        descs = np.ascontiguousarray(descs, dtype=S.BODY_DTYPE)
        out = np.zeros(len(descs), np.uint64)
        _check(self._ptr, self._lib.rp_bodies_insert(self._ptr, len(descs), descs.ctypes.data, out.ctypes.data), "rp_bodies_insert")
        self.bodies._n += len(descs)
        return out

    def insert_body(self, body: np.ndarray) -> RigidBodyHandle:
        return RigidBodyHandle(int(self.insert_bodies(np.array([body], dtype=S.BODY_DTYPE))[0]))

    def insert_colliders(self, descs: np.ndarray, parents: np.ndarray) -> np.ndarray:
        descs = np.ascontiguousarray(descs, dtype=S.COLLIDER_DTYPE)
        parents = self._bh(np.ascontiguousarray(parents, dtype=np.uint64)) if len(descs) else np.zeros(0, np.uint64)
        out = np.zeros(len(descs), np.uint64)
        _check(self._ptr, self._lib.rp_colliders_insert(self._ptr, len(descs), descs.ctypes.data, parents.ctypes.data, out.ctypes.data), "rp_colliders_insert")
        self.colliders._n += len(descs)
        return out

    def insert_collider(self, collider: np.ndarray, parent=None) -> ColliderHandle:
        p = np.array([_ffi.RP_INVALID_HANDLE if parent is None else int(self._bh([int(parent)])[0])], dtype=np.uint64)
        return ColliderHandle(int(self.insert_colliders(np.array([collider], dtype=S.COLLIDER_DTYPE), p)[0]))

    def insert(self, body: np.ndarray, collider: np.ndarray):
        """PhysicsWorld::insert(body, collider) -> (body handle, collider handle)."""
        b = self.insert_body(body)
        return b, self.insert_collider(collider, b)

    def joint_handles(self) -> np.ndarray:
        """ImpulseJointSet::iter as handles (rp_impulse_joints_handles): one entry per joint ever inserted, insertion order; a removed
        joint's entry is RP_INVALID_HANDLE."""
        n = self._lib.rp_impulse_joints_handles(self._ptr, 0, None)
        out = np.zeros(max(n, 0), np.uint64)
        if n > 0:
            self._lib.rp_impulse_joints_handles(self._ptr, n, out.ctypes.data)
        return out

    def _jh(self, handles) -> np.ndarray:
        """Joint handles as the ABI wants them (generation << 32 | arena slot), unchanged; under index_addressing a value below 2^32 is
        the joint's insertion ordinal (what the oracle numbers joints by) and is replaced by that joint's current handle."""
        h = np.ascontiguousarray(np.atleast_1d(np.asarray(handles, dtype=np.uint64))).copy()
        if self.index_addressing:
            idx = h < np.uint64(1 << 32)
            if idx.any():
                cur = self.joint_handles()
                ok = idx & (h < np.uint64(len(cur)))
                h[ok] = cur[h[ok].astype(np.int64)]
        return h

    def insert_impulse_joints(self, descs: np.ndarray) -> np.ndarray:
        descs = np.ascontiguousarray(descs, dtype=S.JOINT_DTYPE).copy()
        if len(descs):  # rp_joint_desc.body1 / body2 are RigidBodyHandles
            descs["body1"], descs["body2"] = self._bh(descs["body1"]), self._bh(descs["body2"])
        out = np.zeros(len(descs), np.uint64)
        _check(self._ptr, self._lib.rp_impulse_joints_insert(self._ptr, len(descs), descs.ctypes.data, out.ctypes.data), "rp_impulse_joints_insert")
        self.impulse_joints._n += len(descs)
        return out

    def impulse_joint_descs(self, handles) -> np.ndarray:
        """ImpulseJointSet::get(handle) (rp_impulse_joints_get): the live descriptors — as inserted plus every motor edit since —,
        body1 / body2 as RigidBodyHandles."""
        h = self._jh(handles)
        out = np.zeros(len(h), dtype=S.JOINT_DTYPE)
        if len(h):
            _check(self._ptr, self._lib.rp_impulse_joints_get(self._ptr, len(h), h.ctypes.data, out.ctypes.data), "rp_impulse_joints_get")
        return out

    def insert_impulse_joint(self, body1, body2, joint: np.ndarray):
        j = np.array([joint], dtype=S.JOINT_DTYPE)
        j["body1"], j["body2"] = int(body1), int(body2)   # ImpulseJointSet::insert(body1: RigidBodyHandle, body2: RigidBodyHandle, ..)
        return int(self.insert_impulse_joints(j)[0])

    # ---- removal (RigidBodySet::remove, ColliderSet::remove, ImpulseJointSet::remove) ----
    def _remove(self, fn, handles, what):
        h = np.ascontiguousarray(np.atleast_1d(np.asarray(handles, dtype=np.uint64)))
        _check(self._ptr, fn(self._ptr, len(h), h.ctypes.data), what)

    def remove_body(self, handles):
        """RigidBodySet::remove(handle, ..., remove_attached_colliders = true)."""
        self._remove(self._lib.rp_bodies_remove, self._bh(handles), "rp_bodies_remove")

    def remove_collider(self, handles):
        self._remove(self._lib.rp_colliders_remove, self._ch(handles), "rp_colliders_remove")

    def remove_impulse_joint(self, handles):
        self._remove(self._lib.rp_impulse_joints_remove, self._jh(handles), "rp_impulse_joints_remove")

    def quarantined(self) -> np.ndarray:
        n = self._lib.rp_quarantine_read(self._ptr, 0, None)
        if n < 0:
            _check(self._ptr, n, "rp_quarantine_read")
        out = np.zeros(n, np.uint64)
        if n:
            self._lib.rp_quarantine_read(self._ptr, n, out.ctypes.data)
        return out

    # ---- stepping ----
    def step(self, nsteps: int = 1):
        """PhysicsWorld::step() x nsteps (asynchronous; see `sync`)."""
        _check(self._ptr, self._lib.rp_step(self._ptr, int(nsteps)), "rp_step")

    def sync(self):
        _check(self._ptr, self._lib.rp_sync(self._ptr), "rp_sync")

    # ---- readback ----
    def read_bodies(self, handles=None):
        if handles is None:
            n = self._lib.rp_num_bodies(self._ptr)
            hp = None
        else:
            h = self._bh(handles)
            n, hp = len(h), h.ctypes.data
        pos = np.zeros((n, 7), np.float32)
        vel = np.zeros((n, 6), np.float32)
        _check(self._ptr, self._lib.rp_bodies_read(self._ptr, n, hp, pos.ctypes.data, vel.ctypes.data), "rp_bodies_read")
        return pos, vel

    def write_bodies(self, handles, pos7=None, vel6=None):
        h = self._bh(handles)
        p = None if pos7 is None else np.ascontiguousarray(pos7, dtype=np.float32)
        v = None if vel6 is None else np.ascontiguousarray(vel6, dtype=np.float32)
        _check(self._ptr, self._lib.rp_bodies_write(self._ptr, len(h), h.ctypes.data, None if p is None else p.ctypes.data,
                                                  None if v is None else v.ctypes.data), "rp_bodies_write")

    def add_force(self, handles, force=None, torque=None, reset: bool = False):
        """RigidBody::{reset_forces, reset_torques} (when ``reset``) then add_force / add_torque(.., wake_up=true)."""
        h = self._bh(handles)
        f = None if force is None else np.ascontiguousarray(np.asarray(force, np.float32).reshape(len(h), 3))
        t = None if torque is None else np.ascontiguousarray(np.asarray(torque, np.float32).reshape(len(h), 3))
        _check(self._ptr, self._lib.rp_bodies_add_force(self._ptr, len(h), h.ctypes.data, None if f is None else f.ctypes.data,
                                                      None if t is None else t.ctypes.data, 1 if reset else 0), "rp_bodies_add_force")

    def apply_impulse(self, handles, impulse=None, torque_impulse=None):
        """RigidBody::{apply_impulse, apply_torque_impulse}(.., wake_up=true)."""
        h = self._bh(handles)
        f = None if impulse is None else np.ascontiguousarray(np.asarray(impulse, np.float32).reshape(len(h), 3))
        t = None if torque_impulse is None else np.ascontiguousarray(np.asarray(torque_impulse, np.float32).reshape(len(h), 3))
        _check(self._ptr, self._lib.rp_bodies_apply_impulse(self._ptr, len(h), h.ctypes.data, None if f is None else f.ctypes.data,
                                                          None if t is None else t.ctypes.data), "rp_bodies_apply_impulse")

    def set_next_kinematic_position(self, handles, pos7):
        """RigidBody::set_next_kinematic_position (rigid_body.rs:1085-1093) for kinematic bodies."""
        h = self._bh(handles)
        p = np.ascontiguousarray(np.asarray(pos7, dtype=np.float32).reshape(len(h), 7))
        _check(self._ptr, self._lib.rp_bodies_set_next_kinematic_position(self._ptr, len(h), h.ctypes.data, p.ctypes.data), "rp_bodies_set_next_kinematic_position")

    def collision_events(self, cap: int | None = None) -> np.ndarray:
        """Drain the CollisionEvent queue: rows (collider1, collider2, started, flags, step).  ``cap`` bounds how many events this
        call takes; the rest stays queued."""
        n = self._lib.rp_collision_events_read(self._ptr, 0, None)
        if n < 0:
            _check(self._ptr, n, "rp_collision_events_read")
        if cap is not None:
            n = min(n, int(cap))
        out = np.zeros((max(n, 1), 5), np.int32)
        m = self._lib.rp_collision_events_read(self._ptr, n, out.ctypes.data)
        if m < 0:
            _check(self._ptr, m, "rp_collision_events_read")
        return out[:m]

    def intersection_pairs(self) -> np.ndarray:
        """NarrowPhase::intersection_pairs: rows (collider1, collider2, intersecting) of every pair that involves a sensor."""
        n = self._lib.rp_intersection_pairs_read(self._ptr, 0, None)
        if n < 0:
            _check(self._ptr, n, "rp_intersection_pairs_read")
        out = np.zeros((max(n, 1), 3), np.int32)
        m = self._lib.rp_intersection_pairs_read(self._ptr, n, out.ctypes.data)
        if m < 0:
            _check(self._ptr, m, "rp_intersection_pairs_read")
        return out[:min(n, m)]

    def intersection_pair(self, c1, c2):
        """NarrowPhase::intersection_pair(c1, c2): True / False, or None when the two colliders form no sensor pair."""
        lo, hi = (int(c1), int(c2)) if int(c1) < int(c2) else (int(c2), int(c1))
        for a, b, i in self.intersection_pairs():
            if (min(a, b), max(a, b)) == (lo, hi):
                return bool(i)
        return None

    def contact_force_events(self):
        """Drain the ContactForceEvent queue: (meta rows (collider1, collider2, step, started), 8 floats per event:
        total_force xyz, total_force_magnitude, max_force_direction xyz, max_force_magnitude)."""
        n = self._lib.rp_contact_force_events_read(self._ptr, 0, None)
        if n < 0:
            _check(self._ptr, n, "rp_contact_force_events_read")
        raw = np.zeros((max(n, 1), 12), np.float32)
        m = self._lib.rp_contact_force_events_read(self._ptr, n, raw.ctypes.data)
        if m < 0:
            _check(self._ptr, m, "rp_contact_force_events_read")
        raw = raw[:m]
        return raw[:, :4].copy().view(np.int32), raw[:, 4:].copy()

    def set_additional_solver_iterations(self, handles, counts):
        """RigidBody::set_additional_solver_iterations: extra TGS substeps for the whole connected component of each body."""
        h = self._bh(handles)
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(counts, dtype=np.int32), h.shape))
        _check(self._ptr, self._lib.rp_bodies_set_additional_solver_iterations(self._ptr, len(h), h.ctypes.data, c.ctypes.data), "rp_bodies_set_additional_solver_iterations")

    def wake_up(self, handles, strong: bool = True):
        """IslandManager::wake_up (island_manager/sleep.rs:31) — effective at the next step, island-wide."""
        h = self._bh(handles)
        _check(self._ptr, self._lib.rp_bodies_wake_up(self._ptr, len(h), h.ctypes.data, 1 if strong else 0), "rp_bodies_wake_up")

    def sleeping(self, handles=None) -> np.ndarray:
        """RigidBody::is_sleeping for the given handles (default: every body, arena order)."""
        if handles is None:
            handles = self.body_handles()
        h = self._bh(handles)
        out = np.zeros(len(h), np.int32)
        _check(self._ptr, self._lib.rp_bodies_is_sleeping(self._ptr, len(h), h.ctypes.data, out.ctypes.data), "rp_bodies_is_sleeping")
        return out.astype(bool)

    def island_labels(self, handles=None) -> np.ndarray:
        """IslandManager::persistent_island_of per body (-1: fixed / removed, or a world without sleepable bodies)."""
        if handles is None:
            handles = self.body_handles()
        h = self._bh(handles)
        out = np.zeros(len(h), np.int32)
        _check(self._ptr, self._lib.rp_bodies_persistent_island(self._ptr, len(h), h.ctypes.data, out.ctypes.data), "rp_bodies_persistent_island")
        return out

    def proximity_groups(self, handles=None) -> np.ndarray:
        """Connected components of the non-fixed bodies over the live broad-phase pairs and the joints (rp_bodies_proximity_group): the
        unit of island sharding over GPUs.  -1: fixed / removed bodies.  Needs one step."""
        if handles is None:
            handles = self.body_handles()
        h = self._bh(handles)
        out = np.zeros(len(h), np.int32)
        _check(self._ptr, self._lib.rp_bodies_proximity_group(self._ptr, len(h), h.ctypes.data, out.ctypes.data), "rp_bodies_proximity_group")
        return out

    def set_shard_guard(self, box_min=None, box_max=None):
        """Boxes that hold the bodies of OTHER shards (rp_world_set_shard_guard); None removes the guard."""
        if box_min is None or len(box_min) == 0:
            _check(self._ptr, self._lib.rp_world_set_shard_guard(self._ptr, 0, None, None), "rp_world_set_shard_guard")
            return
        a = np.ascontiguousarray(np.asarray(box_min, np.float32).reshape(-1, 3)); b = np.ascontiguousarray(np.asarray(box_max, np.float32).reshape(-1, 3))
        _check(self._ptr, self._lib.rp_world_set_shard_guard(self._ptr, len(a), a.ctypes.data, b.ctypes.data), "rp_world_set_shard_guard")

    ISLAND_STATS = ("merged", "multiway_groups", "removals", "connected", "detached", "hot", "over_budget", "sleeping_deferred", "global_splits",
                    "global_split_pieces", "bids", "bid_ties", "sleep_blocked", "order_dependent", "detach_size_ties", "split_keep_ties")

    def begin_subworld(self) -> int:
        """everything inserted from now on belongs to a new sub-world (rp_world_begin_subworld): a batch of small worlds in one device
        world, stepped by the same launches; colliders of different sub-worlds never pair"""
        r = self._lib.rp_world_begin_subworld(self._ptr)
        if r < 0:
            _check(self._ptr, r, "rp_world_begin_subworld")
        return int(r)

    def set_shard_guard_horizon(self, seconds: float):
        """how long a guard hit may wait for take_shard_guard_hits (rp_world_set_shard_guard_horizon)"""
        _check(self._ptr, self._lib.rp_world_set_shard_guard_horizon(self._ptr, float(seconds)), "rp_world_set_shard_guard_horizon")

    def max_linear_speed(self) -> float:
        """max |linvel| over the non-fixed bodies, reduced on the device (rp_world_max_linear_speed)"""
        out = C.c_float(0.0)
        _check(self._ptr, self._lib.rp_world_max_linear_speed(self._ptr, C.byref(out)), "rp_world_max_linear_speed")
        return float(out.value)

    def take_shard_guard_hits(self) -> np.ndarray:
        """Handles of the bodies the shard guard caught since the last call; clears the guard so the world can go on
        (rp_world_shard_guard_take_hits)."""
        n = self._lib.rp_world_shard_guard_take_hits(self._ptr, 0, None)
        if n < 0:
            _check(self._ptr, n, "rp_world_shard_guard_take_hits")
        out = np.zeros(max(n, 1), np.uint64)
        m = self._lib.rp_world_shard_guard_take_hits(self._ptr, len(out), out.ctypes.data)
        if m < 0:
            _check(self._ptr, m, "rp_world_shard_guard_take_hits")
        return out[:max(m, 0)]

    def island_stats(self) -> dict:
        """debug aid: counters of the persistent-island machinery (the slots the device maintains; same names as the oracle's)"""
        out = np.zeros(16, np.int32)
        _check(self._ptr, self._lib.rp_debug_islands(self._ptr, out.ctypes.data, None, -1, None), "rp_debug_islands")
        return dict(zip(self.ISLAND_STATS, (int(v) for v in out)))

    def island_state(self, island: int) -> dict:
        row = np.zeros(5, np.int32)
        _check(self._ptr, self._lib.rp_debug_islands(self._ptr, None, None, int(island), row.ctypes.data), "rp_debug_islands")
        return dict(zip(("used", "nbodies", "dirty", "denied", "sleeping"), (int(v) for v in row)))

    def island_globals(self):
        sp = np.zeros(2, np.int32)
        _check(self._ptr, self._lib.rp_debug_islands(self._ptr, None, sp.ctypes.data, -1, None), "rp_debug_islands")
        return int(sp[0]), int(sp[1])

    def contacts(self):
        m = self._lib.rp_contacts_read(self._ptr, 0, None, None, None)
        if m < 0:
            _check(self._ptr, m, "rp_contacts_read")
        meta = np.zeros((m, 4), np.int32)
        nrm = np.zeros((m, 3), np.float32)
        imp = np.zeros((m, 4), np.float32)
        if m:
            self._lib.rp_contacts_read(self._ptr, m, meta.ctypes.data, nrm.ctypes.data, imp.ctypes.data)
        return meta, nrm, imp

    def read_joints(self):
        """(colour, impulse xyz) of every impulse joint, insertion order."""
        n = self.impulse_joints._n
        col = np.zeros(n, np.int32)
        imp = np.zeros((n, 3), np.float32)
        if n:
            _check(self._ptr, self._lib.rp_impulse_joints_read(self._ptr, n, None, col.ctypes.data, imp.ctypes.data), "rp_impulse_joints_read")
        return col, imp

    def set_joint_motor(self, handles, axes, **motor):
        """GenericJoint::set_motor* through ImpulseJointSet::get_mut(handle, true): joint ``handles[i]`` gets the motor described by
        the keywords (scenes.motor_desc) on axis ``axes[i]`` (0..5 = LinX..AngZ); the motor is enabled and both bodies are woken."""
        h = self._jh(handles)
        a = np.ascontiguousarray(np.broadcast_to(np.atleast_1d(np.asarray(axes, dtype=np.int32)), h.shape))
        m = np.ascontiguousarray(np.broadcast_to(S.motor_desc(**motor), h.shape))
        _check(self._ptr, self._lib.rp_impulse_joints_set_motor(self._ptr, len(h), h.ctypes.data, a.ctypes.data, m.ctypes.data), "rp_impulse_joints_set_motor")

    def joint_motor_impulses(self) -> np.ndarray:
        """JointMotor::impulse of the six axes of every impulse joint, insertion order."""
        n = self.impulse_joints._n
        out = np.zeros((n, 6), np.float32)
        if n:
            _check(self._ptr, self._lib.rp_impulse_joints_read_motor_impulses(self._ptr, n, None, out.ctypes.data), "rp_impulse_joints_read_motor_impulses")
        return out

    def total_contact_impulse(self) -> float:
        _, _, imp = self.contacts()
        return float(imp.sum())

    def enable_timers(self, enable: bool = True):
        _check(self._ptr, self._lib.rp_counters_enable(self._ptr, 1 if enable else 0), "rp_counters_enable")

    def counters(self) -> dict:
        c = _ffi.Counters()
        _check(self._ptr, self._lib.rp_counters_read(self._ptr, C.byref(c)), "rp_counters_read")
        return {n: getattr(c, n) for n, _ in _ffi.Counters._fields_}

    def solver_loop_time_ms(self):
        avg = C.c_float()
        n = C.c_int32()
        self._lib.rp_solver_loop_time_ms(self._ptr, C.byref(avg), C.byref(n))
        return float(avg.value), int(n.value)

    # ---- the collective of a sharded world (include/rapier_hip.h: rp_world_set_global_ids / rp_world_pack_bodies / rp_shard_all_gather) ----
    def set_global_ids(self, ids):
        """ids[i] = the id body row i carries in a packed / gathered row (its index in the unsharded world)"""
        a = np.ascontiguousarray(ids, dtype=np.int64)
        _check(self._ptr, self._lib.rp_world_set_global_ids(self._ptr, len(a), a.ctypes.data), "rp_world_set_global_ids")

    def pack_bodies(self):
        """(device pointer, rows): this shard's live non-fixed bodies packed on the device, one 64-byte row each"""
        import ctypes as C
        ptr, n = C.c_void_p(), C.c_int32()
        _check(self._ptr, self._lib.rp_world_pack_bodies(self._ptr, C.byref(ptr), C.byref(n)), "rp_world_pack_bodies")
        return ptr.value, int(n.value)

    def shard_all_gather(self, comm: "ShardComm", rows_per_rank: int, pos: np.ndarray, vel: np.ndarray):
        """pack -> ncclAllGather on the world's stream -> scatter by global id into pos[n_global, 7] / vel[n_global, 6] (in place; rows no
        rank owns keep their values).  Returns the number of bodies every rank contributed."""
        assert pos.dtype == np.float32 and vel.dtype == np.float32 and pos.flags.c_contiguous and vel.flags.c_contiguous and len(pos) == len(vel)
        per = np.zeros(comm.world_size, np.int32)
        _check(self._ptr, self._lib.rp_shard_all_gather(self._ptr, comm._ptr, int(rows_per_rank), len(pos), pos.ctypes.data, vel.ctypes.data, per.ctypes.data), "rp_shard_all_gather")
        return per

    def close(self):
        if getattr(self, "_ptr", None):
            self._lib.rp_world_destroy(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardComm:
    """rp_comm: one RCCL communicator over the ranks of a job (ncclCommInitRank), bound by the library at run time.  Rank 0 calls
    ShardComm.unique_id() and hands the 128 bytes to the other ranks by any host channel; then every rank constructs its ShardComm."""

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        buf = (C.c_char * 128)()
        rc = _ffi.lib().rp_comm_unique_id(C.byref(buf))
        if rc != 0:
            raise RapierHipError(f"rp_comm_unique_id failed ({rc}): {_ffi.lib().rp_comm_last_error(None).decode()}")
        return bytes(buf)

    def __init__(self, unique_id: bytes, world_size: int, rank: int, device: int = 0):
        import ctypes as C
        assert len(unique_id) == 128
        self._lib = _ffi.lib()
        self.world_size, self.rank, self.device = int(world_size), int(rank), int(device)
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        p = C.c_void_p()
        rc = self._lib.rp_comm_create(C.byref(buf), self.world_size, self.rank, self.device, C.byref(p))
        if rc != 0 or not p.value:
            raise RapierHipError(f"rp_comm_create failed ({rc}): {self._lib.rp_comm_last_error(None).decode()}")
        self._ptr = p

    def close(self):
        if getattr(self, "_ptr", None):
            self._lib.rp_comm_destroy(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def step_many(worlds, steps: int = 1):
    """rp_step_many: `steps` steps of every world, enqueued back to back on the worlds' own streams before anything is waited for
    (small worlds that share ONE parameter set are better served as the sub-worlds of one world: scenes.batch)."""
    worlds = list(worlds)
    if not worlds:
        return
    arr = (C.c_void_p * len(worlds))(*[w._ptr for w in worlds])
    r = worlds[0]._lib.rp_step_many(arr, len(worlds), int(steps))
    if r != 0:
        for w in worlds:
            _check(w._ptr, r, "rp_step_many")
