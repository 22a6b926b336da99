"""ctypes binding of the CPU oracle (oracle/librapier_oracle.so) — test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rapier_amd import scenes as S

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB = None


class Stats(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "num_pairs", "num_active_manifolds", "num_solver_contacts", "num_colors_used",
        "num_parallel_colors", "num_full_updates", "num_recycled", "bp_rebuilt")]


def build_oracle() -> str:
    subprocess.run(["make", "-s", "-C", _ORACLE_DIR], check=True)
    return os.path.join(_ORACLE_DIR, "librapier_oracle.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ORACLE_DIR, "librapier_oracle.so")
        srcs = [os.path.join(_ORACLE_DIR, f) for f in os.listdir(_ORACLE_DIR) if f.endswith((".c", ".h")) or f == "Makefile"]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in srcs):
            build_oracle()
        L = C.CDLL(path)
        L.ro_world_new.restype = C.c_void_p
        L.ro_world_new.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_world_free.argtypes = [C.c_void_p]
        L.ro_set_params.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_add_body.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_add_collider.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.ro_add_joint.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_begin_subworld.argtypes = [C.c_void_p]; L.ro_begin_subworld.restype = C.c_int32
        L.ro_step.argtypes = [C.c_void_p, C.c_int32]
        L.ro_num_bodies.argtypes = [C.c_void_p]
        L.ro_read_bodies.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ro_get_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_total_contact_impulse.argtypes = [C.c_void_p]
        L.ro_total_contact_impulse.restype = C.c_float
        L.ro_dump_manifolds.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ro_combine_coefficient.argtypes = [C.c_float, C.c_float, C.c_int32, C.c_int32]
        L.ro_combine_coefficient.restype = C.c_float
        L.ro_set_body_vel.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.ro_set_threads.argtypes = [C.c_int32]
        L.ro_set_body_pose.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.ro_set_next_kinematic_position.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.ro_collision_events_drain.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.ro_force_events_drain.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.ro_body_mass_props.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.ro_add_force.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
        L.ro_apply_impulse.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.ro_wake_up.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.ro_read_sleeping.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_remove_body.argtypes = [C.c_void_p, C.c_int32]
        L.ro_body_generation.argtypes = [C.c_void_p, C.c_int32]; L.ro_body_generation.restype = C.c_uint32
        L.ro_collider_generation.argtypes = [C.c_void_p, C.c_int32]; L.ro_collider_generation.restype = C.c_uint32
        L.ro_remove_collider.argtypes = [C.c_void_p, C.c_int32]
        L.ro_remove_joint.argtypes = [C.c_void_p, C.c_int32]
        L.ro_num_joints.argtypes = [C.c_void_p]
        L.ro_read_joints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ro_set_joint_motor.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.ro_read_joint_motor_impulses.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_read_island_labels.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_read_island_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_read_slept_at.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_read_island_state.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.ro_read_island_globals.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_set_additional_solver_iterations.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.ro_set_ccd_enabled.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.ro_read_ccd_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_read_solve_group_extras.argtypes = [C.c_void_p, C.c_void_p]
        L.ro_set_collider_sensor.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.ro_intersection_pair.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        _LIB = L
    return _LIB


def set_threads(n: int):
    """OpenMP threads of the oracle's data-parallel loops (results do not depend on it)."""
    lib().ro_set_threads(int(n))


def hull_triangles(points) -> np.ndarray:
    """outward-wound triangles of the convex hull of `points` (scipy / Qhull): test-side stand-in for parry's convex_hull"""
    from scipy.spatial import ConvexHull
    pts = np.asarray(points, np.float64)
    h = ConvexHull(pts)
    tris = h.simplices.copy()
    c = pts[h.vertices].mean(0)
    for k, (a, b, cc) in enumerate(tris):
        n = np.cross(pts[b] - pts[a], pts[cc] - pts[a])
        if n @ (pts[a] - c) < 0:
            tris[k] = (a, cc, b)
    return np.ascontiguousarray(tris, np.uint32)


class OracleWorld:
    """The oracle stepped through the same Scene descriptors as the product."""

    def __init__(self, scene: S.Scene):
        L = lib()
        params = np.ascontiguousarray(scene.params)
        grav = np.asarray(scene.gravity, dtype=np.float32)
        self._w = L.ro_world_new(params.ctypes.data, grav.ctypes.data)
        bodies = scene.body_array()
        for pts, tris in getattr(scene, "polyhedra", []):
            if self.add_convex_polyhedron(pts, tris) < 0:
                raise ValueError("oracle: not a closed convex triangle mesh")
        for comp in getattr(scene, "composites", []):
            if self.add_composite(comp) < 0:
                raise ValueError("oracle: invalid composite shape")
        cols = scene.collider_array()
        parents = scene.parent_array()
        # a batch (scenes.batch): every sub-world's bodies and colliders go in behind its own ro_begin_subworld, like PhysicsWorld.from_scene
        starts = list(getattr(scene, "subworlds", None) or [(0, 0, 0)]) + [(len(bodies), len(cols), 0)]
        for k in range(len(starts) - 1):
            (b0, c0, _), (b1, c1, _) = starts[k], starts[k + 1]
            if k:
                L.ro_begin_subworld(self._w)
            for i in range(b0, b1):
                bi = L.ro_add_body(self._w, bodies[i:i + 1].ctypes.data)
                if int(bodies["additional_solver_iterations"][i]):  # trailing descriptor field (the oracle's struct ends before it)
                    L.ro_set_additional_solver_iterations(self._w, bi, int(bodies["additional_solver_iterations"][i]))
                if int(bodies["ccd_enabled"][i]):
                    L.ro_set_ccd_enabled(self._w, bi, 1)
            for i in range(c0, c1):
                ci = L.ro_add_collider(self._w, cols[i:i + 1].ctypes.data, int(parents[i]))
                if int(cols["sensor"][i]):  # the descriptor's trailing `sensor` field (the oracle's struct ends before it)
                    L.ro_set_collider_sensor(self._w, ci, 1)
        joints = scene.joint_array()
        for i in range(len(joints)):
            if L.ro_add_joint(self._w, joints[i:i + 1].ctypes.data) < 0:
                raise ValueError("oracle: unsupported joint")
        self.n = len(bodies)

    def add_composite(self, comp) -> int:
        """ro_add_compound / ro_add_trimesh / ro_add_heightfield from a Scene.composites entry"""
        L = lib()
        for f in (L.ro_add_compound, L.ro_add_trimesh, L.ro_add_heightfield):
            f.restype = C.c_int32
        if comp[0] == "compound":
            parts = np.ascontiguousarray(comp[1], S.COLLIDER_DTYPE)
            return int(L.ro_add_compound(C.c_void_p(self._w), C.c_int32(len(parts)), C.c_void_p(parts.ctypes.data)))
        if comp[0] == "trimesh":
            v, t = np.ascontiguousarray(comp[1], np.float32), np.ascontiguousarray(comp[2], np.uint32)
            return int(L.ro_add_trimesh(C.c_void_p(self._w), C.c_int32(len(v)), C.c_void_p(v.ctypes.data), C.c_int32(len(t)), C.c_void_p(t.ctypes.data)))
        h, sc = np.ascontiguousarray(comp[1], np.float32), np.ascontiguousarray(comp[2], np.float32)
        return int(L.ro_add_heightfield(C.c_void_p(self._w), C.c_int32(h.shape[0]), C.c_int32(h.shape[1]), C.c_void_p(h.ctypes.data), C.c_void_p(sc.ctypes.data)))

    def pair_clusters(self, c1: int, c2: int):
        """(number of clusters: 0 = plain path, -1 = no such pair; solver contacts per solver manifold)"""
        L = lib(); L.ro_pair_clusters.restype = C.c_int32
        out = np.zeros(4, np.int32)
        n = int(L.ro_pair_clusters(C.c_void_p(self._w), C.c_int32(c1), C.c_int32(c2), C.c_int32(4), C.c_void_p(out.ctypes.data)))
        return n, out[:max(n, 1)].tolist()

    def add_convex_polyhedron(self, points, triangles=None) -> int:
        """ro_add_convex_polyhedron; without triangles the hull comes from scipy (Qhull), wound outwards — the oracle has no hull code"""
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        tris = hull_triangles(pts) if triangles is None else np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
        L = lib()
        L.ro_add_convex_polyhedron.restype = C.c_int32
        return int(L.ro_add_convex_polyhedron(C.c_void_p(self._w), C.c_int32(len(pts)), C.c_void_p(pts.ctypes.data), C.c_int32(len(tris)), C.c_void_p(tris.ctypes.data)))

    def read_convex_polyhedron(self, pid: int) -> dict:
        """the canonical form the oracle built (ro_polyhedron.h)"""
        L = lib()
        cnt = np.zeros(4, np.int32)
        L.ro_read_convex_polyhedron(C.c_void_p(self._w), C.c_int32(pid), C.c_void_p(cnt.ctypes.data), None, None, None, None, None, None, None)
        nv, nf, nl, ne = (int(x) for x in cnt)
        pts, fn = np.zeros((nv, 3), np.float32), np.zeros((nf, 3), np.float32)
        ff, fc, lv, le, props = np.zeros(nf, np.int32), np.zeros(nf, np.int32), np.zeros(nl, np.int32), np.zeros(nl, np.int32), np.zeros(20, np.float32)
        L.ro_read_convex_polyhedron(C.c_void_p(self._w), C.c_int32(pid), C.c_void_p(cnt.ctypes.data), C.c_void_p(pts.ctypes.data), C.c_void_p(fn.ctypes.data),
                                    C.c_void_p(ff.ctypes.data), C.c_void_p(fc.ctypes.data), C.c_void_p(lv.ctypes.data), C.c_void_p(le.ctypes.data), C.c_void_p(props.ctypes.data))
        return dict(points=pts, face_normals=fn, face_first=ff, face_count=fc, loop_vertex=lv, loop_edge=le, n_edges=ne, props=props)

    def step(self, n: int = 1):
        lib().ro_step(self._w, n)

    @property
    def n(self) -> int:
        """rows of the body arena (free slots included): ro_num_bodies — not a count of insertions, since removed slots are reused"""
        return int(lib().ro_num_bodies(self._w))

    @n.setter
    def n(self, value):  # (callers that counted insertions themselves: `o.n += 1`)
        pass

    def body_generation(self, body: int) -> int:
        return int(lib().ro_body_generation(self._w, int(body)))

    def collider_generation(self, collider: int) -> int:
        return int(lib().ro_collider_generation(self._w, int(collider)))

    def set_params(self, params):
        """the IntegrationParameters of the steps that follow (the reference takes them per step)"""
        p = np.ascontiguousarray(params)
        lib().ro_set_params(self._w, p.ctypes.data)

    def add_body(self, **kw) -> int:
        """RigidBodySet::insert into the (possibly already stepped) world."""
        b = np.ascontiguousarray(S.body_desc(**kw))
        h = lib().ro_add_body(self._w, b.ctypes.data)
        if int(np.asarray(b["ccd_enabled"]).reshape(-1)[0]):
            lib().ro_set_ccd_enabled(self._w, h, 1)
        self.n += 1
        return h

    def add_collider(self, parent: int, **kw) -> int:
        c = np.ascontiguousarray(S.collider_desc(**kw))
        ci = lib().ro_add_collider(self._w, c.ctypes.data, int(parent))
        if int(np.asarray(c["sensor"]).reshape(-1)[0]):
            lib().ro_set_collider_sensor(self._w, ci, 1)
        return ci

    def read(self):
        pos = np.zeros((self.n, 7), np.float32)
        vel = np.zeros((self.n, 6), np.float32)
        lib().ro_read_bodies(self._w, pos.ctypes.data, vel.ctypes.data)
        return pos, vel

    def stats(self) -> dict:
        s = Stats()
        lib().ro_get_stats(self._w, C.byref(s))
        return {n: getattr(s, n) for n, _ in Stats._fields_}

    def total_contact_impulse(self) -> float:
        return float(lib().ro_total_contact_impulse(self._w))

    def manifolds(self):
        L = lib()
        m = L.ro_dump_manifolds(self._w, 0, None, None, None)
        meta = np.zeros((m, 4), np.int32)
        nrm = np.zeros((m, 3), np.float32)
        imp = np.zeros((m, 4), np.float32)
        L.ro_dump_manifolds(self._w, m, meta.ctypes.data, nrm.ctypes.data, imp.ctypes.data)
        return meta, nrm, imp

    def remove_body(self, body):
        assert lib().ro_remove_body(self._w, int(body)) == 0

    def remove_collider(self, collider):
        assert lib().ro_remove_collider(self._w, int(collider)) == 0

    def remove_joint(self, joint):
        assert lib().ro_remove_joint(self._w, int(joint)) == 0

    def set_joint_motor(self, joint, axis, **motor):
        m = np.ascontiguousarray(S.motor_desc(**motor))
        assert lib().ro_set_joint_motor(self._w, int(joint), int(axis), m.ctypes.data) == 0

    def joint_motor_impulses(self):
        n = lib().ro_num_joints(self._w)
        out = np.zeros((n, 6), np.float32)
        lib().ro_read_joint_motor_impulses(self._w, out.ctypes.data)
        return out

    def read_joints(self):
        n = lib().ro_num_joints(self._w)
        col = np.zeros(n, np.int32)
        imp = np.zeros((n, 3), np.float32)
        lib().ro_read_joints(self._w, col.ctypes.data, imp.ctypes.data)
        return col, imp

    def set_pose(self, body, pos7):
        p = np.ascontiguousarray(pos7, np.float32)
        lib().ro_set_body_pose(self._w, int(body), p.ctypes.data)

    def set_next_kinematic_position(self, body, pos7):
        p = np.ascontiguousarray(pos7, np.float32)
        lib().ro_set_next_kinematic_position(self._w, int(body), p.ctypes.data)

    def collision_events(self):
        """Drain: rows (collider1, collider2, started, flags, step)."""
        n = lib().ro_collision_events_drain(self._w, 0, None)
        out = np.zeros((n, 5), np.int32)
        lib().ro_collision_events_drain(self._w, n, out.ctypes.data)
        return out

    def force_events(self):
        """Drain: (meta rows (collider1, collider2, step, started), value rows of 8 floats)."""
        n = lib().ro_force_events_drain(self._w, 0, None, None)
        meta = np.zeros((n, 4), np.int32)
        vals = np.zeros((n, 8), np.float32)
        lib().ro_force_events_drain(self._w, n, meta.ctypes.data, vals.ctypes.data)
        return meta, vals

    def mass_props(self, body):
        out = np.zeros(11, np.float32)
        lib().ro_body_mass_props(self._w, int(body), out.ctypes.data)
        return out

    def add_force(self, body, force=None, torque=None, reset=False):
        f = None if force is None else np.ascontiguousarray(force, np.float32)
        t = None if torque is None else np.ascontiguousarray(torque, np.float32)
        lib().ro_add_force(self._w, int(body), None if f is None else f.ctypes.data, None if t is None else t.ctypes.data, 1 if reset else 0)

    def apply_impulse(self, body, impulse=None, torque_impulse=None):
        f = None if impulse is None else np.ascontiguousarray(impulse, np.float32)
        t = None if torque_impulse is None else np.ascontiguousarray(torque_impulse, np.float32)
        lib().ro_apply_impulse(self._w, int(body), None if f is None else f.ctypes.data, None if t is None else t.ctypes.data)

    def wake_up(self, body, strong=True):
        lib().ro_wake_up(self._w, int(body), 1 if strong else 0)

    def set_additional_solver_iterations(self, body, n):
        lib().ro_set_additional_solver_iterations(self._w, int(body), int(n))

    def set_sensor(self, collider, on=True):
        lib().ro_set_collider_sensor(self._w, int(collider), 1 if on else 0)

    def intersection_pair(self, c1, c2):
        r = lib().ro_intersection_pair(self._w, int(c1), int(c2))
        return None if r < 0 else bool(r)

    def solve_group_extras(self):
        out = np.zeros(self.n, np.int32)
        lib().ro_read_solve_group_extras(self._w, out.ctypes.data)
        return out

    def island_labels(self):
        out = np.zeros(self.n, np.int32)
        lib().ro_read_island_labels(self._w, out.ctypes.data)
        return out

    ISLAND_STATS = ("merged", "multiway_groups", "removals", "connected", "detached", "hot", "over_budget", "sleeping_deferred",
                    "global_splits", "global_split_pieces", "bids", "bid_ties", "sleep_blocked", "order_dependent", "detach_size_ties",
                    "split_keep_ties")

    def island_stats(self):
        """counters of the persistent-island machinery since world creation (ro_read_island_stats)"""
        out = np.zeros(len(self.ISLAND_STATS), np.int32)
        lib().ro_read_island_stats(self._w, out.ctypes.data)
        return dict(zip(self.ISLAND_STATS, (int(v) for v in out)))

    def island_state(self, island):
        """PersistentIsland: dict(used, nbodies, dirty = constraint_remove_count > 0, denied = split_denied_until, sleeping)"""
        out = np.zeros(5, np.int32)
        lib().ro_read_island_state(self._w, int(island), out.ctypes.data)
        return dict(zip(("used", "nbodies", "dirty", "denied", "sleeping"), (int(v) for v in out)))

    def island_globals(self):
        """(sleep_scan_stamp, pending split island or -1)"""
        out = np.zeros(2, np.int32)
        lib().ro_read_island_globals(self._w, out.ctypes.data)
        return int(out[0]), int(out[1])

    def ccd_counts(self):
        """((body, step) cases of the CCD fast-body criterion, clamped next_positions) since world creation"""
        out = np.zeros(2, np.int32)
        lib().ro_read_ccd_counts(self._w, out.ctypes.data)
        return int(out[0]), int(out[1])

    def slept_at(self):
        out = np.zeros(self.n, np.int32)
        lib().ro_read_slept_at(self._w, out.ctypes.data)
        return out

    def sleeping(self):
        out = np.zeros(self.n, np.int32)
        lib().ro_read_sleeping(self._w, out.ctypes.data)
        return out.astype(bool)

    def set_vel(self, body, linvel, angvel=(0, 0, 0)):
        lv = np.asarray(linvel, np.float32)
        av = np.asarray(angvel, np.float32)
        lib().ro_set_body_vel(self._w, body, lv.ctypes.data, av.ctypes.data)

    def __del__(self):
        try:
            lib().ro_world_free(self._w)
        except Exception:
            pass
