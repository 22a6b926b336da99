"""ctypes loader for librapier_hip.so (the C ABI of include/rapier_hip.h).

There is no CPU fallback: if the HIP library is missing or no HIP device is usable the
import / world creation fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RP_HIP_LIB") or os.path.join(_HERE, "librapier_hip.so")  # RP_HIP_LIB: an instrumented build of the same library (tools/)

RP_OK = 0
RP_INVALID_HANDLE = 0xFFFFFFFFFFFFFFFF

# every symbol include/rapier_hip.h declares
SYMBOLS = [
    "rp_world_create", "rp_world_destroy", "rp_last_error", "rp_default_params", "rp_params_get",
    "rp_params_set", "rp_bodies_insert", "rp_colliders_insert", "rp_impulse_joints_insert",
    "rp_impulse_joints_read", "rp_impulse_joints_get", "rp_impulse_joints_set_motor", "rp_impulse_joints_read_motor_impulses", "rp_bodies_remove", "rp_colliders_remove", "rp_impulse_joints_remove",
    "rp_compound_create", "rp_trimesh_create", "rp_heightfield_create",
    "rp_quarantine_read", "rp_step",
    "rp_sync", "rp_bodies_read", "rp_bodies_write", "rp_bodies_add_force", "rp_bodies_apply_impulse", "rp_bodies_wake_up", "rp_bodies_set_additional_solver_iterations", "rp_bodies_is_sleeping", "rp_bodies_persistent_island", "rp_bodies_proximity_group", "rp_world_set_shard_guard", "rp_world_shard_guard_take_hits", "rp_world_max_linear_speed", "rp_world_set_shard_guard_horizon", "rp_world_begin_subworld", "rp_step_many", "rp_bodies_handles", "rp_colliders_handles", "rp_impulse_joints_handles", "rp_convex_polyhedron_create", "rp_convex_polyhedron_read", "rp_bodies_set_next_kinematic_position", "rp_num_bodies", "rp_contacts_read",
    "rp_collision_events_read", "rp_intersection_pairs_read", "rp_contact_force_events_read", "rp_counters_enable", "rp_counters_read", "rp_solver_loop_time_ms",
    "rp_comm_unique_id", "rp_comm_create", "rp_comm_destroy", "rp_comm_last_error", "rp_world_set_global_ids", "rp_world_pack_bodies", "rp_shard_all_gather",
]


class Counters(C.Structure):
    """rp_counters — mirror of the reference's Counters (src/counters/mod.rs:18)."""
    _fields_ = [(n, C.c_float) for n in (
        "step_time_ms", "collision_detection_ms", "broad_phase_ms", "narrow_phase_ms",
        "island_construction_ms", "solver_ms", "velocity_assembly_ms", "velocity_resolution_ms",
        "velocity_update_ms")] + [(n, C.c_int32) for n in (
        "num_pairs", "num_manifolds", "num_solver_contacts", "num_colors", "num_parallel_stages",
        "num_dynamic_bodies", "bp_rebuilds", "full_updates", "overflow_flags", "quarantined",
        "fast_steps", "full_steps", "replayed_steps", "num_sleeping_bodies", "ccd_active_count", "ccd_clamp_count", "num_tiles", "tile_sweeps", "bp_large_list", "lean_steps", "fused_steps", "num_islands", "num_global_bodies", "fused_disabled", "fused_launches", "joint_net_steps", "joint_net_disabled", "tile_step_steps")]


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"rapier_amd: HIP library not built ({LIB_PATH} missing). Run __graft_entry__.build() "
            "or `make -C rapier_amd/csrc`. There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, f32 = C.c_void_p, C.c_int32, C.c_uint32, C.c_float
    L.rp_world_create.argtypes = [vp, vp, i32, C.POINTER(vp)]
    L.rp_world_create.restype = i32
    L.rp_world_destroy.argtypes = [vp]
    L.rp_world_destroy.restype = i32
    L.rp_last_error.argtypes = [vp]
    L.rp_last_error.restype = C.c_char_p
    L.rp_default_params.argtypes = [vp]
    L.rp_default_params.restype = None
    L.rp_params_get.argtypes = [vp, vp]
    L.rp_params_set.argtypes = [vp, vp]
    L.rp_bodies_insert.argtypes = [vp, i32, vp, vp]
    L.rp_colliders_insert.argtypes = [vp, i32, vp, vp, vp]
    L.rp_impulse_joints_insert.argtypes = [vp, i32, vp, vp]
    L.rp_impulse_joints_read.argtypes = [vp, i32, vp, vp, vp]
    L.rp_impulse_joints_set_motor.argtypes = [vp, i32, vp, vp, vp]
    L.rp_impulse_joints_get.argtypes = [vp, i32, vp, vp]
    L.rp_impulse_joints_read_motor_impulses.argtypes = [vp, i32, vp, vp]
    L.rp_bodies_remove.argtypes = [vp, i32, vp]
    L.rp_colliders_remove.argtypes = [vp, i32, vp]
    L.rp_impulse_joints_remove.argtypes = [vp, i32, vp]
    L.rp_quarantine_read.argtypes = [vp, i32, vp]
    L.rp_step.argtypes = [vp, u32]
    L.rp_sync.argtypes = [vp]
    L.rp_bodies_read.argtypes = [vp, i32, vp, vp, vp]
    L.rp_bodies_write.argtypes = [vp, i32, vp, vp, vp]
    L.rp_bodies_add_force.argtypes = [vp, i32, vp, vp, vp, i32]
    L.rp_bodies_apply_impulse.argtypes = [vp, i32, vp, vp, vp]
    L.rp_bodies_wake_up.argtypes = [vp, i32, vp, i32]
    L.rp_bodies_set_additional_solver_iterations.argtypes = [vp, i32, vp, vp]
    L.rp_bodies_is_sleeping.argtypes = [vp, i32, vp, vp]
    L.rp_bodies_persistent_island.argtypes = [vp, i32, vp, vp]
    L.rp_bodies_proximity_group.argtypes = [vp, i32, vp, vp]
    L.rp_world_set_shard_guard.argtypes = [vp, i32, vp, vp]
    L.rp_world_shard_guard_take_hits.argtypes = [vp, i32, vp]; L.rp_world_shard_guard_take_hits.restype = i32
    L.rp_world_max_linear_speed.argtypes = [vp, vp]; L.rp_world_max_linear_speed.restype = i32
    L.rp_world_set_shard_guard_horizon.argtypes = [vp, C.c_float]; L.rp_world_set_shard_guard_horizon.restype = i32
    L.rp_world_begin_subworld.argtypes = [vp]; L.rp_world_begin_subworld.restype = i32
    L.rp_step_many.argtypes = [vp, i32, i32]; L.rp_step_many.restype = i32
    L.rp_bodies_handles.argtypes = [vp, i32, vp]; L.rp_bodies_handles.restype = i32
    L.rp_convex_polyhedron_create.argtypes = [vp, i32, vp, i32, vp, vp]; L.rp_convex_polyhedron_create.restype = i32
    L.rp_convex_polyhedron_read.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]; L.rp_convex_polyhedron_read.restype = i32
    L.rp_compound_create.argtypes = [vp, i32, vp, vp]; L.rp_trimesh_create.argtypes = [vp, i32, vp, i32, vp, vp]; L.rp_heightfield_create.argtypes = [vp, i32, i32, vp, vp, vp]
    L.rp_colliders_handles.argtypes = [vp, i32, vp]; L.rp_colliders_handles.restype = i32
    L.rp_impulse_joints_handles.argtypes = [vp, i32, vp]; L.rp_impulse_joints_handles.restype = i32
    L.rp_debug_islands.argtypes = [vp, vp, vp, i32, vp]  # debug aid, not in the header
    L.rp_debug_islands.restype = i32
    L.rp_bodies_set_next_kinematic_position.argtypes = [vp, i32, vp, vp]
    L.rp_num_bodies.argtypes = [vp]
    L.rp_contacts_read.argtypes = [vp, i32, vp, vp, vp]
    L.rp_collision_events_read.argtypes = [vp, i32, vp]
    L.rp_contact_force_events_read.argtypes = [vp, i32, vp]
    L.rp_intersection_pairs_read.argtypes = [vp, i32, vp]
    L.rp_counters_enable.argtypes = [vp, i32]
    L.rp_counters_read.argtypes = [vp, vp]
    L.rp_solver_loop_time_ms.argtypes = [vp, C.POINTER(f32), C.POINTER(i32)]
    L.rp_comm_unique_id.argtypes = [vp]
    L.rp_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.rp_comm_destroy.argtypes = [vp]
    L.rp_comm_last_error.argtypes = [vp]; L.rp_comm_last_error.restype = C.c_char_p
    L.rp_world_set_global_ids.argtypes = [vp, i32, vp]
    L.rp_world_pack_bodies.argtypes = [vp, C.POINTER(vp), C.POINTER(i32)]
    L.rp_shard_all_gather.argtypes = [vp, vp, i32, C.c_int64, vp, vp, vp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("rp_last_error", "rp_default_params", "rp_comm_last_error"):
            fn.restype = i32
    _LIB = L
    return L
