// rp_pairs.h — per-collider / per-pair device helpers shared by the broad-phase, narrow-phase and
// fast-path kernels.
#pragma once
#include "rp_world.h"

// member of the active set: an awake dynamic or kinematic body (IslandManager::active_bodies)
RP_DEV bool flags_active(int fl) { return (fl & RP_BF_TYPE_MASK) != RP_BODY_FIXED && !(fl & RP_BF_SLEEPING); }
RP_DEV bool flags_dyn_awake(int fl) { return (fl & (RP_BF_TYPE_MASK | RP_BF_SLEEPING)) == RP_BODY_DYNAMIC; }
RP_DEV bool body_active(const DevWorld &w, int b) { return b >= 0 && flags_active(w.b_flags[b]); }
RP_DEV bool body_dyn_awake(const DevWorld &w, int b) { return b >= 0 && flags_dyn_awake(w.b_flags[b]); }
RP_DEV bool body_sleeping(const DevWorld &w, int b) { if (b < 0) return false; int fl = w.b_flags[b]; return (fl & RP_BF_TYPE_MASK) != RP_BODY_FIXED && (fl & RP_BF_SLEEPING); }
RP_DEV int cur_step(const DevWorld &w) { return w.flags[FL_STEP] + 1; } // 1-based number of the step in progress
// pair_solver_hints count cleared by clear_asleep_pair_solver_hint_counts_of (solver_graph.rs:21-49): one of the
// pair's bodies fell asleep after the hint was last computed (only meaningful when w.sleep_enabled)
RP_DEV bool pair_hint_cleared(const DevWorld &w, int s, int2 rb) {
    int s1 = rb.x >= 0 ? w.b_slept_at[rb.x] : 0, s2 = rb.y >= 0 ? w.b_slept_at[rb.y] : 0;
    return (s1 > s2 ? s1 : s2) >= w.p_hint_seq[s];
}
// a manifold the solver takes: for_each_desired_manifold (solver_graph.rs:517-571) + qualify_manifold_bqi (:101-124)
RP_DEV bool pair_selected(const DevWorld &w, int s) {
    if (w.p_c1[s] < 0 || w.p_nsc[s] == 0) return false;
    if (!w.sleep_enabled) return true;
    int2 rb = w.p_rb[s];
    if (pair_hint_cleared(w, (w.p_pflags[s] & RP_PF_AUX) ? w.p_aux[s].x : s, rb)) return false; // (a cluster of a composite pair: the PAIR's hint)
    return body_dyn_awake(w, rb.x) || body_dyn_awake(w, rb.y); // PAIR_HINT_DYN_BIT: at least one awake DYNAMIC body
}

// unlink_contact -> journal_removal (persistent.rs:331-343, :395-418): the endpoints of a touching pair that stopped touching or was
// deleted wait for resolve_removals (rp_sleep.hip).  phase 1 = broad-phase deletion, 2 = end-touch transition (0 = joint edits, appended
// by the host); the key orders a phase like the oracle does (ascending collider pair).  Self-loops and parentless sides are not recorded.
RP_DEV void pi_journal(const DevWorld &w, int b1, int b2, int phase, int c1, int c2) {
    if (b1 == b2 || b1 < 0 || b2 < 0) return;
    int k = atomicAdd(&w.flags[FL_PJ_COUNT], 1);
    if (k < w.pj_cap) { w.pj_key[k] = ((unsigned long long)phase << 62) | ((unsigned long long)(unsigned)c1 << 31) | (unsigned long long)(unsigned)c2; w.pj_b[k] = make_int2(b1, b2); }
}

// CollisionEvent queue (EventHandler::handle_collision_event, event_handler.rs:94-130)
RP_DEV bool pair_wants_collision_events(const DevWorld &w, int c1, int c2) {
    return ((__float_as_int(w.c_events[c1].x) | __float_as_int(w.c_events[c2].x)) & RP_EVENTS_COLLISION) != 0;
}
RP_DEV bool pair_is_sensor(const DevWorld &w, int c1, int c2) { return ((__float_as_int(w.c_events[c1].x) | __float_as_int(w.c_events[c2].x)) & RP_EVENTS_SENSOR_BIT) != 0; }
RP_DEV void push_collision_event(const DevWorld &w, int c1, int c2, int started, int flags, int step) {
    int k = atomicAdd(&w.flags[FL_EV_COL], 1);
    if (k < w.ev_cap) w.ev_col[k] = make_int4(c1, c2, started | (flags << 8), step);
}

RP_DEV Pose collider_world_pose(const DevWorld &w, int i) {
    int parent = w.c_parent[i];
    Pose lp; lp.r = q4(w.c_lrot[i]); lp.t = v3(w.c_lpos[i]);
    if (parent < 0) return lp;
    Pose bp; bp.r = q4(w.b_rot[parent]); bp.t = v3(w.b_pos[parent]);
    return pose_mul(bp, lp);
}

// Collider world pose + fat AABB maintenance (BroadPhaseBvh::set_aabb; advance_to_final_positions
// substep.rs:103-119).  One thread per collider.  Runs at the START of a step (the reference runs it
// at the end of the previous one; nothing reads collider poses in between).
// Capsule::aabb: the transformed segment's box loosened by the radius (then by the collision margin).  Kept out of line: the
// cuboid / ball worlds of the hot kernels (k_island_solve validates fat AABBs in its idle lanes) must not pay for it.
__device__ __noinline__ void capsule_collision_aabb(Pose pos, float4 he, float loosen, V3 &mn, V3 &mx) {
    V3 e = capsule_axis_dir((int)he.z);
    V3 pa = pose_tp(pos, e * -he.x), pb = pose_tp(pos, e * he.x);
    V3 r = v3(he.y, he.y, he.y);
    mn = (v3(rp_min(pa.x, pb.x), rp_min(pa.y, pb.y), rp_min(pa.z, pb.z)) - r) - v3(loosen, loosen, loosen);
    mx = (v3(rp_max(pa.x, pb.x), rp_max(pa.y, pb.y), rp_max(pa.z, pb.z)) + r) + v3(loosen, loosen, loosen);
}
// HalfSpace::aabb: half of the float range in every direction, wherever the plane is
__device__ __noinline__ void halfspace_collision_aabb(float loosen, V3 &mn, V3 &mx) {
    const float h = 3.402823466e+38f / 2.0f;
    mn = v3(-h, -h, -h) - v3(loosen, loosen, loosen);
    mx = v3(h, h, h) + v3(loosen, loosen, loosen);
}
RP_DEV bool collider_update_one(const DevWorld &w, int i) { // true = the fat AABB was rewritten
    Pose pos = collider_world_pose(w, i);
    bool finite = isfinite(pos.t.x) && isfinite(pos.t.y) && isfinite(pos.t.z) && isfinite(pos.r.x) && isfinite(pos.r.y) &&
                  isfinite(pos.r.z) && isfinite(pos.r.w);
    if (!finite) { atomicAdd(&w.flags[FL_QUARANTINE], 1); return false; }
    w.c_pos[i] = f4(pos.t, 0.0f);
    w.c_rot[i] = f4(pos.r);
    float4 he = w.c_he[i];
    V3 h;
    if (w.c_shape[i] == RP_SHAPE_CUBOID || w.c_shape[i] >= RP_SHAPE_CYLINDER) { // (Cylinder / Cone::aabb = local_aabb().transform_by(pos): he = (r, hh, r))
        float m[3][3]; quat_to_mat(pos.r, m);
        h = v3(fabsf(m[0][0]) * he.x + fabsf(m[0][1]) * he.y + fabsf(m[0][2]) * he.z,
               fabsf(m[1][0]) * he.x + fabsf(m[1][1]) * he.y + fabsf(m[1][2]) * he.z,
               fabsf(m[2][0]) * he.x + fabsf(m[2][1]) * he.y + fabsf(m[2][2]) * he.z);
    } else {
        h = v3(he.x, he.x, he.x);
    }
    if (w.c_shape[i] >= RP_SHAPE_ROUND_CUBOID) { const float b = w.c_mat[i].w; h = h + v3(b, b, b); } // RoundShape::aabb = inner.aabb(pos).loosened(border_radius)
    float loosen = w.prm.prediction / 2.0f;
    V3 mn = pos.t - h - v3(loosen, loosen, loosen);
    V3 mx = pos.t + h + v3(loosen, loosen, loosen);
    if (w.c_shape[i] == RP_SHAPE_CAPSULE) capsule_collision_aabb(pos, he, loosen, mn, mx);
    if (w.c_shape[i] == RP_SHAPE_HALFSPACE) halfspace_collision_aabb(loosen, mn, mx);
    float4 fmn = w.c_fatmin[i], fmx = w.c_fatmax[i];
    bool inside = fmn.x <= mn.x && fmn.y <= mn.y && fmn.z <= mn.z && fmx.x >= mx.x && fmx.y >= mx.y && fmx.z >= mx.z;
    if (!inside) {
        float s = w.prm.bp_skin;
        w.c_fatmin[i] = f4(mn - v3(s, s, s), 0.0f);
        w.c_fatmax[i] = f4(mx + v3(s, s, s), 0.0f);
        w.flags[FL_BP_DIRTY] = 1;
        // queued once per broad-phase pass for the incremental update (rp_broadphase.hip)
        const int stamp = w.flags[FL_BP_SEQ] + 1;
        if (w.c_chgstamp[i] != stamp) { // ONE atomic per wavefront on the list's counter (a third of b3d_joint_grid's 10,000 colliders queue every pass: 3,559 same-address atomics)
            w.c_chgstamp[i] = stamp;
            w.c_fatold_min[i] = fmn; w.c_fatold_max[i] = fmx; // what the pair set of the last pass was computed from (bp_incr_insert: pairs that already exist)
            const unsigned long long m = __ballot(1);
            const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
            int k = 0;
            if (lane == leader) k = atomicAdd(&w.flags[FL_BP_NCHG], __popcll(m));
            k = __shfl(k, leader, 64) + __popcll(m & ((1ull << lane) - 1ull));
            if (k < w.n_colliders) w.bp_chg_list[k] = i;
        }
        // shard guard: islands are sharded over GPUs without any exchange, which is only sound while no body of this shard comes near
        // a body of another one — a rewritten fat AABB that overlaps a box another shard occupies ends the run with an error
        if (w.sg_bmin && w.c_parent[i] >= 0 && (w.b_flags[w.c_parent[i]] & RP_BF_TYPE_MASK) != RP_BODY_FIXED && w.c_shape[i] != RP_SHAPE_HALFSPACE) {
            int lo[3], hi[3];
            float g = s;                                      // + the way the body travels before the caller looks at the hits
            if (w.sg_horizon > 0.0f) { const float4 lv = w.b_linvel[w.c_parent[i]]; g += w.sg_horizon * sqrtf(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z); }
            const float a[3] = {mn.x - g, mn.y - g, mn.z - g}, b[3] = {mx.x + g, mx.y + g, mx.z + g};
            bool outside = false;
            for (int k = 0; k < 3; ++k) {
                lo[k] = (int)floorf((a[k] - w.sg_origin[k]) * w.sg_inv_cell); hi[k] = (int)floorf((b[k] - w.sg_origin[k]) * w.sg_inv_cell);
                outside |= hi[k] < 0 || lo[k] > w.sg_dims[k] - 1;
                lo[k] = lo[k] < 0 ? 0 : lo[k]; hi[k] = hi[k] > w.sg_dims[k] - 1 ? w.sg_dims[k] - 1 : hi[k];
            }
            bool foreign = false;
            if (!outside)
                for (int z = lo[2]; z <= hi[2]; ++z) for (int y = lo[1]; y <= hi[1]; ++y) for (int x = lo[0]; x <= hi[0]; ++x) {
                    const int c = (z * w.sg_dims[1] + y) * w.sg_dims[0] + x;
                    for (int k = w.sg_cell_start[c]; k < w.sg_cell_start[c + 1]; ++k) {
                        const int bx = w.sg_cell_items[k];
                        const float4 bmn = w.sg_bmin[bx], bmx = w.sg_bmax[bx];
                        foreign |= a[0] <= bmx.x && bmn.x <= b[0] && a[1] <= bmx.y && bmn.y <= b[1] && a[2] <= bmx.z && bmn.z <= b[2];
                    }
                }
            if (foreign) { atomicOr(&w.flags[FL_OVERFLOW], RP_OVF_SHARD); w.sg_hit[w.c_parent[i]] = 1; } // (which bodies: rp_world_shard_guard_take_hits)
        }
    }
    return !inside;
}

// Would collider_update_one rewrite this collider's fat AABB (i.e. must the broad phase run)?  Read-only.
RP_DEV bool collider_left_fat_aabb(const DevWorld &w, int i) {
    Pose pos = collider_world_pose(w, i);
    bool finite = isfinite(pos.t.x) && isfinite(pos.t.y) && isfinite(pos.t.z) && isfinite(pos.r.x) && isfinite(pos.r.y) &&
                  isfinite(pos.r.z) && isfinite(pos.r.w);
    if (!finite) return true;
    float4 he = w.c_he[i];
    V3 h;
    if (w.c_shape[i] == RP_SHAPE_CUBOID || w.c_shape[i] >= RP_SHAPE_CYLINDER) { // (Cylinder / Cone::aabb = local_aabb().transform_by(pos): he = (r, hh, r))
        float m[3][3]; quat_to_mat(pos.r, m);
        h = v3(fabsf(m[0][0]) * he.x + fabsf(m[0][1]) * he.y + fabsf(m[0][2]) * he.z,
               fabsf(m[1][0]) * he.x + fabsf(m[1][1]) * he.y + fabsf(m[1][2]) * he.z,
               fabsf(m[2][0]) * he.x + fabsf(m[2][1]) * he.y + fabsf(m[2][2]) * he.z);
    } else {
        h = v3(he.x, he.x, he.x);
    }
    if (w.c_shape[i] >= RP_SHAPE_ROUND_CUBOID) { const float b = w.c_mat[i].w; h = h + v3(b, b, b); } // RoundShape::aabb = inner.aabb(pos).loosened(border_radius)
    float loosen = w.prm.prediction / 2.0f;
    V3 mn = pos.t - h - v3(loosen, loosen, loosen);
    V3 mx = pos.t + h + v3(loosen, loosen, loosen);
    if (w.c_shape[i] == RP_SHAPE_CAPSULE) capsule_collision_aabb(pos, he, loosen, mn, mx);
    if (w.c_shape[i] == RP_SHAPE_HALFSPACE) halfspace_collision_aabb(loosen, mn, mx);
    float4 fmn = w.c_fatmin[i], fmx = w.c_fatmax[i];
    bool inside = fmn.x <= mn.x && fmn.y <= mn.y && fmn.z <= mn.z && fmx.x >= mx.x && fmx.y >= mx.y && fmx.z >= mx.z;
    return !inside;
}
// Contact recycling test — pair_update.rs:111-171, contact_pair.rs:284-325.  true = the pair keeps
// last step's manifold (no narrow-phase work).
RP_DEV bool pair_recycle_ok(const DevWorld &w, int s, const Pose &pc1, const Pose &pc2, const Pose &pos12) {
    if (!(w.prm.recycle_distance > 0.0f) || !(w.p_pflags[s] & RP_PF_RECYCLE)) return false;
    Pose base; base.t = v3(w.r_t[s]); base.r = q4(w.r_r[s]);
    float4 misc = w.p_misc[s];
    float trans = len(pos12.t - base.t);
    Q4 d = qmul(pos12.r, qconj(base.r));
    float drift = trans + 2.0f * len(v3(d.x, d.y, d.z)) * misc.y;
    float ca = qdot(q4(w.r_rot1[s]), pc1.r), cb = qdot(q4(w.r_rot2[s]), pc2.r);
    float rot_cos = rp_min(2.0f * ca * ca - 1.0f, 2.0f * cb * cb - 1.0f);
    return drift <= misc.z && rot_cos > 0.98f;
}
// ImpulseJointSet::joints_between(b1, b2).any(|j| !j.data.contacts_enabled) — pair_update.rs:191-201
RP_DEV bool joints_disable_contacts(const DevWorld &w, int b1, int b2) {
    if (w.n_nc == 0 || b1 < 0 || b2 < 0) return false;
    unsigned lo = (unsigned)(b1 < b2 ? b1 : b2), hi = (unsigned)(b1 < b2 ? b2 : b1);
    unsigned long long key = ((unsigned long long)lo << 32) | hi;
    int a = 0, b = w.n_nc - 1;
    while (a <= b) { int m = (a + b) >> 1; unsigned long long k = w.nc_keys[m]; if (k == key) return true; if (k < key) a = m + 1; else b = m - 1; }
    return false;
}
RP_DEV Pose collider_world_pose_of(const DevWorld &w, int i, int parent) { // parent = c_parent[i], already known
    Pose lp; lp.r = q4(w.c_lrot[i]); lp.t = v3(w.c_lpos[i]);
    if (parent < 0) return lp;
    Pose bp; bp.r = q4(w.b_rot[parent]); bp.t = v3(w.b_pos[parent]);
    return pose_mul(bp, lp);
}
// The two tests of the fused step's validators (rp_island_stages.h) in FLAT form: every load that depends on the same index is issued
// at once and nothing is decided before the last one is back — the validating lanes walk 3 levels of dependent loads per item instead
// of 6 (they share their SIMD with an issue-bound generate wavefront: 10.6k cycles per item before, profiles/r06_island_stage_cycles.txt).
// Same answers as pair_needs_narrow_phase / collider_left_fat_aabb for every input (a freed slot reads row 0 and answers "no").
RP_DEV bool pair_needs_narrow_phase_flat(const DevWorld &w, int s, int guard_isl = -1, bool *cross = nullptr) {
    const int c1 = w.p_c1[s], c2 = w.p_c2[s], pf = w.p_pflags[s];
    const int2 rb = w.p_rb[s];
    const float4 rt = w.r_t[s], rr = w.r_r[s], misc = w.p_misc[s], q1 = w.r_rot1[s], q2 = w.r_rot2[s];
    const int a = c1 < 0 ? 0 : c1, b = c2 < 0 ? 0 : c2, pa = rb.x < 0 ? 0 : rb.x, pb = rb.y < 0 ? 0 : rb.y;
    const float4 l1r = w.c_lrot[a], l1t = w.c_lpos[a], l2r = w.c_lrot[b], l2t = w.c_lpos[b];
    const float4 b1r = w.b_rot[pa], b1t = w.b_pos[pa], b2r = w.b_rot[pb], b2t = w.b_pos[pb];
    if (cross && c1 >= 0) { // a body that lives in another island of the LDS path (fixed bodies and the bodies of this island: -1 / guard_isl)
        const int i1 = rb.x >= 0 ? w.b_island[pa] : -1, i2 = rb.y >= 0 ? w.b_island[pb] : -1;
        if ((i1 >= 0 && i1 != guard_isl) || (i2 >= 0 && i2 != guard_isl)) *cross = true;
    }
    if (c1 < 0) return false;
    if (w.has_composite && (pf & RP_PF_AUX)) return false;
    if (w.n_nc && (pf & RP_PF_NO_CONTACT)) return false;
    if (w.has_sensors && pair_is_sensor(w, c1, c2)) return false;
    Pose pc1, pc2; pc1.r = q4(l1r); pc1.t = v3(l1t); pc2.r = q4(l2r); pc2.t = v3(l2t);
    if (rb.x >= 0) { Pose bp; bp.r = q4(b1r); bp.t = v3(b1t); pc1 = pose_mul(bp, pc1); }
    if (rb.y >= 0) { Pose bp; bp.r = q4(b2r); bp.t = v3(b2t); pc2 = pose_mul(bp, pc2); }
    const Pose pos12 = pose_inv_mul(pc1, pc2);
    // pair_recycle_ok on the operands fetched above
    if (!(w.prm.recycle_distance > 0.0f) || !(pf & RP_PF_RECYCLE)) return true;
    Pose base; base.t = v3(rt); base.r = q4(rr);
    float trans = len(pos12.t - base.t);
    Q4 d = qmul(pos12.r, qconj(base.r));
    float drift = trans + 2.0f * len(v3(d.x, d.y, d.z)) * misc.y;
    float ca = qdot(q4(q1), pc1.r), cb = qdot(q4(q2), pc2.r);
    float rot_cos = rp_min(2.0f * ca * ca - 1.0f, 2.0f * cb * cb - 1.0f);
    return !(drift <= misc.z && rot_cos > 0.98f);
}
// collider_left_fat_aabb for a collider whose parent is known (the body item of a validator: c = b_collider[parent]); cuboids and balls
// only are decided here, every other shape goes to the general form
RP_DEV bool collider_left_fat_aabb_flat(const DevWorld &w, int i, int parent) {
    const float4 lr = w.c_lrot[i], lt = w.c_lpos[i], he = w.c_he[i], fmn = w.c_fatmin[i], fmx = w.c_fatmax[i];
    const int shape = w.c_shape[i];
    const float4 br = w.b_rot[parent], bt = w.b_pos[parent];
    if (shape != RP_SHAPE_CUBOID && shape != RP_SHAPE_BALL) return collider_left_fat_aabb(w, i);
    Pose lp; lp.r = q4(lr); lp.t = v3(lt);
    Pose bp; bp.r = q4(br); bp.t = v3(bt);
    const Pose pos = pose_mul(bp, lp);
    bool finite = isfinite(pos.t.x) && isfinite(pos.t.y) && isfinite(pos.t.z) && isfinite(pos.r.x) && isfinite(pos.r.y) &&
                  isfinite(pos.r.z) && isfinite(pos.r.w);
    if (!finite) return true;
    V3 h;
    if (shape == RP_SHAPE_CUBOID) {
        float m[3][3]; quat_to_mat(pos.r, m);
        h = v3(fabsf(m[0][0]) * he.x + fabsf(m[0][1]) * he.y + fabsf(m[0][2]) * he.z,
               fabsf(m[1][0]) * he.x + fabsf(m[1][1]) * he.y + fabsf(m[1][2]) * he.z,
               fabsf(m[2][0]) * he.x + fabsf(m[2][1]) * he.y + fabsf(m[2][2]) * he.z);
    } else {
        h = v3(he.x, he.x, he.x);
    }
    float loosen = w.prm.prediction / 2.0f;
    V3 mn = pos.t - h - v3(loosen, loosen, loosen);
    V3 mx = pos.t + h + v3(loosen, loosen, loosen);
    bool inside = fmn.x <= mn.x && fmn.y <= mn.y && fmn.z <= mn.z && fmx.x >= mx.x && fmx.y >= mx.y && fmx.z >= mx.z;
    return !inside;
}
RP_DEV bool pair_needs_narrow_phase(const DevWorld &w, int s) {
    int c1 = w.p_c1[s];
    if (c1 < 0) return false;
    if (w.has_composite && (w.p_pflags[s] & RP_PF_AUX)) return false; // a cluster of a composite pair: the parent slot answers for the pair
    if (w.n_nc && (w.p_pflags[s] & RP_PF_NO_CONTACT)) return false; // filtered by a contact-disabling joint: nothing to compute
    int c2 = w.p_c2[s];
    if (w.has_sensors && pair_is_sensor(w, c1, c2)) return false; // intersection-tested by k_sensor_pass on the fast graph
    int2 rb = w.p_rb[s];
    Pose pc1 = collider_world_pose_of(w, c1, rb.x), pc2 = collider_world_pose_of(w, c2, rb.y);
    Pose pos12 = pose_inv_mul(pc1, pc2);
    return !pair_recycle_ok(w, s, pc1, pc2, pos12);
}

// ---- composite pairs: auxiliary pair slots (rp_composite.h) ---------------------------------------------------------------------------
RP_DEV bool shape_is_composite(int sh) { return sh == RP_SHAPE_COMPOUND || sh == RP_SHAPE_TRIMESH; }
RP_DEV bool pair_is_aux(const DevWorld &w, int s) { return (w.p_pflags[s] & RP_PF_AUX) != 0; }
// the serial order of the overflow colour: ascending (collider1, collider2), then the cluster index of a composite pair's further solver
// manifolds (collider indices are below 2^24: rp_colliders_insert) — the order the oracle sweeps it in
RP_DEV unsigned long long pair_order_key(const DevWorld &w, int s) {
    const unsigned long long k = (w.p_pflags[s] & RP_PF_AUX) ? (unsigned long long)(unsigned)w.p_aux[s].y : 0ull;
    return ((unsigned long long)(unsigned)w.p_c1[s] << 34) | ((unsigned long long)(unsigned)w.p_c2[s] << 2) | k;
}
// solver-manifold slot k of pair slot s (k = 0: s itself)
RP_DEV int sm_slot(const DevWorld &w, int s, int k) { const int4 a = w.p_aux[s]; return k == 0 ? s : (k == 1 ? a.x : (k == 2 ? a.y : a.z)); }

// an aux slot leaves the pool (its parent's cluster count shrank, the pair left the plain path's way, or the pair is deleted)
RP_DEV void aux_slot_free(DevWorld &w, int a, bool deferred) {
    if (w.p_nsc[a] > 0) w.flags[FL_LAYOUT_DIRTY] = 1;
    w.p_c1[a] = -1; w.p_c2[a] = -1; w.p_nsc[a] = 0; w.p_npts[a] = 0; w.p_color[a] = RP_COLOR_UNCOLORED; w.p_pflags[a] = 0;
    if (deferred) { int t = atomicAdd(&w.flags[FL_BP_NFREED], 1); w.free_pending[t] = a; return; }
    int t = atomicAdd(&w.flags[FL_FREE_TOP], 1);
    w.free_stack[t] = a;
}
// every aux slot of parent slot s (pair deletion, ContactPair::clear)
RP_DEV void aux_free_all(DevWorld &w, int s, bool deferred) {
    const int4 a = w.p_aux[s];
    if (a.x >= 0) aux_slot_free(w, a.x, deferred);
    if (a.y >= 0) aux_slot_free(w, a.y, deferred);
    if (a.z >= 0) aux_slot_free(w, a.z, deferred);
    w.p_aux[s] = make_int4(-1, -1, -1, 0);
}

// Closes the incremental broad-phase pass of this step (k_bp_rebuild leaves FL_BP_CLOSE behind): the slots its deletions parked go onto
// the free stack, the pass counters advance, the dirty flag falls.  ONE workgroup, first thing in the kernel that follows the pass
// (k_np_test): nothing between the two reads the free stack, FL_BP_DIRTY or the change list.
RP_DEV void bp_close_incremental(const DevWorld &w) {
    if (!w.flags[FL_BP_CLOSE]) return; // (uniform)
    const int nfreed = w.flags[FL_BP_NFREED], ftop = w.flags[FL_FREE_TOP];
    for (int k = threadIdx.x; k < nfreed; k += blockDim.x) w.free_stack[ftop + k] = w.free_pending[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        w.flags[FL_FREE_TOP] = ftop + nfreed; w.flags[FL_BP_NFREED] = 0;
        w.flags[FL_BP_TOMBS] += nfreed; // every tombstone of the pass parked a slot (a composite pair's cluster slots count too: an upper bound, which is all the rehash threshold needs)
        w.flags[FL_BP_NCHG] = 0; w.flags[FL_BP_SEQ] += 1; w.flags[FL_BP_REBUILDS] += 1;
        w.flags[FL_BP_DIRTY] = 0; w.flags[FL_BP_CLOSE] = 0;
    }
}
