// rp_sleep.hip — sleeping on device: sleep timers, whole-island sleep decision, island-wide wake-up.
//
// Reference: RigidBodyActivation::update_energy (/root/reference/src/dynamics/rigid_body_components.rs:1412-1478),
// the fused active-body pass and its sleep observations (pipeline/physics_pipeline/solve.rs:196-300),
// IslandManager::update_islands / commit_sleeping_chunks (island_manager/manager.rs:335-388, sleep.rs:81-131) and
// IslandManager::wake_up (sleep.rs:31-79).
//
// The reference maintains its persistent islands incrementally on the host (merge on begin-touch / joint link,
// deferred cooldown-throttled splits).  The sleep decision only needs the PARTITION, and on MI355X recomputing it
// is cheaper than maintaining it: a lock-free union-find over the touching pairs of the awake bodies, re-run only
// when the touching set or the awake set changed (FL_LAYOUT_DIRTY).  This is the partition the reference
// converges to once its pending splits are resolved (its split cooldown only DELAYS a sleep by <= 16 steps).
// Label of an island = smallest body index of the component; a sleeping island keeps its label in b_slabel, so
// waking any member wakes every body carrying that label.
//
// Per step (sleep-enabled worlds only; every kernel is one thread per body or per pair slot):
//   after the broad phase : k_wake_spread(0) + k_wake_commit(0)   user wake-ups, pair deletions
//   after the narrow phase: k_wake_spread(1) + k_wake_commit(1)   begin-touch wake-ups
//   then                  : k_sleep_pass = labels (init / union / flatten, only when dirty) + observe, then k_sleep_commit
// and the bucket / island rebuild that follows sees the new awake set.
#include "rp_pairs.h"
#include "rp_gridbar.h"

RP_DEV int slp_ld(int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RP_DEV int slp_find(int *label, int x) {
    int p = slp_ld(&label[x]);
    while (p != x) { x = p; p = slp_ld(&label[x]); }
    return x;
}
RP_DEV void slp_union(int *label, int a, int b) { // the smaller index becomes the root
    for (;;) {
        a = slp_find(label, a); b = slp_find(label, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        if (atomicCAS(&label[a], a, b) == a) return;
    }
}

// Wake requests -> per-island wake marks.  A strong request on an awake body only resets its own timer
// (RigidBodyActivation::wake_up(strong)).
__global__ void k_wake_spread(DevWorld w, int phase) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && phase == 0) { w.flags[FL_N_AWAKE] = 0; w.flags[FL_WAKE_PENDING] = 0; } // recounted by k_sleep_commit; requests consumed below
    if (i >= w.n_bodies) return;
    int r = w.b_wake_req[i];
    if (!r) return;
    w.b_wake_req[i] = 0;
    int fl = w.b_flags[i];
    if ((fl & RP_BF_TYPE_MASK) == RP_BODY_FIXED) return;
    if (fl & RP_BF_SLEEPING) {
        w.lab_wake[w.b_slabel[i]] = cur_step(w);
        w.flags[FL_WAKE_STAMP] = 2 * cur_step(w) + phase;
    } else if (r >= 2) {
        float4 sl = w.b_sleep[i]; sl.x = 0.0f; w.b_sleep[i] = sl;
    }
}
// Whole-island wake: every sleeping body of a marked island wakes with a strong timer reset (sleep.rs:44-70).
__global__ void k_wake_commit(DevWorld w, int phase) {
    if (w.flags[FL_WAKE_STAMP] != 2 * cur_step(w) + phase) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w.n_bodies) return;
    int fl = w.b_flags[i];
    if ((fl & RP_BF_TYPE_MASK) == RP_BODY_FIXED || !(fl & RP_BF_SLEEPING)) return;
    if (w.lab_wake[w.b_slabel[i]] != cur_step(w)) return;
    w.b_flags[i] = fl & ~RP_BF_SLEEPING;
    float4 sl = w.b_sleep[i]; sl.x = 0.0f; w.b_sleep[i] = sl;
    w.flags[FL_LAYOUT_DIRTY] = 1; // the awake set changed: buckets, islands and labels are rebuilt
    if (w.n_joints) w.flags[FL_JOINT_DIRTY] = 1; // ... and so is the joint selection (select_active_interactions)
}
// The user moved a body (rp_bodies_write with a pose): every body that has a pair with it is woken
// (handle_user_changes_on_colliders, pair_management.rs:236-258).
__global__ void k_wake_partners(DevWorld w) {
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < top; s += stride) {
        if (w.p_c1[s] < 0) continue;
        int2 rb = w.p_rb[s];
        bool m1 = rb.x >= 0 && slp_ld(&w.b_wake_req[rb.x]) == 3, m2 = rb.y >= 0 && slp_ld(&w.b_wake_req[rb.y]) == 3;
        if (m1 && rb.y >= 0) atomicMax(&w.b_wake_req[rb.y], 2);
        if (m2 && rb.x >= 0) atomicMax(&w.b_wake_req[rb.x], 2);
    }
}

// ---- sleep-island labels (only when the touching set or the awake set changed) --------------------
RP_DEV void slp_init(DevWorld &w, int gid, int gstride) {
    for (int i = gid; i < w.n_bodies; i += gstride) if (flags_active(w.b_flags[i])) w.b_slabel[i] = i;
}
RP_DEV void slp_union_pairs(DevWorld &w, int gid, int gstride) {
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    for (int s = gid; s < top; s += gstride) {
        if (w.p_c1[s] < 0 || w.p_nsc[s] == 0) continue; // touching pairs link (contacts.rs:352-359), whatever their solver hint
        int2 rb = w.p_rb[s];
        if (body_active(w, rb.x) && body_active(w, rb.y)) slp_union(w.b_slabel, rb.x, rb.y);
    }
    // impulse joints link the islands of their two bodies (ImpulseJointIslandEvent::Link, persistent.rs:13-24)
    for (int j = gid; j < w.n_joints; j += gstride) {
        int b1 = w.j_b1[j], b2 = w.j_b2[j];
        if (body_active(w, b1) && body_active(w, b2)) slp_union(w.b_slabel, b1, b2);
    }
}
RP_DEV void slp_flatten(DevWorld &w, int gid, int gstride) {
    for (int i = gid; i < w.n_bodies; i += gstride) {
        if (!flags_active(w.b_flags[i])) continue;
        int root = slp_find(w.b_slabel, i);
        __hip_atomic_store(&w.b_slabel[i], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// update_body_energy for every awake body + the island observation (an island sleeps once EVERY member is eligible).
// The timer update runs at most once per step NUMBER (b_sleep_stamp): a fast step that observes, then aborts (an island is about to
// fall asleep, k_sleep_check) is replayed on the full graph with the same step number, finds every timer already advanced and only
// repeats the (idempotent) island marks.
RP_DEV void sleep_observe_one(DevWorld &w, int i) {
    if (!flags_active(w.b_flags[i])) return;
    float4 sl = w.b_sleep[i];
    if (w.b_sleep_stamp[i] == cur_step(w)) { if (!(sl.x >= sl.w)) w.lab_awake[w.b_slabel[i]] = cur_step(w); return; }
    w.b_sleep_stamp[i] = cur_step(w);
    if ((w.b_flags[i] & RP_BF_TYPE_MASK) != RP_BODY_DYNAMIC) { // platforms only sleep when both velocities are exactly zero (:1464-1468)
        V3 lv = v3(w.b_linvel[i]), kav = v3(w.b_angvel[i]);
        bool still = dot(lv, lv) == 0.0f && dot(kav, kav) == 0.0f;
        sl.x = still ? sl.x + w.prm.p.dt : 0.0f;
        w.b_sleep[i] = sl;
        if (!(sl.x >= sl.w)) w.lab_awake[w.b_slabel[i]] = cur_step(w);
        return;
    }
    float4 pt = w.b_sprev_t[i];
    Q4 prev_r = q4(w.b_sprev_r[i]);
    V3 pos = v3(w.b_pos[i]); Q4 rot = q4(w.b_rot[i]);
    float max_extent = pt.w;
    w.b_sprev_t[i] = f4(pos, max_extent); w.b_sprev_r[i] = f4(rot);
    float linear_threshold = sl.y * w.prm.p.length_unit;
    V3 av = v3(w.b_angvel[i]);
    float sq_angvel = dot(av, av);
    bool angular_ok;
    if (max_extent > 0.0f) angular_ok = sl.z >= 0.0f && sq_angvel < 1.5707964f * 1.5707964f;
    else angular_ok = sq_angvel < sl.z * fabsf(sl.z);
    float trans = len(pos - v3(pt));
    Q4 d = qmul(rot, qconj(prev_r));
    float drift = trans + 2.0f * len(v3(d.x, d.y, d.z)) * max_extent; // relative_pose_drift, contact_pair.rs:300-323
    bool can_sleep = angular_ok && drift * 0.5f < linear_threshold * w.prm.p.dt;
    sl.x = can_sleep ? sl.x + w.prm.p.dt : 0.0f;
    w.b_sleep[i] = sl;
    if (!(sl.x >= sl.w)) w.lab_awake[w.b_slabel[i]] = cur_step(w);
}
// commit_sleeping_chunks -> RigidBody::sleep (rigid_body.rs:804-807) + clear_asleep_pair_solver_hint_counts_of
RP_DEV void sleep_commit(DevWorld &w, int gid, int gstride) {
    for (int base = 0; base < w.n_bodies; base += gstride) { // wave-uniform trip count: the ballot below needs whole wavefronts
        const int i = base + gid;
        int fl = i < w.n_bodies ? w.b_flags[i] : RP_BODY_FIXED;
        bool active = flags_active(fl);
        bool stays_awake = active && w.lab_awake[w.b_slabel[i]] == cur_step(w);
        // awake bodies left after this pass, one atomic per wavefront (0 = the whole world sleeps: the host may enqueue idle steps)
        unsigned long long awake_mask = __ballot(stays_awake);
        if ((threadIdx.x & 63) == 0 && awake_mask) atomicAdd(&w.flags[FL_N_AWAKE], __popcll(awake_mask));
        if (!active || stays_awake) continue;
        w.b_flags[i] = fl | RP_BF_SLEEPING;
        float4 sl = w.b_sleep[i]; sl.x = sl.w; w.b_sleep[i] = sl;
        w.b_linvel[i] = make_float4(0, 0, 0, 0); w.b_angvel[i] = make_float4(0, 0, 0, 0);
        w.b_slept_at[i] = cur_step(w);
        w.flags[FL_LAYOUT_DIRTY] = 1;
        if (w.n_joints) w.flags[FL_JOINT_DIRTY] = 1;
    }
}
// The sleep pass of a step in TWO launches: (1) the island labels when the touching set or the awake set changed (three passes
// behind grid barriers, rp_gridbar.h — skipped on a clean step) and the per-body observation; (2) the commit, which may only
// run once EVERY member of an island was observed: that dependency is a kernel boundary, cheaper on MI355X than a fenced grid
// barrier that would have to run every step (~4 us against ~7 us).
__global__ void __launch_bounds__(1024) k_sleep_pass(DevWorld w, int fast) {
    if (fast && w.flags[FL_FAST_ABORT]) return; // the fast graph gave up on this step: nothing may change
    const int gid = gbar_item(), gstride = gridDim.x * blockDim.x;
    if (w.flags[FL_LAYOUT_DIRTY]) { // (nothing in this launch writes the flag)
        GridBar bar = gbar_begin(w, 2);
        slp_init(w, gid, gstride);
        gbar_sync(bar);
        slp_union_pairs(w, gid, gstride);
        gbar_sync(bar);
        slp_flatten(w, gid, gstride);
        gbar_sync(bar);
        gbar_end(bar);
    }
    for (int i = gid; i < w.n_bodies; i += gstride) sleep_observe_one(w, i);
}
__global__ void k_sleep_commit(DevWorld w) { sleep_commit(w, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x); }
// fast graph: would the commit put an island to sleep (no member marked it awake this step)?  Then the step needs the full graph
// (the layout changes): abort before anything but the — once-per-step — observation has happened.
__global__ void k_sleep_check(DevWorld w) {
    if (w.flags[FL_FAST_ABORT]) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w.n_bodies || !flags_active(w.b_flags[i])) return;
    if (w.lab_awake[w.b_slabel[i]] != cur_step(w)) w.flags[FL_FAST_ABORT] = 1;
}

// interpolate_kinematic_velocities (substep.rs:242-264): a position-based kinematic body gets the velocity that
// reaches its next_position in one step (RigidBodyPosition::interpolate_velocity, rigid_body_components.rs:147-194).
__global__ void k_kinematic_velocities(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w.n_bodies) return;
    int fl = w.b_flags[i];
    if ((fl & RP_BF_TYPE_MASK) != RP_BODY_KINEMATIC_POSITION || (fl & RP_BF_SLEEPING)) return;
    float inv_dt = w.prm.p.dt == 0.0f ? 0.0f : 1.0f / w.prm.p.dt;
    Pose pos; pos.r = q4(w.b_rot[i]); pos.t = v3(w.b_pos[i]);
    Pose next; next.r = q4(w.b_next_rot[i]); next.t = v3(w.b_next_pos[i]);
    Pose shift; shift.r = q4(0, 0, 0, 1); shift.t = pose_tp(pos, v3(w.b_lcom_invm[i]));
    Pose dpos = pose_mul(pose_mul(pose_mul(pose_inv(shift), next), pose_inv(pos)), shift);
    w.b_linvel[i] = f4(dpos.t * inv_dt, 0.0f);
    w.b_angvel[i] = f4(quat_to_scaled_axis(dpos.r) * inv_dt, 0.0f);
}

// Idle step: while every non-fixed body sleeps and nothing is pending, PhysicsPipeline::step changes nothing but the
// step count (no awake body => no pair is processed, no timer runs, no island can wake).  One tiny kernel verifies
// that on the device and retires the step; otherwise it raises FL_FAST_ABORT and the host replays the step through
// the full graph (same protocol as the steady-state fast path, rp_api.hip).
__global__ void k_idle_step(DevWorld w) {
    if (threadIdx.x == 0) {
        bool idle = !w.flags[FL_FAST_ABORT] && w.flags[FL_N_AWAKE] == 0 && !w.flags[FL_WAKE_PENDING] && !w.flags[FL_BP_DIRTY] &&
                    !w.flags[FL_LAYOUT_DIRTY];
        if (idle) w.flags[FL_STEP] += 1; else w.flags[FL_FAST_ABORT] = 1;
        w.flags[FL_SEQ] += 1;
        w.flags[FL_FULL_UPDATES] = 0;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x < FL_COUNT) {
        int v = __hip_atomic_load(&w.flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&w.host_flags[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
void rp_launch_idle_step(const DevWorld &w, hipStream_t st) { hipLaunchKernelGGL(k_idle_step, dim3(1), dim3(64), 0, st, w); }

static int slp_body_blocks(const DevWorld &w) { int nb = (w.n_bodies + 255) / 256; return nb < 1 ? 1 : nb; }
static int slp_pair_blocks(const DevWorld &w) { int b = (w.pool_cap + 255) / 256; if (b > 2048) b = 2048; return b < 1 ? 1 : b; }

void rp_launch_wake(const DevWorld &w, hipStream_t st, int phase) {
    if (!w.sleep_enabled || w.n_bodies == 0) return;
    hipLaunchKernelGGL(k_wake_spread, dim3(slp_body_blocks(w)), dim3(256), 0, st, w, phase);
    hipLaunchKernelGGL(k_wake_commit, dim3(slp_body_blocks(w)), dim3(256), 0, st, w, phase);
}
void rp_launch_wake_partners(const DevWorld &w, hipStream_t st) {
    if (!w.sleep_enabled || w.n_colliders == 0) return;
    hipLaunchKernelGGL(k_wake_partners, dim3(slp_pair_blocks(w)), dim3(256), 0, st, w);
}
void rp_launch_sleep(const DevWorld &w, hipStream_t st) {
    if (!w.sleep_enabled || w.n_bodies == 0) return;
    int nb = slp_body_blocks(w);
    if (w.has_kinematic_pos) hipLaunchKernelGGL(k_kinematic_velocities, dim3(nb), dim3(256), 0, st, w); // after the narrow phase, before the sleep timers
    // every workgroup must be resident (grid barriers): at most 192 workgroups of 1024 threads
    int blocks = (w.n_bodies + 255) / 256; if (blocks > 192) blocks = 192; if (blocks < 1) blocks = 1; // sized by the bodies (the every-step observation); the pair pass of a relabel is grid-stride
    hipLaunchKernelGGL(k_sleep_pass, dim3(blocks), dim3(1024), 0, st, w, 0);
    hipLaunchKernelGGL(k_sleep_commit, dim3(nb), dim3(256), 0, st, w);
}
// the sleep pass of a FAST step (after k_fast_front): observation + the check that nothing is about to fall asleep
void rp_launch_sleep_fast(const DevWorld &w, hipStream_t st) {
    if (!w.sleep_enabled || w.n_bodies == 0) return;
    int nb = slp_body_blocks(w);
    int blocks = nb > 192 ? 192 : nb;
    hipLaunchKernelGGL(k_sleep_pass, dim3(blocks), dim3(1024), 0, st, w, 1);
    hipLaunchKernelGGL(k_sleep_check, dim3(nb), dim3(256), 0, st, w);
}
