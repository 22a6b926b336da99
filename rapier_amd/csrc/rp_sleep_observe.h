// rp_sleep_observe.h — the per-body half of the sleep pass (update_body_energy + the island marks and split bids), shared by the sleep
// kernels (rp_sleep.hip) and by the validators of the fused island step (rp_islands.hip / rp_islands_lean.h), which run it for the
// bodies of their islands so that a sleep-enabled world whose bodies are all awake keeps the single-kernel step.
#pragma once
#include "rp_pairs.h"

// sleep_scan_stamp as the bids and the split of step `cur_step` must see it: the value BEFORE this step's begin_sleep_scan, whether
// or not an (aborted, replayed) pass of the same step number already bumped it
RP_DEV int pi_stamp_before(const DevWorld &w) {
    unsigned long long v = __hip_atomic_load(&w.pi_w64[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int stamp = (int)(unsigned)(v & 0xffffffffull);
    return (int)(v >> 32) == cur_step(w) ? stamp - 1 : stamp;
}
// update_body_energy for every awake body, its split bid and the island observation (an island sleeps once EVERY member is eligible).
// The timer update runs at most once per step NUMBER (b_sleep_stamp): a fast step that observes, then aborts (an island is about to
// fall asleep, k_sleep_check) is replayed on the full graph with the same step number, finds every timer already advanced and only
// repeats the (idempotent) island marks and bids.
RP_DEV void sleep_mark(const DevWorld &w, int i, float4 sl, int stamp_before) {
    const int isl = w.b_isl[i];
    if (isl < 0) return;
    if (!(sl.x >= sl.w)) { w.lab_awake[isl] = cur_step(w); return; }
    // solve.rs:225-237: an eligible body whose island lost constraints and is out of its cooldown (split_allowed, persistent.rs:181-186)
    // bids its stillness; max (score, island id) wins (:206-211)
    if (w.pi_dirty[isl] && stamp_before >= w.pi_denied[isl])
        atomicMax(&w.pi_w64[1], ((unsigned long long)(unsigned)__float_as_int(sl.x) << 32) | (unsigned)isl);
}
// begin_sleep_scan (persistent.rs:463-473): the first observation of a step bumps the stamp
RP_DEV void sleep_begin_scan(const DevWorld &w) {
    unsigned long long v = __hip_atomic_load(&w.pi_w64[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(v >> 32) != cur_step(w)) atomicCAS(&w.pi_w64[0], v, ((unsigned long long)(unsigned)cur_step(w) << 32) | (unsigned)((unsigned)(v & 0xffffffffull) + 1u));
}
// (returns the body's timer row as it stands after the observation; BEGUN: the caller has bumped the stamp for everyone it observes)
template <bool BEGUN = false> RP_DEV float4 sleep_observe_one(const DevWorld &w, int i, int stamp_before) {
    if (!flags_active(w.b_flags[i])) return make_float4(0, 0, 0, 0);
    if constexpr (!BEGUN) sleep_begin_scan(w);
    float4 sl = w.b_sleep[i];
    if (w.b_sleep_stamp[i] == cur_step(w)) { sleep_mark(w, i, sl, stamp_before); return sl; }
    w.b_sleep_stamp[i] = cur_step(w);
    if ((w.b_flags[i] & RP_BF_TYPE_MASK) != RP_BODY_DYNAMIC) { // platforms only sleep when both velocities are exactly zero (:1464-1468)
        V3 lv = v3(w.b_linvel[i]), kav = v3(w.b_angvel[i]);
        bool still = dot(lv, lv) == 0.0f && dot(kav, kav) == 0.0f;
        sl.x = still ? sl.x + w.prm.p.dt : 0.0f;
        w.b_sleep[i] = sl;
        sleep_mark(w, i, sl, stamp_before);
        return sl;
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
    sleep_mark(w, i, sl, stamp_before);
    return sl;
}
// The fused step's view of one island body: observe it (at most once per step number, like every other caller) and report what the
// step's k_sleep_check would conclude from it.  bit 0: the body belongs to a persistent island; bit 1: it keeps that island awake
// (not eligible yet); bit 2: it bid for a split (the step then belongs to the full graph).  An LDS island is contact-connected, hence
// inside ONE persistent island: when none of its members keeps that island awake the commit might put it to sleep, and the workgroup
// aborts the fused step (conservative: a member of the same persistent island in another workgroup could still keep it awake; the
// replay on the full graph decides exactly).
RP_DEV int sleep_observe_fused(const DevWorld &w, int i, int stamp_before) {
    if (!flags_active(w.b_flags[i])) return 0;
    const float4 sl = sleep_observe_one<true>(w, i, stamp_before); // (the workgroup's first lane began the scan: sleep_begin_scan)
    const int isl = w.b_isl[i];
    if (isl < 0) return 0;
    if (!(sl.x >= sl.w)) return 1 | 2;
    return 1 | ((w.pi_dirty[isl] && stamp_before >= w.pi_denied[isl]) ? 4 : 0);
}
