// rp_sleep.hip — sleeping on device: sleep timers, persistent islands, whole-island sleep decision, island-wide wake-up.
//
// Reference: RigidBodyActivation::update_energy (/root/reference/src/dynamics/rigid_body_components.rs:1412-1478),
// the fused active-body pass with its sleep observations and split bids (pipeline/physics_pipeline/solve.rs:159-300),
// IslandManager::update_islands / commit_sleeping_chunks (island_manager/manager.rs:335-388, sleep.rs:81-131),
// IslandManager::wake_up (sleep.rs:31-79) and the persistent islands themselves (island_manager/persistent.rs,
// local_split.rs, global_split.rs).
//
// Persistent islands (sleep-enabled worlds only).  The reference merges islands EAGERLY when a contact starts touching or a
// joint is inserted and splits them LAZILY: an unlinked edge is journaled; at the top of the next solve a bounded local search
// either proves the endpoints still connected (nothing happens), moves the detached — smaller — component out at once, or
// gives up (both endpoints moving fast, budget, sleeping island; a removed body always) and bumps constraint_remove_count,
// which BLOCKS the island's sleep until the deferred global union-find split — one island per step, chosen by the bid of the
// sleepiest eligible body, then SPLIT_RETRY_COOLDOWN scans of rest — has cleared it.  What those decisions read is the
// partition, the body counts, the flag and the cooldown stamp; on MI355X the partition of the CURRENT graph is cheap to
// recompute (a lock-free union-find over the touching pairs), so the device keeps per body its island id (b_isl) and per
// island {in use, bodies, dirty, denied-until, sleeping}, and one sleep pass of a step in which something changed does, behind
// grid barriers: component labels of the awake bodies -> island-level union of the touching pairs (merge groups; the
// largest island of a group survives) -> relabel -> one lane walks the sorted removal journal at COMPONENT granularity (a
// lockstep dual flood from both endpoints ends exactly as the component sizes say: local_split.rs:371-413) -> the pending
// global split -> relabel.  Island ids are handed out like alloc_island does (most recently freed first).  Where the reference
// visits links in contact-graph edge order — an order owned by parry's BVH traversal, not by /root/reference — a canonical rule
// takes its place (DESIGN.md section 4.7 lists them): a merge group keeps the identity of its largest island (smaller id on
// equal size) and frees the absorbed ids in ascending order, the journal is walked in (joint, deleted pair, end-touch; collider
// pair) order, equal-size detaches move body1's side out, a still-connected verdict is never over budget, the global split
// keeps the largest component (smallest body on equal size) and creates the others in ascending order of their smallest body.
//
// Per step (sleep-enabled worlds only):
//   after the broad phase : k_wake_spread(0) + k_wake_commit(0)   user wake-ups, pair deletions (the unlink is journaled by k_bp_rebuild)
//   after the narrow phase: k_wake_spread(1) + k_wake_commit(1)   begin-touch wake-ups (end-touch unlinks journaled by k_np_update)
//   then                  : [k_pi_link_joints] -> k_sleep_pass = island maintenance (only when something changed) + timers, bids,
//                           observation -> k_sleep_commit = scheduled split, scan stamp, the sleep decision behind the gate
// and the bucket / island rebuild that follows sees the new awake set.
#include "rp_pairs.h"
#include "rp_sleep_observe.h"
#include "rp_gridbar.h"

RP_DEV int slp_ld(int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RP_DEV int slp_find(int *label, int x) { // with path halving (a non-root node is re-pointed at its grandparent: still an ancestor)
    int p = slp_ld(&label[x]);
    while (p != x) {
        int gp = slp_ld(&label[p]);
        if (gp != p) atomicCAS(&label[x], p, gp);
        x = p; p = gp;
    }
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
        int isl = w.b_isl[i];
        if (isl >= 0) { w.lab_wake[isl] = cur_step(w); w.flags[FL_WAKE_STAMP] = 2 * cur_step(w) + phase; }
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
    int isl = w.b_isl[i];
    if (isl < 0 || w.lab_wake[isl] != cur_step(w)) return;
    w.pi_sleeping[isl] = 0; // the whole persistent island wakes (sleep.rs:44-70)
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

// ---- persistent islands --------------------------------------------------------------------------------------------------
#define PI_COOLDOWN 16      // SPLIT_RETRY_COOLDOWN, persistent.rs:31
#define PI_SEARCH_BUDGET 1024 // SEARCH_BUDGET, local_split.rs:21
enum { PIS_MERGED = 0, PIS_MULTIWAY, PIS_REMOVALS, PIS_CONNECTED, PIS_DETACHED, PIS_HOT, PIS_OVER_BUDGET, PIS_SLEEPING_DEFERRED, PIS_GLOBAL_SPLITS,
       PIS_GLOBAL_SPLIT_PIECES, PIS_BIDS, PIS_BID_TIES, PIS_SLEEP_BLOCKED, PIS_ORDER_DEPENDENT, PIS_DETACH_SIZE_TIES, PIS_SPLIT_KEEP_TIES };

RP_DEV bool body_member(const DevWorld &w, int b) { return b >= 0 && (w.b_flags[b] & RP_BF_TYPE_MASK) != RP_BODY_FIXED && w.b_isl[b] >= 0; }
// alloc_island (persistent.rs:196-205): the most recently freed id first, else the next unused one.  One lane at a time.
RP_DEV int pi_alloc(const DevWorld &w, int nbodies, int sleeping) {
    int nf = w.flags[FL_PI_NFREE], id;
    if (nf > 0) { id = w.pi_free[nf - 1]; w.flags[FL_PI_NFREE] = nf - 1; } else { id = w.flags[FL_PI_NEXT]; w.flags[FL_PI_NEXT] = id + 1; }
    w.pi_used[id] = 1; w.pi_nb[id] = nbodies; w.pi_dirty[id] = 0; w.pi_denied[id] = 0; w.pi_sleeping[id] = sleeping;
    return id;
}
RP_DEV void pi_free_id(const DevWorld &w, int id) { // free_island (:207-215)
    w.pi_used[id] = 0; w.pi_nb[id] = 0;
    if (w.flags[FL_PI_PENDING] == id + 1) w.flags[FL_PI_PENDING] = 0;
    int nf = w.flags[FL_PI_NFREE];
    w.pi_free[nf] = id; w.flags[FL_PI_NFREE] = nf + 1;
}
// (k_pi_ensure follows block_compact)
// remove_body_raw (:254-288): the island loses a body — and is dirtied, a body can be a cut vertex — the last body frees it
__global__ void k_pi_remove_body(DevWorld w, int b) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int id = w.b_isl[b];
    w.b_isl[b] = -1;
    if (id < 0 || !w.pi_used[id]) return;
    w.pi_nb[id] -= 1; w.pi_dirty[id] = 1;
    if (w.pi_nb[id] == 0) pi_free_id(w, id);
}
// ImpulseJointIslandEvent::Link of the device joints [FL_PI_JLINK - 1, n_joints), in insertion order (substep.rs:357-362) ->
// link_joint -> merge_islands (:361-393, :420-461): pairwise union by size, the island of body1 survives equal sizes, the absorbed id
// is freed at once.  One workgroup: lane 0 walks the joints over an island-level union-find, then every lane relabels bodies.
__global__ void __launch_bounds__(1024) k_pi_link_joints(DevWorld w) {
    const int from = w.flags[FL_PI_JLINK] - 1;
    if (from < 0) return;
    const int n_isl = w.flags[FL_PI_NEXT];
    for (int s = threadIdx.x; s < n_isl; s += blockDim.x) w.pi_uf[s] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int j = from; j < w.n_joints; ++j) {
            int b1 = w.j_b1[j], b2 = w.j_b2[j];
            if (!body_member(w, b1) || !body_member(w, b2)) continue; // a fixed side does not connect
            int a = slp_find(w.pi_uf, w.b_isl[b1]), b = slp_find(w.pi_uf, w.b_isl[b2]);
            if (a == b) continue;
            int big = w.pi_nb[a] >= w.pi_nb[b] ? a : b, small = big == a ? b : a;
            w.pi_uf[small] = big;
            w.pi_nb[big] += w.pi_nb[small]; w.pi_dirty[big] |= w.pi_dirty[small]; w.pi_sleeping[big] &= w.pi_sleeping[small];
            pi_free_id(w, small);
            w.pi_stats[PIS_MERGED] += 1;
        }
        w.flags[FL_PI_JLINK] = 0;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < w.n_bodies; b += blockDim.x) { int id = w.b_isl[b]; if (id >= 0) w.b_isl[b] = slp_find(w.pi_uf, id); }
}

// Ascending compaction by one workgroup: out[base + rank] = i for every i in [0, n) with pred(i); returns the number kept.
template <typename P> RP_DEV int block_compact(int n, int *out, int base, int *lds /* [17] */, P pred) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int total = 0;
    for (int c0 = 0; c0 < n; c0 += blockDim.x) {
        int i = c0 + threadIdx.x;
        bool keep = i < n && pred(i);
        unsigned long long m = __ballot(keep);
        if (lane == 0) lds[wave] = __popcll(m);
        __syncthreads();
        int off = 0, sum = 0;
        for (int k = 0; k < nw; ++k) { int c = lds[k]; if (k < wave) off += c; sum += c; }
        if (keep) out[base + total + off + __popcll(m & ((1ull << lane) - 1ull))] = i;
        total += sum;
        __syncthreads();
    }
    return total;
}

// ensure_body (:217-232) for the bodies [first, first + count), in index order: run by the host after it appended body rows (one
// workgroup).  The k-th new non-fixed body takes the k-th id alloc_island would hand out: the free stack from its top, then fresh ids.
// `reset`: bootstrap (persistent.rs:600-625) — every island is dropped first, all joints are linked again by the next sleep pass.
__global__ void __launch_bounds__(1024) k_pi_ensure(DevWorld w, int first, int count, int reset) {
    __shared__ int lds[32];
    if (reset) {
        for (int i = threadIdx.x; i < w.n_bodies; i += blockDim.x) w.b_isl[i] = -1;
        if (threadIdx.x == 0) { w.flags[FL_PI_NEXT] = 0; w.flags[FL_PI_NFREE] = 0; w.flags[FL_PI_PENDING] = 0; w.flags[FL_PJ_COUNT] = 0; w.flags[FL_PI_JLINK] = w.n_joints > 0 ? 1 : 0; w.flags[FL_LAYOUT_DIRTY] = 1; }
        __syncthreads();
    }
    for (int i = first + threadIdx.x; i < first + count; i += blockDim.x) w.b_isl[i] = -1;
    __syncthreads();
    const int nf = w.flags[FL_PI_NFREE], next = w.flags[FL_PI_NEXT];
    const int total = block_compact(count, w.pi_list, 0, lds, [&](int k) { return (w.b_flags[first + k] & RP_BF_TYPE_MASK) != RP_BODY_FIXED; });
    for (int k = threadIdx.x; k < total; k += blockDim.x) {
        const int b = first + w.pi_list[k], id = k < nf ? w.pi_free[nf - 1 - k] : next + (k - nf);
        w.b_isl[b] = id;
        w.pi_used[id] = 1; w.pi_nb[id] = 1; w.pi_dirty[id] = 0; w.pi_denied[id] = 0; w.pi_sleeping[id] = (w.b_flags[b] & RP_BF_SLEEPING) ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) { w.flags[FL_PI_NFREE] = total < nf ? nf - total : 0; w.flags[FL_PI_NEXT] = next + (total > nf ? total - nf : 0); }
}
// ImpulseJointIslandEvent::Unlink -> unlink_joint -> journal_removal (persistent.rs:395-418): only a joint whose Link was applied
// is linked (joint_link_locs), so only it is journaled
__global__ void k_pj_append_joint(DevWorld w, int dev_joint, int b1, int b2, int key) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int from = w.flags[FL_PI_JLINK] - 1;
    if (from >= 0 && dev_joint >= from) return; // never linked
    if (b1 == b2 || b1 < 0 || b2 < 0) return;
    int k = atomicAdd(&w.flags[FL_PJ_COUNT], 1);
    if (k < w.pj_cap) { w.pj_key[k] = (unsigned long long)(unsigned)key; w.pj_b[k] = make_int2(b1, b2); }
}

// resolve_removals (local_split.rs:164-255), lane 0 of workgroup 0, over the journal sorted by (phase, key).  The lockstep dual
// search of :371-413 ends as the component sizes say: same component = the floods meet; otherwise side 0 (body1's) runs dry
// after |C0| expansions of either side when |C0| <= |C1|, side 1 after |C1| + 1 and |C1| otherwise.
RP_DEV void pi_resolve_serial(const DevWorld &w, int n) {
    int last_isl = -1, last_cnt = 0;
    for (int k = 0; k < n; ++k) {
        int2 e = w.pj_b[k];
        w.pi_stats[PIS_REMOVALS] += 1;
        if (!body_member(w, e.x) || !body_member(w, e.y)) continue;      // :186-196 a fixed / missing side carried no connectivity
        const int f1 = w.b_flags[e.x], f2 = w.b_flags[e.y];
        const int c1 = w.b_slabel[e.x], c2 = w.b_slabel[e.y];
        const int i1 = (f1 & RP_BF_SLEEPING) ? w.b_isl[e.x] : w.pi_cisl[c1], i2 = (f2 & RP_BF_SLEEPING) ? w.b_isl[e.y] : w.pi_cisl[c2];
        if (i1 != i2) continue;                                           // :198-202 an earlier removal of the batch separated them
        if (w.pi_sleeping[i1]) { w.pi_dirty[i1] = 1; w.pi_stats[PIS_SLEEPING_DEFERRED] += 1; continue; } // :207-210
        bool hot = true;                                                  // :212-231
        for (int q = 0; q < 2; ++q) {
            int b = q ? e.y : e.x;
            float lin_threshold = w.b_sleep[b].y * w.prm.p.length_unit;
            if (lin_threshold < 0.0f) continue;
            V3 lv = v3(w.b_linvel[b]), av = v3(w.b_angvel[b]);
            float max_point_vel = sqrtf(dot(lv, lv)) + sqrtf(dot(av, av)) * w.b_sprev_t[b].w;
            if (!(max_point_vel > lin_threshold)) hot = false;
        }
        if (hot) { w.pi_dirty[i1] = 1; w.pi_stats[PIS_HOT] += 1; continue; }
        if (c1 == c2) { w.pi_stats[PIS_CONNECTED] += 1; continue; }
        const int s1 = w.pi_csize[c1], s2 = w.pi_csize[c2];
        const int side = s1 <= s2 ? 0 : 1, expansions = side == 0 ? 2 * s1 : 2 * s2 + 1;
        if (expansions >= PI_SEARCH_BUDGET) { w.pi_dirty[i1] = 1; w.pi_stats[PIS_OVER_BUDGET] += 1; continue; }
        if (s1 == s2) w.pi_stats[PIS_DETACH_SIZE_TIES] += 1;
        const int c = side == 0 ? c1 : c2, sz = side == 0 ? s1 : s2;     // move_component_out (:260-345)
        int id = pi_alloc(w, sz, 0);
        w.pi_nb[i1] -= sz;
        w.pi_cisl[c] = id;
        w.pi_stats[PIS_DETACHED] += 1;
        // (statistics only) two detaching removals on one island in one step: the journal order can matter.  Exact when they are
        // adjacent in the journal, which is all the tests need; the oracle counts every case.
        if (i1 == last_isl) { if (++last_cnt == 2) w.pi_stats[PIS_ORDER_DEPENDENT] += 1; } else { last_isl = i1; last_cnt = 1; }
    }
}
// run_pending_split -> split_island_now (global_split.rs:44-308) by workgroup 0: the largest component keeps the base island (the one
// with the smallest body on equal size), the others become islands in ascending order of their smallest body.
RP_DEV void pi_split_pending(const DevWorld &w, int *lds, unsigned long long *lds64) {
    const int id = w.flags[FL_PI_PENDING] - 1;
    __syncthreads();
    if (threadIdx.x == 0) w.flags[FL_PI_PENDING] = 0;                    // take()
    if (id < 0 || !w.pi_used[id] || w.pi_sleeping[id]) return;           // :46-53 (uniform over the workgroup)
    if (threadIdx.x == 0) { w.pi_stats[PIS_GLOBAL_SPLITS] += 1; lds64[0] = 0ull; lds[16] = 0; lds64[1] = 0ull; }
    __syncthreads();
    int ncomp = 0;
    if (w.pi_nb[id] > 1) {
        // roots of the island's components: their smallest bodies (awake island: every member is awake)
        unsigned long long best = 0ull; int cnt = 0, ties = 0;
        for (int b = threadIdx.x; b < w.n_bodies; b += blockDim.x) {
            if (!flags_active(w.b_flags[b]) || w.b_slabel[b] != b || w.pi_cisl[b] != id) continue;
            unsigned long long v = ((unsigned long long)(unsigned)w.pi_csize[b] << 32) | (unsigned)~(unsigned)b;
            if (v > best) best = v;
            cnt++;
        }
        atomicMax(&lds64[0], best);
        atomicAdd(&lds[16], cnt);
        __syncthreads();
        best = lds64[0]; ncomp = lds[16];
        const int keep = (int)~(unsigned)(best & 0xffffffffull), keep_size = (int)(best >> 32);
        if (ncomp > 1) {
            int nlist = block_compact(w.n_bodies, w.pi_list, 0, lds, [&](int b) {
                return flags_active(w.b_flags[b]) && w.b_slabel[b] == b && w.pi_cisl[b] == id && b != keep; });
            for (int b = threadIdx.x; b < w.n_bodies; b += blockDim.x)
                if (flags_active(w.b_flags[b]) && w.b_slabel[b] == b && w.pi_cisl[b] == id && b != keep && w.pi_csize[b] == keep_size) ties = 1;
            if (ties) lds[16] = -1;
            __syncthreads();
            if (threadIdx.x == 0) {
                if (lds[16] < 0) w.pi_stats[PIS_SPLIT_KEEP_TIES] += 1;
                for (int k = 0; k < nlist; ++k) { int c = w.pi_list[k]; w.pi_cisl[c] = pi_alloc(w, w.pi_csize[c], 0); }
                w.pi_stats[PIS_GLOBAL_SPLIT_PIECES] += nlist;
                w.pi_nb[id] = keep_size;
            }
        }
    }
    if (threadIdx.x == 0) { w.pi_dirty[id] = 0; w.pi_denied[id] = pi_stamp_before(w) + PI_COOLDOWN; } // :69-74, :156-162, :302-305
    __syncthreads();
}
// removal journal sorted by key (phase << 62 | pair key): in-place bitonic sort of the padded arrays by workgroup 0
RP_DEV void pi_sort_journal(const DevWorld &w, int n) {
    int m = 1; while (m < n) m <<= 1;
    for (int i = n + threadIdx.x; i < m; i += blockDim.x) { w.pj_key[i] = ~0ull; w.pj_b[i] = make_int2(-1, -1); }
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < m; i += blockDim.x) {
                int l = i ^ j;
                if (l > i) {
                    unsigned long long a = w.pj_key[i], b = w.pj_key[l];
                    bool up = (i & k) == 0;
                    if ((a > b) == up) { w.pj_key[i] = b; w.pj_key[l] = a; int2 t = w.pj_b[i]; w.pj_b[i] = w.pj_b[l]; w.pj_b[l] = t; }
                }
            }
            __syncthreads();
        }
}

// finish_sleep_scan (persistent.rs:498-516): does island `isl` stay awake this step?  Some member was not eligible, or it lost
// constraints and holds more than one body (it must split first)
RP_DEV bool island_stays_awake(const DevWorld &w, int isl) {
    return w.lab_awake[isl] == cur_step(w) || (w.pi_dirty[isl] && w.pi_nb[isl] > 1);
}
// commit_sleeping_chunks -> RigidBody::sleep (rigid_body.rs:804-807) + clear_asleep_pair_solver_hint_counts_of
RP_DEV void sleep_commit(DevWorld &w, int gid, int gstride) {
    for (int base = 0; base < w.n_bodies; base += gstride) { // wave-uniform trip count: the ballot below needs whole wavefronts
        const int i = base + gid;
        int fl = i < w.n_bodies ? w.b_flags[i] : RP_BODY_FIXED;
        bool active = flags_active(fl);
        const int isl = active ? w.b_isl[i] : -1;
        bool stays_awake = active && (isl < 0 || island_stays_awake(w, isl));
        // awake bodies left after this pass, one atomic per wavefront (0 = the whole world sleeps: the host may enqueue idle steps)
        unsigned long long awake_mask = __ballot(stays_awake);
        if ((threadIdx.x & 63) == 0 && awake_mask) atomicAdd(&w.flags[FL_N_AWAKE], __popcll(awake_mask));
        if (!active || stays_awake) continue;
        w.b_flags[i] = fl | RP_BF_SLEEPING;
        float4 sl = w.b_sleep[i]; sl.x = sl.w; w.b_sleep[i] = sl;
        w.b_linvel[i] = make_float4(0, 0, 0, 0); w.b_angvel[i] = make_float4(0, 0, 0, 0);
        w.b_slept_at[i] = cur_step(w);
        w.pi_sleeping[isl] = 1; // mark_island_sleeping (:520-523)
        w.flags[FL_LAYOUT_DIRTY] = 1;
        if (w.n_joints) w.flags[FL_JOINT_DIRTY] = 1;
    }
}
// The sleep pass of a step in TWO launches: (1) the island maintenance when something changed (passes behind grid barriers,
// rp_gridbar.h — skipped on a clean step) and the per-body observation; (2) the commit, which may only run once EVERY member of an
// island was observed: that dependency is a kernel boundary, cheaper on MI355X than a fenced grid barrier that would have to run
// every step (~4 us against ~7 us).
__global__ void __launch_bounds__(1024) k_sleep_pass(DevWorld w, int fast) {
    if (fast && w.flags[FL_FAST_ABORT]) return; // the fast graph gave up on this step: nothing may change
    __shared__ int lds[32];
    __shared__ unsigned long long lds64[2];
    const int gid = gbar_item(), gstride = gridDim.x * blockDim.x;
    // (every workgroup reads these before its first barrier; workgroup 0 changes them only after several)
    const bool work = !fast && (w.flags[FL_LAYOUT_DIRTY] || w.flags[FL_PJ_COUNT] > 0 || w.flags[FL_PI_PENDING] > 0);
    if (work) {
        GridBar bar = gbar_begin(w, 2);
        const int n_isl = w.flags[FL_PI_NEXT], n_journal = w.flags[FL_PJ_COUNT] < w.pj_cap ? w.flags[FL_PJ_COUNT] : w.pj_cap;
        // P0: component labels start as singletons; island-level union-find and merge scratch reset
        slp_init(w, gid, gstride);
        for (int s = gid; s < n_isl; s += gstride) { w.pi_uf[s] = s; w.pi_best[s] = 0ull; w.pi_new[s] = w.pi_used[s] ? s : -1; }
        for (int i = gid; i < w.n_bodies; i += gstride) w.pi_csize[i] = 0;
        if (gid == 0) w.flags[FL_PI_MERGED] = 0;
        GBAR_SYNC(bar);
        // P1: components of the awake bodies over touching pairs and joints; link_contact (persistent.rs:293-329): a touching pair
        // whose endpoints sit in different islands joins them (already-linked pairs join nothing)
        slp_union_pairs(w, gid, gstride);
        {
            int top = w.flags[FL_POOL_TOP];
            if (top > w.pool_cap) top = w.pool_cap;
            for (int s = gid; s < top; s += gstride) {
                if (w.p_c1[s] < 0 || w.p_nsc[s] == 0) continue;
                int2 rb = w.p_rb[s];
                if (!body_member(w, rb.x) || !body_member(w, rb.y)) continue;
                int i1 = w.b_isl[rb.x], i2 = w.b_isl[rb.y];
                if (i1 != i2) { slp_union(w.pi_uf, i1, i2); w.flags[FL_PI_MERGED] = 1; }
            }
        }
        GBAR_SYNC(bar);
        const bool merged = w.flags[FL_PI_MERGED] != 0;
        // P2: final component labels; merge_islands (:420-461): the identity that survives a group is its largest island as of
        // the start of the step, the smaller id on equal size
        slp_flatten(w, gid, gstride);
        if (merged)
            for (int s = gid; s < n_isl; s += gstride)
                if (w.pi_used[s]) atomicMax(&w.pi_best[slp_find(w.pi_uf, s)], ((unsigned long long)(unsigned)w.pi_nb[s] << 32) | (unsigned)~(unsigned)s);
        GBAR_SYNC(bar);
        // P3: absorbed islands hand their bodies, flag and sleep state to the survivor
        if (merged)
            for (int s = gid; s < n_isl; s += gstride) {
                if (!w.pi_used[s]) continue;
                const int r = slp_find(w.pi_uf, s), win = (int)~(unsigned)(w.pi_best[r] & 0xffffffffull);
                w.pi_new[s] = win;
                if (win == s) continue;
                atomicAdd(&w.pi_nb[win], w.pi_nb[s]);
                if (w.pi_dirty[s]) atomicOr(&w.pi_dirty[win], 1);
                if (!w.pi_sleeping[s]) atomicAnd(&w.pi_sleeping[win], 0);
                if (w.flags[FL_PI_PENDING] == s + 1) w.flags[FL_PI_PENDING] = 0; // free_island drops a pending split of the absorbed island
                atomicAdd(&w.pi_stats[PIS_MERGED], 1);
            }
        GBAR_SYNC(bar);
        // P4: bodies follow; component sizes and the island of every component; absorbed ids are freed in ascending order
        for (int i = gid; i < w.n_bodies; i += gstride) {
            int isl = w.b_isl[i];
            if (isl < 0) continue;
            if (merged) { isl = w.pi_new[isl]; w.b_isl[i] = isl; }
            if (flags_active(w.b_flags[i])) { int c = w.b_slabel[i]; atomicAdd(&w.pi_csize[c], 1); if (c == i) w.pi_cisl[i] = isl; }
        }
        if (merged && blockIdx.x == 0) {
            int nf = w.flags[FL_PI_NFREE];
            int freed = block_compact(n_isl, w.pi_free, nf, lds, [&](int s) { return w.pi_new[s] >= 0 && w.pi_new[s] != s; });
            for (int k = threadIdx.x; k < freed; k += blockDim.x) { int s = w.pi_free[nf + k]; w.pi_used[s] = 0; w.pi_nb[s] = 0; }
            if (threadIdx.x == 0) w.flags[FL_PI_NFREE] = nf + freed;
        }
        GBAR_SYNC(bar);
        // P5 (workgroup 0): resolve_removals over the sorted journal, then the pending global split
        if (blockIdx.x == 0) {
            if (n_journal > 0) {
                pi_sort_journal(w, n_journal);
                if (threadIdx.x == 0) { pi_resolve_serial(w, n_journal); w.flags[FL_PJ_COUNT] = 0; }
                __syncthreads();
            }
            pi_split_pending(w, lds, lds64);
        }
        GBAR_SYNC(bar);
        // P6: awake bodies follow their component's island
        for (int i = gid; i < w.n_bodies; i += gstride) if (flags_active(w.b_flags[i]) && w.b_isl[i] >= 0) w.b_isl[i] = w.pi_cisl[w.b_slabel[i]];
        gbar_end(bar);
    }
    const int stamp_before = pi_stamp_before(w);
    for (int i = gid; i < w.n_bodies; i += gstride) sleep_observe_one(w, i, stamp_before);
}
// The bid of the step becomes next step's pending split (schedule_split, solve.rs:293-295) unless its island falls asleep in this very
// commit (mark_island_sleeping -> clear_pending_split_of); then the sleep decision.  (The bid word is reset for the next step.)
__global__ void k_sleep_commit(DevWorld w) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long bid = w.pi_w64[1];
        if (bid) {
            int isl = (int)(unsigned)(bid & 0xffffffffull);
            w.flags[FL_PI_PENDING] = island_stays_awake(w, isl) ? isl + 1 : 0;
            w.pi_w64[1] = 0ull;
            w.pi_stats[PIS_BIDS] += 1;
        }
    }
    sleep_commit(w, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
// fast graph: would the commit put an island to sleep (no member marked it awake this step, no gate)?  Did an island bid for a split?
// Then the step needs the full graph: abort before anything but the — once-per-step — observation has happened.
__global__ void k_sleep_check(DevWorld w) {
    if (w.flags[FL_FAST_ABORT]) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && w.pi_w64[1] != 0ull) w.flags[FL_FAST_ABORT] = 1;
    if (i >= w.n_bodies || !flags_active(w.b_flags[i])) return;
    int isl = w.b_isl[i];
    if (isl >= 0 && !island_stays_awake(w, isl)) w.flags[FL_FAST_ABORT] = 1;
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
    for (int k = threadIdx.x; k < FL_COUNT; k += blockDim.x) {
        int v = __hip_atomic_load(&w.flags[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&w.host_flags[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
    // every workgroup must be resident (grid barriers): at most DevWorld::gbar_blocks workgroups of 1024 threads
    int blocks = (w.n_bodies + 255) / 256; if (blocks > w.gbar_blocks) blocks = w.gbar_blocks; if (blocks < 1) blocks = 1; // sized by the bodies (the every-step observation); the pair pass of a relabel is grid-stride
    if (w.n_joints > 0) hipLaunchKernelGGL(k_pi_link_joints, dim3(1), dim3(1024), 0, st, w); // joint Link events queued by the host (early exit otherwise)
    hipLaunchKernelGGL(k_sleep_pass, dim3(blocks), dim3(1024), 0, st, w, 0);
    hipLaunchKernelGGL(k_sleep_commit, dim3(nb), dim3(256), 0, st, w);
}
// the sleep pass of a FAST step (after k_fast_front): observation + the check that nothing is about to fall asleep
void rp_launch_sleep_fast(const DevWorld &w, hipStream_t st) {
    if (!w.sleep_enabled || w.n_bodies == 0) return;
    int nb = slp_body_blocks(w);
    int blocks = nb > w.gbar_blocks ? w.gbar_blocks : nb;
    hipLaunchKernelGGL(k_sleep_pass, dim3(blocks), dim3(1024), 0, st, w, 1);
    hipLaunchKernelGGL(k_sleep_check, dim3(nb), dim3(256), 0, st, w);
}

// host-triggered island bookkeeping (rp_api.hip): new body rows, a removed body
void rp_launch_pi_ensure(const DevWorld &w, hipStream_t st, int first, int count, int reset) { if (count > 0 || reset) hipLaunchKernelGGL(k_pi_ensure, dim3(1), dim3(1024), 0, st, w, first, count, reset); }
void rp_launch_pj_append_joint(const DevWorld &w, hipStream_t st, int dev_joint, int b1, int b2, int key) { hipLaunchKernelGGL(k_pj_append_joint, dim3(1), dim3(64), 0, st, w, dev_joint, b1, b2, key); }
void rp_launch_pi_remove_body(const DevWorld &w, hipStream_t st, int b) { hipLaunchKernelGGL(k_pi_remove_body, dim3(1), dim3(64), 0, st, w, b); }

// workgroups of k_sleep_pass (1024 threads) one CU holds at once (0 = the query failed): input of DevWorld::gbar_blocks (rp_api.hip)
int rp_occ_sleep_pass(void) { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_sleep_pass, 1024, 0) != hipSuccess) n = 0; return n; }
