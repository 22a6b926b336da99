// rp_broadphase.hip — broad phase on device.
//
// Contract: the pair SET of the reference's fat-AABB BVH
// (/root/reference/src/geometry/broad_phase_bvh/mod.rs:171-263, update.rs:35-602):
//   * every collider keeps an AABB fattened by CHANGE_DETECTION_FACTOR (0.04) that is rewritten only
//     when its tight collision AABB (shape AABB loosened by prediction/2, collider.rs:553-557) leaves it;
//   * a pair exists exactly while the two fat AABBs intersect and it passes the filters of
//     update.rs:334-396 (same parent, collision types, interaction groups);
//   * AddPair creates an empty ContactPair, DeletePair frees its solver colour (pair_management.rs:382).
// The tree is free to differ: here a hashed uniform grid of fixed-slot buckets (HBM-bound integer
// work, one thread per collider, wave-coalesced SoA loads) plus a brute-force list for colliders
// spanning more than 3 cells (and for the rare collider that met a full bucket).  Like the reference's change detection the whole pass is skipped
// (every kernel early-exits on FL_BP_DIRTY == 0) while no fat AABB changed, and — like its refit of the changed
// leaves only (update.rs:139-331, :448-601) — a pass in which fat AABBs changed touches only those colliders: the grid in service
// follows every collider whose fat AABB moves to other cells (bp_grid_follow, round 4), so a changed collider finds its new partners
// by walking the cells under its new AABB and the large list, and the pairs it lost by one sweep over the pair slots; deleted pairs
// leave a tombstone in the live hash table.  The pair SET equals the full rebuild's; a full rebuild runs when more than half of the
// colliders changed, a bucket filled up, the tombstones grew, a large collider moved or the topology changed (FL_BP_GRID_OK).
#include "rp_pairs.h"
#include "rp_gridbar.h"
#include "rp_grid.h"

RP_DEV void bp_grid_follow(const DevWorld &w, int i, float4 omn, float4 omx) {
    if (!w.bp_incremental || !w.flags[FL_BP_GRID_OK]) return; // no grid in service / a full rebuild is due anyway
    if (w.c_inlarge[i]) { w.flags[FL_BP_FORCE_FULL] = 1; return; } // on the large list: only a rebuild WITH its build pass renews that list
    const CellRange o = cell_range_of(w, omn, omx), r = cell_range(w, i);
    if (r.large) { w.flags[FL_BP_FORCE_FULL] = 1; return; }        // (likewise)
    const bool valid_before = omn.x <= omx.x && omn.y <= omx.y && omn.z <= omx.z; // (a freshly inserted collider has an empty box and no entry)
    if (valid_before && !o.large && o.lo[0] == r.lo[0] && o.lo[1] == r.lo[1] && o.lo[2] == r.lo[2] && o.hi[0] == r.hi[0] && o.hi[1] == r.hi[1] && o.hi[2] == r.hi[2]) return;
    const int ver = w.c_rver[i] + 1;
    w.c_rver[i] = ver;
    if (ver >= 8) { w.flags[FL_BP_GRID_OK] = 0; return; }      // the 3 version bits would wrap onto entries still in the buckets
    const int cur = BP_GPAR(w);
    int *cnt = w.bk_cnt[cur]; int *items = w.bk_items[cur];
    int ord = 0;
    for (int z = r.lo[2]; z <= r.hi[2]; ++z)
        for (int y = r.lo[1]; y <= r.hi[1]; ++y)
            for (int x = r.lo[0]; x <= r.hi[0]; ++x, ++ord) {
                const int h = (int)(rp_hash64(cell_key_of(w, i, x, y, z)) & (unsigned long long)(w.grid_cap - 1));
                const int k = atomicAdd(&cnt[h], 1);
                if (k < RP_BP_BUCKET) items[(size_t)h * RP_BP_BUCKET + k] = bp_entry(i, ord, ver);
                else w.flags[FL_BP_GRID_OK] = 0;               // bucket full: rebuild
            }
}

__global__ void k_collider_update(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    // (FL_FAST_ABORT == 2: a lean step died behind its collision stage — rp_world.h "lean step graphs" — and this is either its resume
    // on the full graph or a lean graph enqueued before the host noticed: the stage already ran for this step, nothing is reset)
    if (collision_done(w)) return;
    if (i == 0) { // a full step is starting: the fast path may be tried again later; per-step narrow-phase counters
        if (!w.lean) w.flags[FL_FAST_ABORT] = 0;
        w.flags[FL_FULL_UPDATES] = 0; w.flags[FL_TODO_COUNT] = 0; w.flags[FL_NP_COUNT] = 0; w.flags[FL_CCD_N] = 0;
    }
    if (i >= w.n_colliders) return;
    const float4 omn = w.c_fatmin[i], omx = w.c_fatmax[i];
    if (collider_update_one(w, i)) bp_grid_follow(w, i, omn, omx);
}

// Steady-state fast path, first kernel (see rp_api.hip "fast graph"): collider poses + fat-AABB checks
// for every collider AND the contact-recycling test of every pair (pair_update.rs:111-171) computed
// straight from the body poses.  If any fat AABB changed (broad phase must run) or any pair fails its
// recycle test (narrow phase must run) the step cannot be done by the fast graph: FL_FAST_ABORT is
// raised, nothing else has been modified that the full path would not recompute identically, and the
// remaining fast kernels exit; the host replays the step through the full graph.
__global__ void k_fast_front(DevWorld w, int no_global_kernel) {
    if (w.flags[FL_FAST_ABORT]) return;
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid == 0) {
        w.flags[FL_FULL_UPDATES] = 0; w.flags[FL_CCD_N] = 0;
        if (w.flags[FL_BP_DIRTY]) w.flags[FL_FAST_ABORT] = 1;
        // sleep-enabled worlds: the awake set must be settled — no layout change or wake-up request waiting for a full step
        if (w.sleep_enabled && (w.flags[FL_LAYOUT_DIRTY] || w.flags[FL_WAKE_PENDING] || w.flags[FL_N_AWAKE] == 0 || w.flags[FL_PI_PENDING] || w.flags[FL_PJ_COUNT] || w.flags[FL_PI_JLINK])) w.flags[FL_FAST_ABORT] = 1; // (a pending island split, journaled removals, joint links: the full graph's sleep pass)
        // this graph variant carries no global-path kernel: it is only valid while everything lives in LDS islands
        if (no_global_kernel && (w.flags[FL_N_CONS] > 0 || w.flags[FL_N_GLOB_BODIES] > 0 || w.n_joints > 0)) w.flags[FL_FAST_ABORT] = 1;
    }
    bool abort = false;
    // the continuous-collision pass only exists on the full graph: a dynamic body that may move more than a quarter of its thinnest
    // extent this step (velocity + one step of gravity, rotation about the farthest point: twice the margin of the activation
    // criterion, rigid_body_components.rs:1131-1157) sends the step there
    if (w.prm.p.max_ccd_substeps != 0)
        for (int b = gid; b < w.n_bodies; b += gridDim.x * blockDim.x) {
            if (!flags_dyn_awake(w.b_flags[b])) continue;
            const float thickness = w.b_damp[b].w;
            if (!(thickness < 3.0e38f)) continue;
            const float dt = w.prm.p.dt;
            const V3 lv = v3(w.b_linvel[b]), av = v3(w.b_angvel[b]), g = v3(w.prm.gravity[0], w.prm.gravity[1], w.prm.gravity[2]);
            const float motion = (len(lv) + len(g) * dt + len(av) * w.b_invpi[b].w) * dt;
            if (motion > 0.25f * thickness) abort = true;
        }
    if (gid < w.n_colliders) {
        const float4 omn = w.c_fatmin[gid], omx = w.c_fatmax[gid];
        if (collider_update_one(w, gid)) { abort = true; bp_grid_follow(w, gid, omn, omx); } // rewritten fat AABB => FL_BP_DIRTY (the replay on the full graph finds it rewritten: the grid follows here)
    }
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    int stride = gridDim.x * blockDim.x;
    for (int s = gid; s < top; s += stride) {
        if (w.has_composite && w.p_c1[s] >= 0 && pair_is_aux(w, s)) continue; // (a cluster of a composite pair: its parent slot is the pair)
        if (w.sleep_enabled) {
            if (w.p_c1[s] < 0) continue;
            int2 rb = w.p_rb[s];
            if (!body_active(w, rb.x) && !body_active(w, rb.y)) continue;   // pair_update.rs:98-106: neither body awake -> skipped
            if (pair_hint_cleared(w, s, rb)) abort = true;                  // a count-cleared hint is recomputed by the narrow phase
        }
        if (pair_needs_narrow_phase(w, s)) abort = true;
    }
    if (abort) w.flags[FL_FAST_ABORT] = 1;
}

// ---- the rebuild, pass by pass (k_bp_rebuild below runs them behind grid barriers; gid / gstride span the whole launch) ----
// The grid: grid_cap hash buckets of RP_BP_BUCKET fixed slots, two copies.  A slot is ONE word — the collider and which of its (at
// most 27) cells the entry stands for (bp_entry): 32 slots = one 128-byte line.  Cells that share a bucket are told apart without a
// stored key: a reader at cell X accepts an entry only if the cell it stands for — recomputed from the partner's fat AABB, which the
// overlap test loads anyway — is X (bp_entry_is_cell).  A rebuild fills the copy that is out
// of service straight away — one atomic per (collider, cell) hands out the slot, no counting sort: three passes (build | pairs |
// finish) where the counting sort took five (count | scan | add + fill | pairs | finish), and a pass of a rebuild is a grid barrier
// plus a chain of cold misses whatever its work — ~10 us each on MI355X.  The copy in service (epoch parity, like the pair hash
// tables) keeps serving until the rebuild closes; its counters are emptied by the finish pass, so the next rebuild finds its target at
// rest (allocation provides the first rest state).  A collider that finds a bucket full joins the large list for this rebuild
// (c_inlarge): everybody tests against that list anyway, its bucket entries are skipped — slower, never wrong.
#define BP_LARGE_SCRATCH 1024 // scan_block[1024]: large colliders counted by the running rebuild (FL_N_LARGE keeps serving incremental passes until then)
RP_DEV void bp_large_append(DevWorld &w, int i) {
    int k = atomicAdd(&w.scan_block[BP_LARGE_SCRATCH], 1);
    if (k < w.large_cap) w.large_list[k] = i; else atomicOr(&w.flags[FL_OVERFLOW], RP_OVF_LARGE);
}
// Batches of small worlds (n_sub > 1): the large list just built, counting-sorted by sub-world; large_sub_begin[s] .. [s + 1] is
// sub-world s's segment (large_range_of).  ONE workgroup (the list holds a slab or two per sub-world), behind the build pass's barrier.
RP_DEV void bp_large_by_sub(DevWorld &w, int nl) {
    __shared__ int part[1024];
    const int t = threadIdx.x, n = w.n_sub + 1; // entries [0, n_sub]: counts shifted by one, then their inclusive prefix = the begins
    if (nl > w.large_cap) nl = w.large_cap;
    for (int s = t; s <= w.n_sub + 1 && s <= w.sub_cap + 1; s += blockDim.x) w.large_sub_begin[s] = 0;
    __threadfence(); __syncthreads();
    for (int q = t; q < nl; q += blockDim.x) { const int L = w.large_list[q]; w.large_tmp[q] = L; atomicAdd(&w.large_sub_begin[w.c_sub[L] + 1], 1); }
    __threadfence(); __syncthreads();
    const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x, lo = t * per, hi = lo + per < n ? lo + per : n;
    int sum = 0;
    for (int s = lo; s < hi; ++s) sum += __hip_atomic_load(&w.large_sub_begin[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < (int)blockDim.x; off <<= 1) { int v = t >= off ? part[t - off] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
    int run = part[t] - sum;
    for (int s = lo; s < hi; ++s) { run += __hip_atomic_load(&w.large_sub_begin[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); w.large_sub_begin[s] = run; w.large_sub_cur[s] = run; }
    __threadfence(); __syncthreads();
    // (entry s now holds the END of sub-world s - 1 = the begin of sub-world s; entry 0 = 0)
    for (int q = t; q < nl; q += blockDim.x) { const int L = w.large_tmp[q]; const int pos = atomicAdd(&w.large_sub_cur[w.c_sub[L]], 1); w.large_list[pos] = L; }
}
RP_DEV void bp_build(DevWorld &w, int gid, int gstride, int nxt) {
    int *cnt = w.bk_cnt[nxt]; int *items = w.bk_items[nxt];
    for (int i = gid; i < w.n_colliders; i += gstride) {
        CellRange r = cell_range(w, i);
        w.c_inlarge[i] = r.large ? 1 : 0; w.c_rver[i] = 0; // the grid is being rebuilt: every range version starts from zero
        if (r.large) { bp_large_append(w, i); continue; }
        bool full = false;
        int o = 0;
        for (int z = r.lo[2]; z <= r.hi[2]; ++z)
            for (int y = r.lo[1]; y <= r.hi[1]; ++y)
                for (int x = r.lo[0]; x <= r.hi[0]; ++x, ++o) {
                    unsigned long long key = cell_key_of(w, i, x, y, z);
                    int h = (int)(rp_hash64(key) & (unsigned long long)(w.grid_cap - 1));
                    int k = atomicAdd(&cnt[h], 1);
                    if (k < RP_BP_BUCKET) items[(size_t)h * RP_BP_BUCKET + k] = bp_entry(i, o, 0);
                    else full = true;
                }
        if (full) { w.c_inlarge[i] = 1; bp_large_append(w, i); } // (only this thread writes c_inlarge[i] in this pass; readers sit behind the barrier)
    }
}

__device__ __forceinline__ bool fat_overlap(const DevWorld &w, int a, int b, V3 &imin) {
    float4 amn = w.c_fatmin[a], amx = w.c_fatmax[a], bmn = w.c_fatmin[b], bmx = w.c_fatmax[b];
    bool ov = amn.x <= bmx.x && bmn.x <= amx.x && amn.y <= bmx.y && bmn.y <= amx.y && amn.z <= bmx.z && bmn.z <= amx.z;
    imin = v3(fmaxf(amn.x, bmn.x), fmaxf(amn.y, bmn.y), fmaxf(amn.z, bmn.z));
    return ov;
}
// update.rs:334-396: same parent / ActiveCollisionTypes::default() / InteractionGroups::test
__device__ __forceinline__ bool pair_allowed(const DevWorld &w, int a, int b) {
    if (w.n_sub > 1 && w.c_sub[a] != w.c_sub[b]) return false; // colliders of different sub-worlds never meet (rp_world_begin_subworld)
    int pa = w.c_parent[a], pb = w.c_parent[b];
    if (pa >= 0 && pa == pb) return false;
    bool da = pa >= 0 && (w.b_flags[pa] & RP_BF_TYPE_MASK) == RP_BODY_DYNAMIC;
    bool db = pb >= 0 && (w.b_flags[pb] & RP_BF_TYPE_MASK) == RP_BODY_DYNAMIC;
    if (!da && !db) return false;
    uint2 ga = w.c_groups[a], gb = w.c_groups[b];
    return (ga.x & gb.y) != 0 && (gb.x & ga.y) != 0;
}

__device__ __forceinline__ int hash_find(const unsigned long long *keys, const int *slots, int cap, unsigned long long key) {
    int h = (int)(rp_hash64(key) & (unsigned long long)(cap - 1));
    for (int probe = 0; probe < cap; ++probe) {
        unsigned long long k = keys[h];
        if (k == key) return slots[h];
        if (k == RP_EMPTY_KEY) return -1;
        h = (h + 1) & (cap - 1);
    }
    return -1;
}

RP_DEV bool pair_touches_island(const DevWorld &w, int rb1, int rb2) { return (rb1 >= 0 && w.b_island[rb1] >= 0) || (rb2 >= 0 && w.b_island[rb2] >= 0); }
// A pair slot for every lane that is active here: the free stack first, then the bump allocator — ONE atomic per wavefront and
// counter instead of one per lane (the first step of b3d_many_pyramids creates 28,420 pairs: their 85 k same-address atomics were
// the whole cost of the pair pass once it ran wide)
__device__ int pair_slot_alloc(DevWorld &w) {
    const unsigned long long m = __ballot(1);
    const int lane = threadIdx.x & 63, n = __popcll(m), leader = __ffsll((long long)m) - 1, rank = __popcll(m & ((1ull << lane) - 1ull));
    int t = 0, p = 0;
    if (lane == leader) {
        t = atomicSub(&w.flags[FL_FREE_TOP], n);
        const int from_free = t > 0 ? (t < n ? t : n) : 0, from_pool = n - from_free;
        if (from_pool) { atomicAdd(&w.flags[FL_FREE_TOP], from_pool); p = atomicAdd(&w.flags[FL_POOL_TOP], from_pool); }
    }
    t = __shfl(t, leader, 64); p = __shfl(p, leader, 64);
    const int from_free = t > 0 ? (t < n ? t : n) : 0;
    return rank < from_free ? w.free_stack[t - 1 - rank] : p + (rank - from_free);
}
// AddPair (NarrowPhase::add_pair, pair_management.rs:572): find-or-create the pair slot and
// register it in the next-epoch table.  Each unordered pair reaches this exactly once per rebuild.
__device__ void bp_insert_pair(DevWorld &w, int c1, int c2, bool incremental = false) {
    int epoch = w.flags[FL_BP_EPOCH];
    int cur = epoch & 1, nxt = cur ^ 1;
    unsigned long long key = ((unsigned long long)(unsigned)c1 << 32) | (unsigned)c2;
    int slot = hash_find(w.h_key[cur], w.h_slot[cur], w.hash_cap, key);
    if (incremental && slot >= 0) return; // the pair lives on
    if (slot < 0) {
        slot = pair_slot_alloc(w);
        if (slot >= w.pool_cap) { atomicOr(&w.flags[FL_OVERFLOW], RP_OVF_POOL); return; }
        w.p_c1[slot] = c1; w.p_c2[slot] = c2; w.p_rb[slot] = make_int2(w.c_parent[c1], w.c_parent[c2]);
        w.p_color[slot] = RP_COLOR_UNCOLORED; w.p_nsc[slot] = 0; w.p_npts[slot] = 0; w.p_pflags[slot] = 0;
        w.p_reldom[slot] = 0; w.p_colorb[slot] = make_int2(-1, -1); w.p_conspos[slot] = -1; w.p_hint_seq[slot] = 0;
        w.p_aux[slot] = make_int4(-1, -1, -1, 0); w.p_sub[slot] = make_int2(-1, -1);
        w.p_ln1[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); w.p_ln2[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // (a recycled slot: the last normal is GJK's first guess, rp_convex.h)
        // the lists of an LDS island also hold its pairs WITHOUT solver contacts (the fused step recycle-tests them): a new pair
        // changes the layout only when one of its bodies lives in such an island — not when both sit on the global path (the
        // creeping 20,100-body island of b3d_large_pyramid gains and loses near-miss pairs every step)
        if (pair_touches_island(w, w.c_parent[c1], w.c_parent[c2])) w.flags[FL_LAYOUT_DIRTY] = 1;
    }
    w.p_stamp[slot] = incremental ? epoch : epoch + 1;
    const int tab = incremental ? cur : nxt; // an incremental pass keeps the live table; a rebuild fills the next one
    int h = (int)(rp_hash64(key) & (unsigned long long)(w.hash_cap - 1));
    for (int probe = 0; probe < w.hash_cap; ++probe) {
        unsigned long long prev = atomicCAS(&w.h_key[tab][h], RP_EMPTY_KEY, key);
        if (prev == RP_EMPTY_KEY) { w.h_slot[tab][h] = slot; return; }
        h = (h + 1) & (w.hash_cap - 1);
    }
    atomicOr(&w.flags[FL_OVERFLOW], RP_OVF_HASH);
}

// The pair pass of a full rebuild: EIGHT lanes per collider, each walking every eighth cell of the collider's range (at most 27; a
// pair is reported from the cell that holds the min corner of the two fat AABBs' intersection, so once) and every eighth entry of
// the (short) list of large colliders — ground slabs, walls.  One thread per collider walked its 27 cells one after the other: a
// chain of ~100 dependent L2 round trips, 150 us of the 213 us rebuild on b3d_large_pyramid (tools/pass_profile.py); a whole
// wavefront per collider does not fit the resident grid of a barrier kernel (16 rounds: slower).
#define BP_GROUP 8
RP_DEV void bp_pairs(DevWorld &w, int nxt, int nl) { // nl: the large list that goes with grid copy `nxt` (just built: scan_block's count; kept: FL_N_LARGE)
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, sub = tid & (BP_GROUP - 1), ngroups = (gridDim.x * blockDim.x) / BP_GROUP;
    const float ic = w.prm.inv_cell_size;
    if (nl > w.large_cap) nl = w.large_cap;
    const int *cnt = w.bk_cnt[nxt]; const int *items = w.bk_items[nxt];
    for (int i = tid / BP_GROUP; i < w.n_colliders; i += ngroups) {
        const bool ilarge = w.c_inlarge[i] != 0;
        int q0, q1; large_range_of(w, i, nl, q0, q1);
        for (int q = q0 + sub; q < q1; q += BP_GROUP) {
            int L = w.large_list[q];
            if (L == i || (ilarge && i > L)) continue; // large-large pairs reported from the lower index
            V3 imin;
            if (!fat_overlap(w, i, L, imin) || !pair_allowed(w, i, L)) continue;
            bp_insert_pair(w, i < L ? i : L, i < L ? L : i);
        }
        if (ilarge) continue;
        CellRange r = cell_range(w, i);
        const int nx = r.hi[0] - r.lo[0] + 1, ny = r.hi[1] - r.lo[1] + 1, nz = r.hi[2] - r.lo[2] + 1;
        for (int c = sub; c < nx * ny * nz; c += BP_GROUP) {
            const int x = r.lo[0] + c % nx, y = r.lo[1] + (c / nx) % ny, z = r.lo[2] + c / (nx * ny);
            unsigned long long key = cell_key_of(w, i, x, y, z);
            int h = (int)(rp_hash64(key) & (unsigned long long)(w.grid_cap - 1));
            int n = cnt[h]; if (n > RP_BP_BUCKET) n = RP_BP_BUCKET;
            for (int e = 0; e < n; ++e) {
                const int it = items[(size_t)h * RP_BP_BUCKET + e], j = it & 0xffffff;
                if (j <= i) continue;
                V3 imin;
                if (!fat_overlap(w, i, j, imin)) continue;
                if (cell_coord(imin.x, ic) != x || cell_coord(imin.y, ic) != y || cell_coord(imin.z, ic) != z) continue;
                if (!bp_entry_is_cell(w, it, x, y, z)) continue; // (another cell of j that shares this bucket)
                if (w.c_inlarge[j]) continue; // (a collider on the large list was met above; asked last: few entries get this far)
                if (!pair_allowed(w, i, j)) continue;
                bp_insert_pair(w, i, j);
            }
        }
    }
}

// DeletePair (NarrowPhase::remove_pair, pair_management.rs:382) of pair slot s: free the colour, raise the events and wake-ups,
// journal the unlink, recycle the slot.
// (`deferred`: the slot is parked in free_pending instead of going onto the free stack — an incremental pass inserts and deletes in
// ONE pass, and its inserts pop that stack meanwhile; the pass's last workgroup moves the parked slots over)
RP_DEV void bp_delete_pair(DevWorld &w, int s, bool deferred = false) {
    int color = w.p_color[s];
    if (color < RP_COLOR_OVERFLOW) {
        int2 cb = w.p_colorb[s];
        unsigned bit = 1u << (color & 31);
        if (cb.x >= 0) atomicAnd(&w.b_cmask[4 * cb.x + (color >> 5)], ~bit);
        if (cb.y >= 0) atomicAnd(&w.b_cmask[4 * cb.y + (color >> 5)], ~bit);
    }
    { // a dead pair changes the layout when it was a solver manifold or when an LDS island lists it (see bp_insert_pair)
        int2 drb = w.p_rb[s];
        if (w.p_nsc[s] > 0 || pair_touches_island(w, drb.x, drb.y)) w.flags[FL_LAYOUT_DIRTY] = 1;
    }
    if (w.p_nsc[s] > 0 && pair_wants_collision_events(w, w.p_c1[s], w.p_c2[s])) {
        // Stopped event of a touching pair: remove_pair (pair_management.rs:554-558), remove_collider (:101-110, REMOVED)
        uint2 e1 = w.c_groups[w.p_c1[s]], e2 = w.c_groups[w.p_c2[s]];
        bool removed = (e1.x == 0 && e1.y == 0) || (e2.x == 0 && e2.y == 0);
        push_collision_event(w, w.p_c1[s], w.p_c2[s], 0, removed ? RP_COLLISION_EVENT_REMOVED : 0, cur_step(w));
    }
    if ((w.p_pflags[s] & RP_PF_INTERSECTING) && pair_wants_collision_events(w, w.p_c1[s], w.p_c2[s])) {
        // remove_pair / remove_collider on the intersection graph (pair_management.rs:382-460): Stopped | SENSOR
        uint2 e1 = w.c_groups[w.p_c1[s]], e2 = w.c_groups[w.p_c2[s]];
        bool removed = (e1.x == 0 && e1.y == 0) || (e2.x == 0 && e2.y == 0);
        push_collision_event(w, w.p_c1[s], w.p_c2[s], 0, (removed ? RP_COLLISION_EVENT_REMOVED : 0) | RP_COLLISION_EVENT_SENSOR, cur_step(w));
    }
    if (w.sleep_enabled) {
        // remove_pair wakes the bodies of a touching pair (pair_management.rs:541-552); remove_collider wakes every
        // body that had a pair with the removed collider (:88-99)
        int c1 = w.p_c1[s], c2 = w.p_c2[s];
        uint2 g1 = w.c_groups[c1], g2 = w.c_groups[c2];
        bool gone = (g1.x == 0 && g1.y == 0) || (g2.x == 0 && g2.y == 0);
        int2 rb = w.p_rb[s];
        if (w.p_nsc[s] > 0 || gone) {
            if (rb.x >= 0) atomicMax(&w.b_wake_req[rb.x], 2);
            if (rb.y >= 0) atomicMax(&w.b_wake_req[rb.y], 2);
        }
        if (w.p_nsc[s] > 0) pi_journal(w, rb.x, rb.y, 1, c1, c2); // unlink_contact of a removed touching pair (pair_management.rs:531)
    }
    if (w.has_composite) aux_free_all(w, s, deferred); // the clusters of a composite pair go with it
    // (BOTH collider fields are scrubbed: an incremental pass inserts and deletes in one pass, and its delete sweep may look at a slot
    // that an insert is filling at that moment — with -1 in whichever field has not landed yet the sweep skips it, see bp_incr_delete)
    w.p_c1[s] = -1; w.p_c2[s] = -1; w.p_nsc[s] = 0; w.p_npts[s] = 0; w.p_color[s] = RP_COLOR_UNCOLORED;
    if (deferred) { int t = atomicAdd(&w.flags[FL_BP_NFREED], 1); w.free_pending[t] = s; return; }
    int t = atomicAdd(&w.flags[FL_FREE_TOP], 1);
    w.free_stack[t] = s;
}

// DeletePair of a rebuild: slots not re-stamped by it are dead.
RP_DEV void bp_finish_pairs(DevWorld &w, int gid, int gstride, bool keep_grid, int gcur) {
    int epoch = w.flags[FL_BP_EPOCH];
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    for (int s = gid; s < top; s += gstride) {
        if (w.p_c1[s] < 0) continue;
        if (w.has_composite && pair_is_aux(w, s)) continue; // (a cluster of a composite pair: deleted with its parent)
        if (w.p_stamp[s] == epoch + 1) continue;
        bp_delete_pair(w, s);
    }
    // rest state for the next rebuild: the grid copy and the pair table that go out of service (they become the next rebuild's targets)
    if (!keep_grid) { int *old = w.bk_cnt[gcur]; for (int i = gid; i < w.grid_cap; i += gstride) old[i] = 0; }
    { unsigned long long *old = w.h_key[epoch & 1]; for (int i = gid; i < w.hash_cap; i += gstride) old[i] = RP_EMPTY_KEY; }
    if (gid == 0 && !keep_grid) { // the new large list goes into service with the new grid
        const int nl = w.scan_block[BP_LARGE_SCRATCH];
        w.flags[FL_N_LARGE] = nl < w.large_cap ? nl : w.large_cap;
        w.scan_block[BP_LARGE_SCRATCH] = 0;
    }
}

// ---- incremental pass ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void hash_erase(unsigned long long *keys, int cap, unsigned long long key) {
    int h = (int)(rp_hash64(key) & (unsigned long long)(cap - 1));
    for (int probe = 0; probe < cap; ++probe) {
        unsigned long long k = keys[h];
        if (k == key) { keys[h] = RP_TOMB_KEY; return; }
        if (k == RP_EMPTY_KEY) return;
        h = (h + 1) & (cap - 1);
    }
}
// Did the pair (i, j) exist after the LAST pass?  The pair set of a pass is exactly the allowed pairs of overlapping fat AABBs, so it
// did iff the boxes the two colliders had then overlap: a collider queued in this pass keeps that box in c_fatold (collider_update_one),
// everybody else still has it.  The incremental pass skips such partners before any of the status words, the filter and the hash probe
// are fetched — on b3d_joint_grid (a third of the fat AABBs rewritten per pass) nearly every candidate is an existing pair.
struct BpOld { float4 mn, mx; bool valid; };
RP_DEV bool box_valid(float4 mn, float4 mx) { return mn.x <= mx.x && mn.y <= mx.y && mn.z <= mx.z; }
RP_DEV bool box_overlap(float4 amn, float4 amx, float4 bmn, float4 bmx) { return amn.x <= bmx.x && bmn.x <= amx.x && amn.y <= bmx.y && bmn.y <= amx.y && amn.z <= bmx.z && bmn.z <= amx.z; }
RP_DEV bool bp_pair_existed(const DevWorld &w, const BpOld &oi, int j, bool jchg, float4 jmn, float4 jmx) {
    if (!oi.valid) return false;
    if (jchg) { jmn = w.c_fatold_min[j]; jmx = w.c_fatold_max[j]; if (!box_valid(jmn, jmx)) return false; }
    return box_overlap(oi.mn, oi.mx, jmn, jmx);
}
__device__ __forceinline__ void bp_try_pair(DevWorld &w, int i, int j, const BpOld &oi, int stamp) {
    const float4 imn = w.c_fatmin[i], imx = w.c_fatmax[i], jmn = w.c_fatmin[j], jmx = w.c_fatmax[j];
    if (!box_overlap(imn, imx, jmn, jmx)) return;
    if (bp_pair_existed(w, oi, j, w.c_chgstamp[j] == stamp, jmn, jmx)) return;
    if (!pair_allowed(w, i, j)) return;
    bp_insert_pair(w, i < j ? i : j, i < j ? j : i, true);
}
// new partners of the colliders on bp_chg_list.  Round 5, third form: a WAVEFRONT per changed collider, its 64 lanes spread over
// (cell, bucket slot) — eight cells at a time, eight lanes per cell, lane s of a cell takes the bucket entries s, s + 8, ... — so a
// lane's dependent chain is one or two entries long (bucket count -> entry -> the partner's box -> its stamp), not the whole bucket of
// its cell (the eight-lanes-per-collider form: a cell per lane, its 4-12 entries one after the other: ~25 dependent loads).  Consecutive
// list entries go to DIFFERENT workgroups: a few thousand changed colliders are a wavefront or two on every CU, not sixteen wavefronts of
// scattered 16-byte loads on the first thirty CUs.  (Round 3's wavefront form gave a lane a whole CELL: 56 of 64 lanes idle on a ball.)
RP_DEV void bp_incr_insert(DevWorld &w, int nchg) {
    const int lane = threadIdx.x & 63, cl = lane >> 3, sl = lane & 7;
    const int stamp = w.flags[FL_BP_SEQ] + 1;
    const float ic = w.prm.inv_cell_size;
    int nl = w.flags[FL_N_LARGE]; if (nl > w.large_cap) nl = w.large_cap;
    const int cur = BP_GPAR(w); // the grid copy in service
    const int *cnt = w.bk_cnt[cur]; const int *items = w.bk_items[cur];
    const int waves_per_block = (int)blockDim.x >> 6;
    for (int k = (int)(threadIdx.x >> 6) * (int)gridDim.x + (int)blockIdx.x; k < nchg; k += waves_per_block * (int)gridDim.x) {
        const int i = w.bp_chg_list[k];
        CellRange r = cell_range(w, i);
        if (r.large || w.c_inlarge[i]) continue; // (never here: bp_grid_follow raised FL_BP_FORCE_FULL when it rewrote that AABB, and this launch chose the full rebuild)
        // (a) the large colliders (ground slabs, walls: never stale)
        BpOld oi; oi.mn = w.c_fatold_min[i]; oi.mx = w.c_fatold_max[i]; oi.valid = box_valid(oi.mn, oi.mx); // (queued in this pass: c_fatold holds the box of the last pass)
        { int q0, q1; large_range_of(w, i, nl, q0, q1); for (int q = q0 + lane; q < q1; q += 64) bp_try_pair(w, i, w.large_list[q], oi, stamp); }
        // (b) everybody else through the grid, which follows every collider (bp_grid_follow): a pair is reported from the cell that
        // holds the min corner of the intersection, which lies in both cell ranges; two colliders that both changed in this pass find
        // each other — reported from the smaller index
        const int nx = r.hi[0] - r.lo[0] + 1, ny = r.hi[1] - r.lo[1] + 1, nz = r.hi[2] - r.lo[2] + 1;
        const float4 imn = w.c_fatmin[i], imx = w.c_fatmax[i];
        for (int c = cl; c < nx * ny * nz; c += 8) {
            const int x = r.lo[0] + c % nx, y = r.lo[1] + (c / nx) % ny, z = r.lo[2] + c / (nx * ny);
            unsigned long long key = cell_key_of(w, i, x, y, z);
            int h = (int)(rp_hash64(key) & (unsigned long long)(w.grid_cap - 1));
            int n = cnt[h]; if (n > RP_BP_BUCKET) n = RP_BP_BUCKET;
            for (int e = sl; e < n; e += 8) {
                const int it = items[(size_t)h * RP_BP_BUCKET + e], j = it & 0xffffff;
                if (j == i) continue;
                const float4 jmn = w.c_fatmin[j], jmx = w.c_fatmax[j];
                if (!box_overlap(imn, imx, jmn, jmx)) continue; // (geometry first: the status words below are only fetched for the few entries that get past it)
                const bool jchg = w.c_chgstamp[j] == stamp;
                if (bp_pair_existed(w, oi, j, jchg, jmn, jmx)) continue; // an existing pair: nothing to insert (the common case by far)
                if (cell_coord(fmaxf(imn.x, jmn.x), ic) != x || cell_coord(fmaxf(imn.y, jmn.y), ic) != y || cell_coord(fmaxf(imn.z, jmn.z), ic) != z) continue;
                if (w.c_inlarge[j] || (jchg && j < i)) continue; // the large list: covered by (a)
                if (!bp_entry_is_cell(w, it, x, y, z)) continue; // (another cell of j that shares this bucket, or an entry of a cell range j has left)
                if (!pair_allowed(w, i, j)) continue;
                bp_insert_pair(w, i < j ? i : j, i < j ? j : i, true);
            }
        }
    }
}
// lost partners: every pair with a changed collider whose fat AABBs no longer intersect
RP_DEV void bp_incr_delete(DevWorld &w, int gid, int gstride, int nchg) {
    const int stamp = w.flags[FL_BP_SEQ] + 1, cur = w.flags[FL_BP_EPOCH] & 1;
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    for (int s = gid; s < top; s += gstride) {
        const int c1 = w.p_c1[s];
        if (c1 < 0) continue;
        const int c2 = w.p_c2[s];
        if (c2 < 0) continue; // (a recycled slot that an insert of this very pass is filling: half written — a new pair, nothing to delete)
        if (w.has_composite && pair_is_aux(w, s)) continue;
        if (w.c_chgstamp[c1] != stamp && w.c_chgstamp[c2] != stamp) continue;
        V3 imin;
        if (fat_overlap(w, c1, c2, imin)) continue;
        hash_erase(w.h_key[cur], w.hash_cap, ((unsigned long long)(unsigned)c1 << 32) | (unsigned)c2);
        // (FL_BP_TOMBS is NOT touched here: every workgroup of this launch decides `incremental` from it, at its own time — the tombstones
        // of this pass are the slots it parks, counted by FL_BP_NFREED and folded into FL_BP_TOMBS by bp_close_incremental)
        bp_delete_pair(w, s, true);
    }
}

// The whole pass in ONE launch (see rp_gridbar.h): a clean step (no fat AABB changed) costs a single early exit, a step in which few
// colliders left their fat AABBs two passes over those colliders and the pair slots, anything else the full rebuild.
__global__ void __launch_bounds__(1024) k_bp_rebuild(DevWorld w) {
    if (!w.flags[FL_BP_DIRTY]) return; // (cleared only after the last barrier: every workgroup reads the same value)
    if (collision_done(w)) return;     // (a dead lean step's collision stage is not repeated: rp_world.h "lean step graphs")
    const int gid = gbar_item(), gstride = gridDim.x * blockDim.x;
    // the mode is decided from scalars that NO workgroup of this launch writes before its first barrier (the incremental pass has none:
    // it writes none of them at all — a late workgroup must not see a different answer than an early one that is already deleting)
    // (an incremental pass spends a wavefront per changed collider, the full rebuild eight lanes per collider: beyond a quarter of the
    // colliders the rebuild is the cheaper one — measured on b3d_joint_grid, where 3,559 of 10,000 change per pass: 36 us as a rebuild)
    const int nchg = w.flags[FL_BP_NCHG] < w.n_colliders ? w.flags[FL_BP_NCHG] : w.n_colliders;
    const bool incremental = w.bp_incremental && w.flags[FL_BP_GRID_OK] && !w.flags[FL_BP_FORCE_FULL] && nchg > 0 && nchg <= w.n_colliders / w.bp_incr_div + 16 && w.flags[FL_BP_TOMBS] < w.hash_cap / 8;
    GridBar bar = gbar_begin(w, 0);
#ifdef RP_PASS_PROFILE // why a pass was (not) incremental: dbg[240..] (tools/pass_profile.py)
    if (gid == 0) {
        w.dbg[240] += 1; w.dbg[241] += incremental ? 1 : 0; w.dbg[242] += w.flags[FL_BP_GRID_OK] ? 0 : 1; w.dbg[243] += (nchg > w.n_colliders / w.bp_incr_div + 16) ? 1 : 0;
        w.dbg[244] += 0; w.dbg[245] += (w.flags[FL_BP_TOMBS] >= w.hash_cap / 8) ? 1 : 0; w.dbg[246] += nchg; w.dbg[247] += 0; w.dbg[248] = w.flags[FL_N_LARGE];
    }
#endif
    if (incremental) {
        // ONE pass, no grid barrier (round 5): new partners of the changed colliders and the pairs they lost are independent of each
        // other — a pair is in exactly one of the two sets — the hash table takes CAS inserts and tombstones side by side, and freed
        // slots are parked (bp_delete_pair deferred) while the inserts pop the free stack.
        bp_incr_insert(w, nchg);
        bp_incr_delete(w, gid, gstride, nchg);
        // The pass is CLOSED by the next kernel of the step (bp_close_incremental, workgroup 0 of k_np_test: parked slots onto the free
        // stack, counters, the dirty flag): behind a kernel boundary, not behind a last-workgroup ticket — the agent-scope release fence
        // every workgroup paid for that ticket was most of a pass that found nothing to do (b3d_joint_grid: not one pair, ~4,000 fat
        // AABBs rewritten per pass: 19 us).
        if (gid == 0) w.flags[FL_BP_CLOSE] = 1;
        return;
    }
    const int epoch = w.flags[FL_BP_EPOCH];
    // Round 4: the grid in service follows every collider (bp_grid_follow), so a rebuild of the PAIR SET — what a step needs when many fat
    // AABBs were rewritten — can read it as it stands: build | pairs | finish becomes pairs | finish.  The build pass (which also renews
    // the large list and compacts the buckets: entries of left cell ranges die with it) runs when the grid is not in order, when a
    // large collider moved, and every 32nd rebuild.  (Scalars read here only change behind this launch's barriers or in the launch before.)
    const int gcur = BP_GPAR(w);
    const bool keep_grid = w.bp_incremental && w.flags[FL_BP_GRID_OK] && !w.flags[FL_BP_FORCE_FULL] && w.lay_state[9] < 32;
    RP_PASS_BEGIN();
    if (!keep_grid) {
        bp_build(w, gid, gstride, gcur ^ 1); GBAR_SYNC(bar);
        if (w.n_sub > 1) { if (blockIdx.x == 0) bp_large_by_sub(w, w.scan_block[BP_LARGE_SCRATCH]); GBAR_SYNC(bar); } // (n_sub: the same for every workgroup)
    }
    RP_PASS_STAMP(w, 220);
    bp_pairs(w, keep_grid ? gcur : gcur ^ 1, keep_grid ? w.flags[FL_N_LARGE] : w.scan_block[BP_LARGE_SCRATCH]);
    GBAR_SYNC(bar); RP_PASS_STAMP(w, 220);
    bp_finish_pairs(w, gid, gstride, keep_grid, gcur);
    GBAR_SYNC(bar); RP_PASS_STAMP(w, 220);
    gbar_end(bar);
    if (gid == 0) { // the rebuild is closed: epoch flip, dirty flag; the grid is valid and nobody is stale
        w.flags[FL_BP_EPOCH] = epoch + 1;
        w.flags[FL_BP_REBUILDS] += 1;
        w.flags[FL_BP_NCHG] = 0; w.flags[FL_BP_TOMBS] = 0; w.flags[FL_BP_FORCE_FULL] = 0; w.flags[FL_BP_SEQ] += 1;
        if (keep_grid) w.lay_state[9] += 1;
        else {
            w.lay_state[8] = gcur ^ 1; w.lay_state[9] = 0;
            w.flags[FL_BP_GRID_OK] = (w.flags[FL_OVERFLOW] & (RP_OVF_CELLS | RP_OVF_LARGE)) ? 0 : 1;
        }
        __hip_atomic_store(&w.flags[FL_BP_DIRTY], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The pairs of removed colliders (groups zeroed: ColliderSet::remove) leave the pair set at once — what the next pass would do to
// them — so that the collider's arena slot can be handed out again before a step has run (rp_api.hip purge_dead_pairs).
__global__ void k_purge_dead_pairs(DevWorld w) {
    const int cur = w.flags[FL_BP_EPOCH] & 1;
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    const int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < top; s += stride) {
        const int c1 = w.p_c1[s];
        if (c1 < 0) continue;
        if (w.has_composite && pair_is_aux(w, s)) continue; // (a cluster of a composite pair: its parent frees it)
        const int c2 = w.p_c2[s];
        const uint2 g1 = w.c_groups[c1], g2 = w.c_groups[c2];
        if (!((g1.x == 0 && g1.y == 0) || (g2.x == 0 && g2.y == 0))) continue;
        hash_erase(w.h_key[cur], w.hash_cap, ((unsigned long long)(unsigned)c1 << 32) | (unsigned)c2);
        atomicAdd(&w.flags[FL_BP_TOMBS], 1);
        bp_delete_pair(w, s);
    }
}
void rp_launch_purge_dead_pairs(const DevWorld &w, hipStream_t st) {
    int blocks = (w.pool_cap + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_purge_dead_pairs, dim3(blocks), dim3(256), 0, st, w);
}

void rp_launch_collider_update(const DevWorld &w, hipStream_t st) {
    if (w.n_colliders == 0) return;
    hipLaunchKernelGGL(k_collider_update, dim3((w.n_colliders + 255) / 256), dim3(256), 0, st, w);
}

void rp_launch_fast_front(const DevWorld &w, hipStream_t st, int no_global_kernel) {
    int n = w.n_colliders > w.pool_cap ? w.n_colliders : w.pool_cap;
    int blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    int need = (w.n_colliders + 255) / 256; if (blocks < need) blocks = need;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_fast_front, dim3(blocks), dim3(256), 0, st, w, no_global_kernel);
}

// A world that moved to larger arrays (rp_api.hip: carry_over): its live pairs enter the current-epoch hash table, whose size changed.
__global__ void k_bp_rehash(DevWorld w) {
    int cur = w.flags[FL_BP_EPOCH] & 1;
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < top; s += stride) {
        int c1 = w.p_c1[s];
        if (c1 < 0) continue;
        unsigned long long key = ((unsigned long long)(unsigned)c1 << 32) | (unsigned)w.p_c2[s];
        int h = (int)(rp_hash64(key) & (unsigned long long)(w.hash_cap - 1));
        bool placed = false;
        for (int probe = 0; probe < w.hash_cap && !placed; ++probe) {
            unsigned long long prev = atomicCAS(&w.h_key[cur][h], RP_EMPTY_KEY, key);
            if (prev == RP_EMPTY_KEY) { w.h_slot[cur][h] = s; placed = true; }
            h = (h + 1) & (w.hash_cap - 1);
        }
        if (!placed) atomicOr(&w.flags[FL_OVERFLOW], RP_OVF_HASH);
    }
}
void rp_launch_bp_rehash(const DevWorld &w, hipStream_t st) {
    int blocks = (w.pool_cap + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_bp_rehash, dim3(blocks), dim3(256), 0, st, w);
}

void rp_launch_broadphase(const DevWorld &w, hipStream_t st) {
    if (w.n_colliders == 0) return;
    // every workgroup must be resident (grid barriers): at most DevWorld::gbar_blocks workgroups of 1024 threads (rp_gridbar.h)
    int blocks = (w.n_colliders + 127) / 128; // the pair pass gives every collider 8 lanes (BP_GROUP): one round when the grid allows
    { const int wide = 32; // (the incremental pass: a wavefront per changed collider, 16 per workgroup)
      const int b2 = (w.n_colliders + wide - 1) / wide; if (b2 > blocks) blocks = b2; }
    if (blocks < 8) blocks = 8;    // the clears and the pair-slot sweep are sized by capacities, not by the collider count
    if (blocks > w.gbar_blocks) blocks = w.gbar_blocks;
    hipLaunchKernelGGL(k_bp_rebuild, dim3(blocks), dim3(1024), 0, st, w);
}

// workgroups of k_bp_rebuild (1024 threads) one CU holds at once (0 = the query failed): input of DevWorld::gbar_blocks (rp_api.hip)
int rp_occ_bp_rebuild(void) { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bp_rebuild, 1024, 0) != hipSuccess) n = 0; return n; }
