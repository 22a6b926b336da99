// rp_islands.hip — contact islands and the LDS-resident per-island TGS megakernel.
//
// The reference solves all awake bodies as ONE colour-ordered Gauss-Seidel system
// (/root/reference/src/dynamics/island_manager/manager.rs:20-39, staged_island_solver/worker.rs) and
// pays a barrier per colour per sweep.  Constraints of different connected components never share a
// body, so sweeping each component through the same colour sequence independently is bit-identical
// to the global sweep (SURVEY Appendix B.3 applied across components).  On MI355X that turns ~100
// dependent kernel launches per step into ONE launch: each workgroup owns one island, keeps every
// manifold's constraint in the VGPRs of a lane pair and the solver bodies (64 B each) in LDS, and runs
// generate -> 4 x (increment + body-centric warm start, biased sweep, integrate, pose stage, relaxed
// sweep) -> restitution -> write-back with workgroup barriers between colour stages.  HBM is touched
// once per step per manifold (pair data in, impulses out) instead of 12 sweeps.  The kernel is bound
// by the latency of its ~60 dependent stages, not by bandwidth.
// Islands that do not fit (more than RP_ISL_NB_MAX bodies or RP_ISL_NC_MAX manifolds) and bodies
// without contacts stay on the global per-colour path (rp_solver.hip).
//
// Island discovery (only when the active-manifold set changed): lock-free union-find over the active
// dynamic-dynamic pairs, then island numbering and list filling — five grid-wide kernels.
// persistent.rs keeps comparable connected components for sleeping; here they drive scheduling only.
#include "rp_global.h"
#include "rp_lanepair.h"
#include "rp_pairs.h"
#include "rp_gridbar.h"
#include "rp_island_stages.h"

RP_DEV int ld_i32(int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// find with path halving: a non-root node is re-pointed at its grandparent (an ancestor stays an ancestor under concurrent
// hooks, which only ever write roots), so the chains of a giant component — b3d_large_pyramid hooks 20,100 bodies into one —
// stay short however the unions interleave
RP_DEV int uf_find(int *label, int x) {
    int p = ld_i32(&label[x]);
    while (p != x) {
        int gp = ld_i32(&label[p]);
        if (gp != p) atomicCAS(&label[x], p, gp);
        x = p; p = gp;
    }
    return x;
}
RP_DEV unsigned uf_priority(int x) { return (unsigned)x * 2654435761u; } // hooking priority: a bijective scramble of the body index (see uf_union)

// Lock-free union: a root is hooked under the root of HIGHER priority, retry on races.  The priority is a bijective scramble of the
// body index, not the index itself: bodies are numbered along the rows of a stack, and "hook the larger index under the smaller"
// turned every row of b3d_large_pyramid into one 200-link chain (all its hooks succeed at once), which the finds of the same pass
// then walked at one L2 round trip per link — 330 us of a 510 us layout rebuild.  Random linking keeps the trees O(log n) deep.
// (Which member ends up as the root is irrelevant here: the labels only name components.)
RP_DEV void uf_union(int *label, int a, int b) {
    if (ld_i32(&label[a]) == ld_i32(&label[b])) return; // siblings (after compression: most pairs of a big component) — the root's line is not touched
    for (;;) {
        a = uf_find(label, a); b = uf_find(label, b);
        if (a == b) return;
        if (uf_priority(a) < uf_priority(b)) { int t = a; a = b; b = t; }
        if (atomicCAS(&label[a], a, b) == a) return;
    }
}
RP_DEV bool pair_active(const DevWorld &w, int s) { return pair_selected(w, s); }

// Island discovery runs only when the set of active manifolds changed (FL_LAYOUT_DIRTY): passes of k_layout_rebuild below,
// every pass one thread per body or per pair slot (gid / gstride span the whole launch).
// WARM START of the components (round 4).  The layout of a settling pile is rebuilt whenever a pair begins or ends to touch — 60 % of
// the steps of b3d_large_pyramid's first seconds — and 80 of the 215 us of a rebuild went into finding, from singleton labels, the ONE
// component the pile has been all along.  The labels of the last rebuild are a valid starting point whenever no body came, went or
// changed its kind since (lay_state[0]; sleep-enabled worlds change the awake set: always cold): unions over the current edges then
// only add what merged.  What a warm start cannot see is a SPLIT — so it is only taken while the global path holds a giant component
// (>= 4,096 bodies: a piece that left it stays on the global path with it, which is where bodies without an island go anyway; any
// partition into unions of components gives the same bits), and every 16th rebuild is cold so that pieces become islands again.
RP_DEV bool lay_warm(const DevWorld &w) {
    return !w.sleep_enabled && w.lay_state[0] == 1 && (w.lay_state[1] & 15) != 0 && w.lay_state[2] >= 4096; // ([2]: global-path bodies of the last rebuild — FL_N_GLOB_BODIES itself is reset by this launch)
}
RP_DEV void lay_isl_init(DevWorld &w, int gid, int gstride, bool warm) {
    const int i = gid;
    if (i == 0) { w.flags[FL_N_ISLANDS] = 0; w.flags[FL_N_GLOB_BODIES] = 0; w.flags[FL_ISL_BODY_CURSOR] = 0; w.flags[FL_ISL_CONS_CURSOR] = 0; w.flags[FL_ISL_ICONS_CURSOR] = 0; w.lay_state[4] = 0; w.lay_state[6] = 0; }
    for (size_t k = i, n = (size_t)128 * w.cb_words; k < n; k += (size_t)gstride) w.cb_bits[k] = 0u; // owner bitmaps of the colour stages
    for (int b = gid; b < w.n_bodies; b += gstride) { if (!warm) w.b_label[b] = b; w.r_nb[b] = 0; w.r_nc[b] = 0; w.r_ni[b] = 0; w.r_island[b] = -1; w.b_island[b] = -1; w.b_local[b] = -1; w.bun_nb[b] = 0; w.bun_nc[b] = 0; w.bun_ni[b] = 0; w.bun_id[b] = -1; }
}
// the edges of the island graph, compacted (one atomic per wavefront): active pairs whose two sides are awake non-fixed bodies
RP_DEV void lay_isl_edges(DevWorld &w, int gid, int gstride) {
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    const int lane = threadIdx.x & 63;
    for (int base = 0; base < top; base += gstride) { // wave-uniform trip count
        const int s = base + gid;
        int2 rb = make_int2(-1, -1);
        bool edge = false;
        if (s < top && pair_active(w, s)) { rb = w.p_rb[s]; edge = is_dyn(w, rb.x) && is_dyn(w, rb.y); }
        const unsigned long long m = __ballot(edge);
        if (!m) continue;
        int at = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) at = atomicAdd(&w.flags[FL_UF_NPAIRS], __popcll(m));
        at = __shfl(at, leader, 64);
        if (edge) w.uf_pairs[at + __popcll(m & ((1ull << lane) - 1ull))] = rb;
    }
}
// The components themselves.  Worlds of up to LAY_LDS_BODIES bodies: ONE workgroup runs the whole union-find in LDS (a hop costs ~100
// cycles there, an L2 round trip ~1,500: the 60 k edges of b3d_large_pyramid's single component took 143 us across the chip) and
// writes flat labels; larger worlds: every workgroup, in global memory.
#define LAY_LDS_BODIES 36864
RP_DEV int lds_find(int *label, int x) {
    int p = label[x];
    while (p != x) { int gp = label[p]; if (gp != p) atomicCAS(&label[x], p, gp); x = p; p = gp; }
    return x;
}
RP_DEV void lay_isl_union_lds(DevWorld &w, int *label, bool warm) { // workgroup 0
    for (int b = threadIdx.x; b < w.n_bodies; b += blockDim.x) label[b] = warm ? w.b_label[b] : b;
    __syncthreads();
    const int n = w.flags[FL_UF_NPAIRS];
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        int2 e = w.uf_pairs[k];
        int a = e.x, b = e.y;
        if (label[a] == label[b]) continue;
        for (;;) {
            a = lds_find(label, a); b = lds_find(label, b);
            if (a == b) break;
            if (uf_priority(a) < uf_priority(b)) { int t = a; a = b; b = t; }
            if (atomicCAS(&label[a], a, b) == a) break;
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < w.n_bodies; b += blockDim.x) w.b_label[b] = lds_find(label, b);
}
// connected components over the listed edges (worlds too large for the LDS form)
RP_DEV void lay_isl_union(DevWorld &w, int gid, int gstride) {
    const int n = w.flags[FL_UF_NPAIRS];
    for (int k = gid; k < n; k += gstride) { int2 e = w.uf_pairs[k]; uf_union(w.b_label, e.x, e.y); }
}
// flatten the labels; per-root body and manifold counts.  Lanes of a wavefront that name the same root (a giant component: all of them)
// add through ONE lane: the counts saturate just above the island limits, so what the others would do is a load of one hot line each.
RP_DEV void wave_add_saturating(int *counts, int root, int amount, int limit, bool active) {
    unsigned long long todo = __ballot(active);
    const int lane = threadIdx.x & 63;
    // components are mostly runs of neighbouring indices: when no lane shares its root with its neighbour the wavefront holds (about)
    // as many roots as lanes — b3d_joint_grid: 9,900 singletons — and grouping would only cost a round per lane
    if (!__ballot(active && lane > 0 && root == __shfl_up(root, 1, 64) && ((todo >> (lane - 1)) & 1ull))) {
        if (active && ld_i32(&counts[root]) <= limit) atomicAdd(&counts[root], amount);
        return;
    }
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int r = __shfl(root, leader, 64);
        const unsigned long long same = __ballot(active && root == r) & todo;
        // sum the amounts of the matching lanes (a 64-lane ballot walk is cheaper than a segmented reduction here: one or two groups per wave)
        int total = 0;
        for (unsigned long long m = same; m; m &= m - 1) total += __shfl(amount, __ffsll((long long)m) - 1, 64);
        if (lane == leader && ld_i32(&counts[r]) <= limit) atomicAdd(&counts[r], total > limit + 1 ? limit + 1 : total);
        todo &= ~same;
    }
}
RP_DEV void lay_isl_count(DevWorld &w, int gid, int stride) {
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    for (int base = 0; base < w.n_bodies; base += stride) { // wave-uniform trip count (ballots / shuffles inside)
        const int b = base + gid;
        const bool act = b < w.n_bodies && is_dyn(w, b);
        int root = 0, poison = 0;
        if (act) {
            root = uf_find(w.b_label, b);
            __hip_atomic_store(&w.b_label[b], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // a body that carries a joint is solved on the global path (joints live there): poison its component
            // (FrictionModel::Coulomb: the islands are solved by k_island_generic, the twist-only k_island_solve is not launched)
            // ... and so are kinematic bodies (solver bodies with zero inverse mass and their own write-back rule)
            // ... and so is everything in a world with substep solve-groups (additional_solver_iterations: rp_groups.h)
            if (w.b_njoints[b] > 0 || (w.b_flags[b] & RP_BF_TYPE_MASK) != RP_BODY_DYNAMIC || w.n_groups > 1) poison = RP_ISL_NC_MAX + 1;
        }
        // saturating counts: a component that already exceeds the island limits stays on the global path whatever its exact size
        wave_add_saturating(w.r_nb, root, 1, RP_ISL_NB_MAX, act);
        wave_add_saturating(w.r_nc, root, poison, RP_ISL_NC_MAX, act && poison > 0);
    }
    for (int base = 0; base < top; base += stride) {
        const int s = base + gid;
        bool inactive_pair = false, active_pair = false;
        int root = 0;
        if (s < top && w.p_c1[s] >= 0) {
            int b1 = w.c_parent[w.p_c1[s]], b2 = w.c_parent[w.p_c2[s]];
            int b = is_dyn(w, b1) ? b1 : b2;
            if (is_dyn(w, b)) { root = uf_find(w.b_label, b); if (pair_active(w, s)) active_pair = true; else inactive_pair = true; } // inactive: a pair without solver contacts
        }
        wave_add_saturating(w.r_nc, root, 1, RP_ISL_NC_MAX, active_pair);
        if (inactive_pair) atomicAdd(&w.r_ni[root], 1);
    }
}
// number the islands that fit one workgroup (registers + LDS)
// Round 4: k_island_solve gives an island a whole workgroup (512 lanes for up to 145 manifolds) and runs 240 of them at a time — right
// for piles, wasteful for debris: 4,400 islands of one to five bodies (thousands of small shapes on a floor) are 18 passes, 1.35 ms.
// When the previous rebuild counted more candidates than DevWorld::isl_many (one resident pass of k_island_solve: 240 on MI355X), components of at most DevWorld::isl_tiny_nc (8) manifolds stay on the global
// path, whose colour stages / LDS tiles take them all at once (either path gives the same bits: a routing decision, not a result).
// lay_state[3] = candidates of the last rebuild (written behind its last barrier), [4] = this rebuild's count.
//
// Round 5, BUNDLES: in a world whose bodies never sleep the tiny components do not go to the global path either — several of them
// share ONE island of the kernel.  An island is solved colour stage by colour stage in the world's stage order and a constraint only
// touches its own two bodies, so a union of components that share no body gives every one of them the bits it gets alone (the same
// argument as for the routing).  A component of cnb bodies and cnc manifolds weighs max(cnb, ceil(2 cnc / 5)) (an island holds 64
// bodies and 160 manifolds); a running sum of the weights (lay_state[6]) is cut every lay_bundle_span() units, a component belongs to
// the bundle its first unit falls into, which therefore holds at most span - 1 + (isl_tiny_nc + 1) = 64 units.  The totals of a bundle
// are only known once every component has been seen: lay_isl_bundles(), one grid barrier later, numbers the bundles as islands.
// 256 sub-worlds of 18 bodies (rolling capsules: piles of a few bodies + singles): the tile sweeps, their preparation and the
// dataflow ranks leave the step.  Worlds with sleeping keep the routing: the fused step's "may this island fall asleep" test
// (fused_sleep_abort) speaks about ONE persistent island per kernel island.
RP_DEV int lay_bundle_span(const DevWorld &w) { return RP_ISL_NB_MAX - w.isl_tiny_nc; }
RP_DEV bool lay_route_tiny(const DevWorld &w) { return w.isl_route_tiny && w.lay_state[3] > w.isl_many; }
// (lay_state[3] is only written behind the last barrier of a rebuild: every workgroup of a launch gets the same answer)
RP_DEV bool lay_bundling(const DevWorld &w) { return lay_route_tiny(w) && w.isl_bundle_tiny && !w.sleep_enabled && w.isl_tiny_nc >= 1 && w.isl_tiny_nc <= 32; }
RP_DEV void lay_isl_number(DevWorld &w, int gid, int gstride) {
  const bool route_tiny = lay_route_tiny(w);
  const bool bundle = lay_bundling(w);
  for (int b = gid; b < w.n_bodies; b += gstride) {
    if (!is_dyn(w, b) || w.b_label[b] != b) continue;
    int cnb = w.r_nb[b], cnc = w.r_nc[b];
    const bool candidate = cnc > 0 && cnb <= RP_ISL_NB_MAX && cnc <= RP_ISL_NC_MAX;
    if (candidate) atomicAdd(&w.lay_state[4], 1);
    if (candidate && bundle && cnc <= w.isl_tiny_nc && cnb <= w.isl_tiny_nc + 1) {
        const int wc = (2 * cnc + 4) / 5, wgt = cnb > wc ? cnb : wc;
        const int k = atomicAdd(&w.lay_state[6], wgt) / lay_bundle_span(w);
        atomicAdd(&w.bun_nb[k], cnb); atomicAdd(&w.bun_nc[k], cnc); atomicAdd(&w.bun_ni[k], w.r_ni[b]);
        w.r_island[b] = -2 - k; // (lay_isl_fill reads the island id through bun_id)
        continue;
    }
    if (candidate && !(route_tiny && cnc <= w.isl_tiny_nc)) {
        int id = atomicAdd(&w.flags[FL_N_ISLANDS], 1);
        w.isl_body_begin[id] = atomicAdd(&w.flags[FL_ISL_BODY_CURSOR], cnb);
        w.isl_cons_begin[id] = atomicAdd(&w.flags[FL_ISL_CONS_CURSOR], cnc);
        w.isl_nb[id] = cnb; w.isl_nc[id] = cnc; w.isl_fill_b[id] = 0; w.isl_fill_c[id] = 0; w.isl_sorted[id] = 0; w.isl_nstages[id] = 0;
        int cni = w.r_ni[b];
        w.isl_ni[id] = cni; w.isl_fill_i[id] = 0; w.isl_icons_begin[id] = atomicAdd(&w.flags[FL_ISL_ICONS_CURSOR], cni);
        w.r_island[b] = id;
    }
  }
}
// the bundles of tiny components become islands (their totals are complete: one barrier behind lay_isl_number)
RP_DEV void lay_isl_bundles(DevWorld &w, int gid, int gstride) {
    const int span = lay_bundle_span(w), nbun = (w.lay_state[6] + span - 1) / span;
    for (int k = gid; k < nbun; k += gstride) {
        const int cnb = w.bun_nb[k], cnc = w.bun_nc[k], cni = w.bun_ni[k];
        if (cnc == 0) continue;
        const int id = atomicAdd(&w.flags[FL_N_ISLANDS], 1);
        w.isl_body_begin[id] = atomicAdd(&w.flags[FL_ISL_BODY_CURSOR], cnb);
        w.isl_cons_begin[id] = atomicAdd(&w.flags[FL_ISL_CONS_CURSOR], cnc);
        w.isl_nb[id] = cnb; w.isl_nc[id] = cnc; w.isl_fill_b[id] = 0; w.isl_fill_c[id] = 0; w.isl_sorted[id] = 0; w.isl_nstages[id] = 0;
        w.isl_ni[id] = cni; w.isl_fill_i[id] = 0; w.isl_icons_begin[id] = atomicAdd(&w.flags[FL_ISL_ICONS_CURSOR], cni);
        w.bun_id[k] = id;
    }
}
RP_DEV int lay_island_of_root(const DevWorld &w, int root) { const int id = w.r_island[root]; return id <= -2 ? w.bun_id[-2 - id] : id; }
// fill the island lists; count the global-path manifolds per colour
RP_DEV void lay_isl_fill(DevWorld &w, int gid, int stride, int *hist, int &n_glob) {
    for (int c = threadIdx.x; c < RP_NUM_COLORS; c += blockDim.x) hist[c] = 0;
    if (threadIdx.x == 0) n_glob = 0;
    __syncthreads();
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    for (int b = gid; b < w.n_bodies; b += stride) {
        if (!is_dyn(w, b)) continue;
        int id = lay_island_of_root(w, w.b_label[b]);
        if (id >= 0) {
            int k = atomicAdd(&w.isl_fill_b[id], 1);
            w.isl_bodies[w.isl_body_begin[id] + k] = b;
            w.b_island[b] = id; w.b_local[b] = k;
        } else atomicAdd(&n_glob, 1); // dynamic bodies left to the global path (launch-plan hint)
    }
    for (int s = gid; s < top; s += stride) {
        if (w.p_c1[s] < 0) { w.p_island[s] = -1; continue; }
        int b1 = w.c_parent[w.p_c1[s]], b2 = w.c_parent[w.p_c2[s]];
        int b = is_dyn(w, b1) ? b1 : b2;
        int id = is_dyn(w, b) ? lay_island_of_root(w, w.b_label[b]) : -1;
        if (!pair_active(w, s)) { // owned by the island of its first dynamic body (recycle-tested there)
            w.p_island[s] = -1;
            if (id >= 0) { int k = atomicAdd(&w.isl_fill_i[id], 1); w.isl_icons[w.isl_icons_begin[id] + k] = s; }
            continue;
        }
        w.p_island[s] = id;
        if (id >= 0) { int k = atomicAdd(&w.isl_fill_c[id], 1); w.isl_cons[w.isl_cons_begin[id] + k] = s; }
        else {
            int color = w.p_color[s];
            if (color <= RP_COLOR_OVERFLOW) atomicAdd(&hist[color], 1);
            if (color < RP_COLOR_OVERFLOW) { // owner = the first awake dynamic body: at most one manifold per colour names it
                int owner = body_dyn_awake(w, b1) ? b1 : b2;
                if (w.b_order) owner = w.b_order[owner]; // worlds that tile: the owners' Morton order (rp_tiles.hip), any injective order will do
                atomicOr(&w.cb_bits[(size_t)color * w.cb_words + (owner >> 5)], 1u << (owner & 31));
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < RP_NUM_COLORS; c += blockDim.x) if (hist[c]) atomicAdd(&w.color_count_glob[c], hist[c]);
    if (threadIdx.x == 0 && n_glob) atomicAdd(&w.flags[FL_N_GLOB_BODIES], n_glob);
}


// The same pass for a world that holds NO island: every pair belongs to the global path (k_layout_rebuild's all-global form)
RP_DEV void lay_glob_fill(DevWorld &w, int gid, int stride, int *hist) {
    for (int c = threadIdx.x; c < RP_NUM_COLORS; c += blockDim.x) hist[c] = 0;
    __syncthreads();
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    for (int s = gid; s < top; s += stride) {
        w.p_island[s] = -1;
        if (w.p_c1[s] < 0 || !pair_active(w, s)) continue;
        const int b1 = w.c_parent[w.p_c1[s]], b2 = w.c_parent[w.p_c2[s]];
        const int color = w.p_color[s];
        if (color <= RP_COLOR_OVERFLOW) atomicAdd(&hist[color], 1);
        if (color < RP_COLOR_OVERFLOW) {
            int owner = body_dyn_awake(w, b1) ? b1 : b2;
            if (w.b_order) owner = w.b_order[owner];
            atomicOr(&w.cb_bits[(size_t)color * w.cb_words + (owner >> 5)], 1u << (owner & 31));
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < RP_NUM_COLORS; c += blockDim.x) if (hist[c]) atomicAdd(&w.color_count_glob[c], hist[c]);
}

// ---- solver contact graph buckets + stage layout (were rp_narrowphase.hip kernels) ----
RP_DEV void lay_bucket_clear(DevWorld &w) { // workgroup 0
    if (threadIdx.x < RP_NUM_COLORS) { w.color_count[threadIdx.x] = 0; w.color_count_glob[threadIdx.x] = 0; }
    if (threadIdx.x == 0) w.flags[FL_N_SC] = 0;
}
RP_DEV void lay_bucket_count(DevWorld &w, int gid, int stride, int *hist, int &nsc_sum) {
    for (int c = threadIdx.x; c < RP_NUM_COLORS; c += blockDim.x) hist[c] = 0;
    if (threadIdx.x == 0) nsc_sum = 0;
    __syncthreads();
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    for (int s = gid; s < top; s += stride) {
        w.p_conspos[s] = -1;
        if (!pair_selected(w, s)) continue;
        int color = w.p_color[s];
        if (color > RP_COLOR_OVERFLOW) continue;
        atomicAdd(&hist[color], 1);
        atomicAdd(&nsc_sum, w.p_nsc[s]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < RP_NUM_COLORS; c += blockDim.x) if (hist[c]) atomicAdd(&w.color_count[c], hist[c]);
    if (threadIdx.x == 0 && nsc_sum) atomicAdd(&w.flags[FL_N_SC], nsc_sum);
}
// per colour, the exclusive prefix popcount of the owner bitmap along its words (rank of a body among the colour's owners): one
// WAVEFRONT per colour over the whole launch (shuffles, no workgroup barrier) — the 128 colours used to be scanned one after the
// other by workgroup 0 with 20 workgroup barriers each (~15 us per colour in use while every other workgroup waited)
RP_DEV void lay_owner_prefix(DevWorld &w) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const int words = w.cb_words, per = (words + 63) / 64, lo = lane * per, hi = lo + per < words ? lo + per : words;
    for (int c = wave; c < RP_COLOR_OVERFLOW; c += nwaves) {
        if (w.color_count_glob[c] == 0) continue; // wave-uniform
        const unsigned *bits = w.cb_bits + (size_t)c * words;
        int *pre = w.cb_prefix + (size_t)c * words;
        int sum = 0;
        for (int i = lo; i < hi; ++i) sum += __popc(bits[i]);
        int incl = sum;
        for (int off = 1; off < 64; off <<= 1) { int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
        int run = incl - sum;
        for (int i = lo; i < hi; ++i) { pre[i] = run; run += __popc(bits[i]); }
    }
}
RP_DEV void lay_bucket_layout(DevWorld &w) { // workgroup 0: the stage order (serial, a few hundred scalar operations)
    if (threadIdx.x != 0) return;
    int nst = 0, npar = 0, pos = 0, maxs = 0, ncol = 0, mall = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int c = 0; c < RP_NUM_COLORS - 1; ++c) {
            int n = w.color_count[c];
            if (n == 0) continue;
            bool par = n >= RP_PARALLEL_MIN_MANIFOLDS;
            if ((pass == 0) != par) continue;
            int ng = w.color_count_glob[c];
            w.stage_color[nst] = c; w.stage_begin[nst] = pos; w.stage_count[nst] = ng;
            w.color_begin[c] = pos; w.color_cursor[c] = pos; w.color_rank[c] = nst;
            pos += ng; nst++; ncol++; mall += n;
            if (par) npar++;
            if (ng > maxs) maxs = ng;
        }
    int nov = w.color_count[RP_COLOR_OVERFLOW], novg = w.color_count_glob[RP_COLOR_OVERFLOW];
    w.stage_color[nst] = RP_COLOR_OVERFLOW; w.stage_begin[nst] = pos; w.stage_count[nst] = novg;
    w.color_begin[RP_COLOR_OVERFLOW] = pos; w.color_cursor[RP_COLOR_OVERFLOW] = pos; w.color_rank[RP_COLOR_OVERFLOW] = nst;
    pos += novg; mall += nov;
    if (nov) ncol++;
    w.flags[FL_N_STAGES] = nst; w.flags[FL_N_PARALLEL] = npar; w.flags[FL_MAX_STAGE] = maxs;
    w.flags[FL_HAS_OVERFLOW_COLOR] = novg > 0; w.flags[FL_N_COLORS] = ncol;
    w.flags[FL_N_CONS] = pos; w.flags[FL_N_CONS_ALL] = mall;
    w.flags[FL_FLOW_DIRTY] = 1; // constraint positions are about to move: the dataflow solver's toucher ranks follow (rp_flow.hip)
    if (pos > w.cons_cap) atomicOr(&w.flags[FL_OVERFLOW], RP_OVF_CONS);
}
RP_DEV void lay_bucket_scatter(DevWorld &w, int gid, int stride, int *cnt, int *base) {
    // two passes per block: count its manifolds per colour, reserve one range per colour, then place
    for (int c = threadIdx.x; c < RP_NUM_COLORS; c += blockDim.x) cnt[c] = 0;
    __syncthreads();
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    // colour stages: position = rank of the owner body among the colour's owners (no atomics, ascending with the body index);
    // the overflow colour (not body-disjoint) keeps its reserve-and-place scheme and is ranked by the closing workgroup below
    for (int s = gid; s < top; s += stride) {
        if (!pair_selected(w, s) || w.p_island[s] >= 0) continue;
        int color = w.p_color[s];
        if (color == RP_COLOR_OVERFLOW) atomicAdd(&cnt[color], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) { base[RP_COLOR_OVERFLOW] = cnt[RP_COLOR_OVERFLOW] ? atomicAdd(&w.color_cursor[RP_COLOR_OVERFLOW], cnt[RP_COLOR_OVERFLOW]) : 0; cnt[RP_COLOR_OVERFLOW] = 0; }
    __syncthreads();
    for (int s = gid; s < top; s += stride) {
        if (!pair_selected(w, s) || w.p_island[s] >= 0) continue;
        int color = w.p_color[s];
        if (color > RP_COLOR_OVERFLOW) continue;
        int pos;
        if (color == RP_COLOR_OVERFLOW) pos = base[color] + atomicAdd(&cnt[color], 1);
        else {
            int2 rb = w.p_rb[s];
            int owner = body_dyn_awake(w, rb.x) ? rb.x : rb.y;
            if (w.b_order) owner = w.b_order[owner];
            size_t wi = (size_t)color * w.cb_words + (owner >> 5);
            pos = w.color_begin[color] + w.cb_prefix[wi] + __popc(w.cb_bits[wi] & ((1u << (owner & 31)) - 1u));
        }
        if (pos < w.cons_cap) { w.cons_pair[pos] = s; w.p_conspos[s] = pos; }
    }
}
RP_DEV void lay_rank_overflow(DevWorld &w) { // workgroup 0, after a grid barrier
    // The overflow colour is swept serially and is not body-disjoint: its order is part of the result, and the scatter above is
    // ordered by atomics.  The closing workgroup ranks it by (collider1, collider2) — the order the oracle uses.
    const int nst = ld_i32(&w.flags[FL_N_STAGES]);
    const int ob = ld_i32(&w.stage_begin[nst]), on = ld_i32(&w.flags[FL_HAS_OVERFLOW_COLOR]) ? ld_i32(&w.stage_count[nst]) : 0;
    if (on > 1 && ob + on <= w.cons_cap) {
        for (int i = threadIdx.x; i < on; i += blockDim.x) w.todo_tmp[i] = ld_i32(&w.cons_pair[ob + i]);
        __threadfence(); __syncthreads();
        for (int i = threadIdx.x; i < on; i += blockDim.x) {
            int si = ld_i32(&w.todo_tmp[i]);
            unsigned long long ki = pair_order_key(w, si);
            int rank = 0;
            for (int j = 0; j < on; ++j) { int sj = ld_i32(&w.todo_tmp[j]); unsigned long long kj = pair_order_key(w, sj); rank += kj < ki; }
            w.cons_pair[ob + rank] = si; w.p_conspos[si] = ob + rank;
        }
        __threadfence(); __syncthreads();
    }
    // Who may sweep the overflow colour in parallel (tail_sweep, rp_global.h): a manifold with ONE dynamic side — its other side a
    // kinematic body whose 120 colours ran out: b3d_washer's ring holds thousands of contacts — only shares a side nobody writes, so
    // manifolds of different owners commute bit for bit.  ov_owner[i] = that dynamic body; lay_state[5] = every overflow manifold has one
    if (threadIdx.x == 0) w.lay_state[5] = 0; // (no overflow manifold, or more than the rows hold: never the owner-parallel sweep on stale owners)
    __threadfence(); __syncthreads();
    if (on >= 1 && ob + on <= w.cons_cap) {
        if (threadIdx.x == 0) w.lay_state[5] = 1;
        __threadfence(); __syncthreads();
        for (int i = threadIdx.x; i < on; i += blockDim.x) {
            const int2 rb = w.p_rb[ld_i32(&w.cons_pair[ob + i])];
            const bool d1 = rb.x >= 0 && (w.b_flags[rb.x] & RP_BF_TYPE_MASK) == RP_BODY_DYNAMIC, d2 = rb.y >= 0 && (w.b_flags[rb.y] & RP_BF_TYPE_MASK) == RP_BODY_DYNAMIC;
            w.ov_owner[i] = d1 && d2 ? -1 : (d1 ? rb.x : (d2 ? rb.y : 0));
            if (d1 && d2) w.lay_state[5] = 0; // (two dynamic sides: the serial order is part of the result)
        }
        __threadfence(); __syncthreads();
    }
}

// The whole layout rebuild in ONE launch (rp_gridbar.h): colour buckets (maintain_solver_contact_graph, solver_graph.rs:129-361),
// contact islands, the stage order of init.rs:163-254 and the constraint positions.  A step whose active-manifold set did not
// change pays a single early exit instead of nine.
__global__ void __launch_bounds__(1024) k_layout_rebuild(DevWorld w) {
    if (!w.flags[FL_LAYOUT_DIRTY]) return; // (cleared only after the last barrier: every workgroup reads the same value)
    __shared__ int lds_a[1024], lds_b[RP_NUM_COLORS], lds_scalar, lds_uf[LAY_LDS_BODIES];
    const int gid = gbar_item(), gstride = gridDim.x * blockDim.x;
    GridBar bar = gbar_begin(w, 1);
    RP_PASS_BEGIN();
    const bool warm = lay_warm(w); // (lay_state is only written behind the last barrier of a launch, or by an edit between steps: every workgroup reads the same)
    // ALL-GLOBAL form (round 6).  A settling pile rebuilds its layout on a third of its steps (b3d_large_pyramid: ~300 of the first 1,060)
    // and each time the island passes — edges, union-find, counts, numbering, fill: five of the eight grid barriers and 140 of the 200 us —
    // find what the warm start already assumes: one giant component, no island.  While the labels are warm and the last rebuild left no
    // island (lay_state[7]), a body that leaves the pile stays on the global path anyway (lay_warm: a routing decision, never a result), so
    // only the colour buckets, the stage order and the constraint positions are rebuilt; every 16th rebuild is cold and looks again.
    if (warm && w.lay_state[7] == 1) {
        if (blockIdx.x == 0) lay_bucket_clear(w);
        for (size_t k = gid, n = (size_t)128 * w.cb_words; k < n; k += (size_t)gstride) w.cb_bits[k] = 0u;
        GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
        lay_bucket_count(w, gid, gstride, lds_a, lds_scalar);
        __syncthreads();
        lay_glob_fill(w, gid, gstride, lds_a);
        GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
        lay_owner_prefix(w);
        if (blockIdx.x == 0) lay_bucket_layout(w);
        GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
        lay_bucket_scatter(w, gid, gstride, lds_a, lds_b);
        GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
        if (ld_i32(&w.flags[FL_HAS_OVERFLOW_COLOR])) {
            if (blockIdx.x == 0) lay_rank_overflow(w);
            GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
        }
        gbar_end(bar);
        if (gid == 0) { w.lay_state[1] += 1; __hip_atomic_store(&w.flags[FL_LAYOUT_DIRTY], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        return;
    }
    if (blockIdx.x == 0) lay_bucket_clear(w);
    lay_isl_init(w, gid, gstride, warm);
    lay_isl_edges(w, gid, gstride);
    GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
    lay_bucket_count(w, gid, gstride, lds_a, lds_scalar);
#ifdef RP_PASS_PROFILE
    GBAR_SYNC(bar); RP_PASS_STAMP(w, 200); // (profiling only: the bucket count apart from the union)
#endif
    if (w.n_bodies <= LAY_LDS_BODIES) { if (blockIdx.x == 0) lay_isl_union_lds(w, lds_uf, warm); }
    else lay_isl_union(w, gid, gstride);
    GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
    lay_isl_count(w, gid, gstride);
    GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
    lay_isl_number(w, gid, gstride);
    GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
    // (a rebuild that bundles pays one more barrier; every workgroup takes the same branch: see lay_bundling)
    if (lay_bundling(w)) { lay_isl_bundles(w, gid, gstride); GBAR_SYNC(bar); RP_PASS_STAMP(w, 200); }
    __syncthreads();
    lay_isl_fill(w, gid, gstride, lds_a, lds_scalar);
    GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
    lay_owner_prefix(w);
    if (blockIdx.x == 0) lay_bucket_layout(w);
    GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
    lay_bucket_scatter(w, gid, gstride, lds_a, lds_b);
    GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
    // (the overflow colour is ranked by workgroup 0 behind one more barrier — which a world without a global-path manifold in that
    // colour does not pay: FL_HAS_OVERFLOW_COLOR was written two barriers ago, every workgroup reads the same value)
    if (ld_i32(&w.flags[FL_HAS_OVERFLOW_COLOR])) {
        if (blockIdx.x == 0) lay_rank_overflow(w);
        GBAR_SYNC(bar); RP_PASS_STAMP(w, 200);
    }
    gbar_end(bar);
    if (gid == 0) {
        // The routing of tiny islands went by the candidates of the LAST rebuild; a world that settles at once (debris dropped on a
        // floor: one rebuild, then nothing changes any more) would keep thousands of one-manifold islands in workgroups of their own
        // for good.  When this rebuild's count answers the routing question differently, the layout stays dirty: the next step
        // rebuilds it with the right answer (a routing decision, never a result; the host keeps the full graph while it is dirty).
        const int again = (w.isl_route_tiny && (w.lay_state[4] > w.isl_many) != (w.lay_state[3] > w.isl_many)) ? 1 : 0;
        w.flags[FL_UF_NPAIRS] = 0; w.lay_state[0] = 1; w.lay_state[1] += 1; w.lay_state[2] = w.flags[FL_N_GLOB_BODIES]; w.lay_state[3] = w.lay_state[4];
        w.lay_state[7] = w.flags[FL_N_ISLANDS] == 0 ? 1 : 0; // (no island, no bundle: the next warm rebuilds take the all-global form)
        __hip_atomic_store(&w.flags[FL_LAYOUT_DIRTY], again, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// One workgroup = one island; lanes 2m, 2m+1 = manifold m (sorted by sweep stage), threads < nb also own a body.
//
// `fused` (steady-state fast graph whose ONLY kernel this is): the launch first proves that this step
// needs neither the broad phase nor the narrow phase — every workgroup checks the fat AABBs of its
// islands' bodies and the recycle tests of its islands' pairs (pair_update.rs:111-171) against the
// start-of-step poses, raises FL_FAST_ABORT on a failure and then arrives on FL_ARRIVE.  The solve
// runs meanwhile; nothing is written back until every workgroup has arrived (~70 us later: the wait is
// free) and no abort was raised.  An aborted launch leaves the world untouched and the host replays the
// step on the full graph.  Needs every workgroup resident at once: the grid is capped below the CU count.
// This form: 512 threads, every warm-start term at once (57 KB of W), 256 VGPRs: ONE island per CU — the fastest form per island, chosen
//   while the islands fit one pass of the resident grid (b3d_many_pyramids: 196 islands on 256 CUs).
// The register-lean form (rp_islands_lean.h, k_island_solve_dense): 320 threads, 168 VGPRs, two islands per CU — chosen when there
//   are more islands than the resident grid of this form holds (b3d_many_pyramids at C4's density: 365 islands per GPU, 2,916 on one).
// `nsteps` (fused launches only; round 6): SEVERAL steps in one launch.  A step boundary inside the launch is a workgroup barrier, not a
// kernel boundary: what a step wrote back (bodies, impulses) is read by the next one from this CU's own caches instead of from a cold L2
// — every launch used to open with all workgroups pulling ~100 KB each through the fabric at once — and every step keeps the fused
// step's protocol (validate, arrive, solve, commit only when every workgroup arrived and none aborted): FL_ARRIVE counts (step + 1) x
// grid arrivals, the first aborted step ends the launch for every workgroup at the same step, FL_STEP advances by the steps committed
// and the host replays the rest on the full graph.  Needs one island per workgroup at most, and islands whose pairs name no body of
// ANOTHER island (a workgroup may read another one's bodies only across a kernel boundary): otherwise the launch runs one step.
template <bool WIDE, bool COUL = false> __device__ __forceinline__ void island_solve_body(const DevWorld &w, int has_restitution, int fast, int retire, int fused, int nsteps = 1) {
    constexpr int THREADS = ISL_THREADS;
    // `fused` bit 1 / `nsteps` bit 16 (the host sets them while aborts are recent, step_once): ask for the verdict on a step EARLY — a lane
    // without work polls FL_ARRIVE in substep 0's pose stage, and a step some workgroup has aborted ends at the top of substep 1 instead of
    // behind its last relaxed sweep: a doomed step then costs a third of a step (a shard of C4 aborts every third step: its pyramids creep)
    const bool early_poll = ((fused & 2) != 0) || ((nsteps & (1 << 16)) != 0);
    fused &= 1; nsteps &= 0xffff;
    const bool aborted = (fast && w.flags[FL_FAST_ABORT]) || lean_dead(w); // fast graph gave up on this step (rp_api.hip) / dead lean step (rp_world.h)
    if (retire && blockIdx.x == 0) {
        // SINGLE mode: workgroup 0 retires the step and publishes the scalars to the host hint buffer
        // up front, so the PCIe writes overlap the solve instead of ending the step
        if (threadIdx.x == 0) { w.flags[FL_SEQ] += (fused && nsteps > 1) ? nsteps : 1; if (!aborted && !fused) w.flags[FL_STEP] += 1; if (fused) w.flags[FL_FULL_UPDATES] = 0; }
        __threadfence(); __syncthreads();
        publish_flags(w);
    }
    if (aborted) return;
    __shared__ int s_abort, s_go, s_slp, s_cross, s_early;
    if (threadIdx.x == 0) { s_cross = 0; s_early = 0; }
#ifdef RP_ISL_PROFILE
    long long t_fused0 = (long long)__builtin_readcyclecounter();
#endif
    if (fused) {
        // validation of this workgroup's islands BEYOND its first one (the first is validated under its own
        // load phase below, where the latency of these loads hides behind the island's loads); FL_ARRIVE
        // counts arrivals in its low 16 bits and aborting workgroups above: one atomic, no fence needed
        const int t = threadIdx.x;
        const int n_islands = w.flags[FL_N_ISLANDS];
        if (t == 0) {
            s_abort = 0; s_slp = 0;
            if (blockIdx.x == 0 && fused_world_abort<WIDE>(w)) s_abort = 1; // block 0: the conditions k_fast_front checks for the whole world
            if constexpr (WIDE) if (w.sleep_enabled) sleep_begin_scan(w); // ONE stamp bump per workgroup (every observing lane trying it: thousands of CAS on one word)
        }
        __syncthreads();
        bool bad = false;
        const int stamp_before = (WIDE && w.sleep_enabled) ? pi_stamp_before(w) : 0;
        for (int isl = blockIdx.x + gridDim.x; isl < n_islands; isl += gridDim.x) {
            int slp = 0;
            if (fused_validate_island<WIDE>(w, isl, t, blockDim.x, stamp_before, slp)) bad = true;
            if constexpr (WIDE) if (w.sleep_enabled && fused_sleep_abort(__syncthreads_or(slp))) bad = true;
        }
        if (bad) s_abort = 1;
        if ((int)blockIdx.x >= n_islands && !(nsteps > 1 && n_islands <= (int)gridDim.x)) { // no island at all: arrive now (a launch of several steps: once per step, below)
            __syncthreads();
            if (t == 0) atomicAdd(&w.flags[FL_ARRIVE], 1 + (s_abort ? (1 << 16) : 0));
        }
#ifdef RP_ISL_PROFILE
        if (blockIdx.x == 0 && threadIdx.x == 0) w.dbg[10] += (long long)__builtin_readcyclecounter() - t_fused0;
#endif
    }
    bool decided = !fused, go = true;
    __shared__ float4 B_lin[RP_ISL_NB_MAX], B_ang[RP_ISL_NB_MAX], B_rot[RP_ISL_NB_MAX], B_trans[RP_ISL_NB_MAX], B_axes[RP_ISL_NB_MAX];
    __shared__ float4 L_E[4 * RP_ISL_NC_MAX], L_F[4 * RP_ISL_NC_MAX], L_B0[COUL ? 1 : RP_ISL_NC_MAX], L_B1[COUL ? 1 : RP_ISL_NC_MAX]; // (the Coulomb model has no friction centre)
    __shared__ int S_a[RP_ISL_NC_MAX], S_b[RP_ISL_NC_MAX], S_c[RP_ISL_NC_MAX], S_d[RP_ISL_NC_MAX];
    __shared__ float4 W[(COUL ? WS_SLOTS_COUL : WS_SLOTS) * WS_STRIDE];
    __shared__ int any_bouncy;

    const int t = threadIdx.x, m = t >> 1;
    const bool odd = (t & 1) != 0;
    const int n_islands = w.flags[FL_N_ISLANDS];
    const int nst_global = w.flags[FL_N_STAGES];
    const rp_integration_params &prm = w.prm.p;
    const bool fib = prm.friction_in_bias_pass || prm.num_internal_stabilization_iterations == 0;
    IslLds L;
    L.lin = B_lin; L.ang = B_ang; L.rot = B_rot; L.trans = B_trans; L.E = L_E; L.F = L_F; L.B0 = L_B0; L.B1 = L_B1;

    const int ns = (fused && nsteps > 1 && n_islands <= (int)gridDim.x) ? nsteps : 1; // (more islands than workgroups: one step, islands in rounds)
    // what a workgroup knows about its island once and for all steps of a launch (loaded by the first step; a launch of one step visits
    // its islands in rounds and loads them per island): the island's lists, each lane's manifold, each body thread's constants
    constexpr int ROLE_LIN0 = ISL_LANES, ROLE_ANG0 = ISL_LANES + RP_ISL_NB_MAX;
    static_assert(ROLE_ANG0 + RP_ISL_NB_MAX <= THREADS, "the body roles need two wavefronts beyond the manifold lanes");
    const int bt = t & (RP_ISL_NB_MAX - 1);
    int nb = 0, nc = 0, bb = 0, cb = 0, nls = 0, v_first = 0;
    bool role_lin = false, role_ang = false, live = false, pair_static = false;
    int b_gid = -1, b_fl = 0, slot = -1, myq = -1, own_g = -1, own_l = -1, ws_row = 0, inc_begin = 0, inc_cnt = 0;
    V3 b_inc = v3(0, 0, 0) /* the linear increment on a body's linear thread, the angular one on its angular thread */, b_invpi = b_inc, b_pi = b_inc;
    V3 b_aux = b_inc; // (launches of several steps) linear thread: the local centre of mass; angular thread: the user torque — what the per-step reload needs besides the state
    Q4 b_pframe = q4(0, 0, 0, 1);
    for (int step = 0; step < ns; ++step) {
    // FL_ARRIVE: arrivals in the low 16 bits (cumulative over the steps of the launch), aborting workgroups above them in TWO 8-bit fields
    // by the parity of the step they abort: workgroups are at most one step apart, and an abort raised by a faster one for step s + 1 must
    // not undo step s for a slower one that is still waiting for the verdict on s (every workgroup commits a step or none does)
    const int arrive_target = (step + 1) * (int)gridDim.x, ab_shift = 16 + 8 * (step & 1);
    const unsigned ab_one = 1u << ab_shift;
    if (step > 0) { // the next step of this launch: fresh verdicts (the barrier at the end of the last step ordered its write-back before this)
        decided = false;
        if (t == 0) { s_abort = s_cross; s_slp = 0; s_early = 0; } // (an island with a pair into another island takes no second step: see fused_validate_island)
        __syncthreads();
    }
    if (ns > 1 && (int)blockIdx.x >= n_islands) { // a workgroup without an island follows the protocol of every step: arrive, wait for the verdict
        if (t == 0) {
            atomicAdd((unsigned *)&w.flags[FL_ARRIVE], 1u + (s_abort ? ab_one : 0u));
            unsigned v; int spins = 0;
            while (((v = __hip_atomic_load((unsigned *)&w.flags[FL_ARRIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffffu) < (unsigned)arrive_target && ((v >> ab_shift) & 0xffu) == 0 && ++spins < (1 << 23)) __builtin_amdgcn_s_sleep(8);
            s_go = (v & 0xffffu) >= (unsigned)arrive_target && ((v >> ab_shift) & 0xffu) == 0; // (a timeout is raised by the workgroups that hold islands)
        }
        __syncthreads();
        go = s_go != 0;
    }
    for (int isl = blockIdx.x; isl < n_islands; isl += gridDim.x) {
        if (step == 0) { nb = w.isl_nb[isl]; nc = w.isl_nc[isl]; bb = w.isl_body_begin[isl]; cb = w.isl_cons_begin[isl]; }
        __syncthreads(); // previous island of this workgroup fully written back
#ifdef RP_ISL_PROFILE
        long long t_prev = (long long)__builtin_readcyclecounter();
#endif
        if (step == 0 && !w.isl_sorted[isl]) island_sort(w, isl, nc, cb, nst_global, S_a, S_b, S_c, S_d);
        // ---- bodies -> LDS (+ per-body constants in the owning thread's registers) (S0) ----
        // The body roles live in the wavefronts that hold NO manifold lane (round 6): threads [320, 320 + nb) own the linear half of
        // body t - 320 (and integrate / write it back), threads [384, 384 + nb) the angular half: the two halves of the increment and of
        // the body-centric warm start are independent chains, and on wavefronts of their own the increment (the gyroscopic term is a
        // ~200-instruction dependent chain) runs WHILE the manifold lanes write their warm-start terms instead of behind them.
        if (step == 0) {
            v_first = (2 * nc + 63) & ~63; // first wavefront without any manifold lane
            role_lin = t >= ROLE_LIN0 && t < ROLE_LIN0 + nb; role_ang = t >= ROLE_ANG0 && t < ROLE_ANG0 + nb;
            b_gid = -1; b_fl = 0;
            if (role_lin || role_ang) {
                int g = w.isl_bodies[bb + bt];
                V3 lin, ang, trans; Q4 rot;
                V3 incl, inca;
                body_begin(w, g, lin, ang, rot, trans, incl, inca);
                b_gid = g; b_fl = w.b_flags[g];
                if (role_lin) { B_lin[bt] = f4(lin, 0.0f); B_rot[bt] = f4(rot); B_trans[bt] = f4(trans, 0.0f); b_inc = incl; }
                else { B_ang[bt] = f4(ang, 0.0f); b_inc = inca; }
                b_invpi = v3(w.b_invpi[g]); b_pframe = q4(w.b_pframe[g]);
                if (ns > 1) b_aux = role_lin ? v3(w.b_lcom_invm[g]) : v3(w.b_utorque[g]);
                // what body_increment's gyroscopic term derives from constants / from the pose alone, once instead of on every substep's
                // critical path: the principal inertia (three IEEE divisions) and the principal axes in world space (updated behind integrate)
                if (role_ang && (b_fl & RP_BF_GYRO)) { b_pi = v3(rp_inv(b_invpi.x), rp_inv(b_invpi.y), rp_inv(b_invpi.z)); B_axes[bt] = f4(qmul(rot, b_pframe)); } // (each thread reads back only what it stored itself)
            }
            nls = w.isl_nstages[isl];
            live = m < nc;
            slot = -1; myq = -1; own_g = -1; own_l = -1; ws_row = 0; pair_static = false;
            if (live) {
                slot = w.isl_cons[cb + m]; myq = w.isl_cstage[cb + m];
                const int l1 = w.isl_cl1[cb + m], l2 = w.isl_cl2[cb + m];
                own_g = (odd ? w.isl_cg2 : w.isl_cg1)[cb + m]; own_l = odd ? l2 : l1;
                pair_static = l1 < 0 || l2 < 0;
                ws_row = w.isl_inc_pos[2 * cb + t];
            }
            inc_begin = 0; inc_cnt = 0;
            if (role_lin || role_ang) { inc_begin = w.isl_inc_begin[bb + bt]; inc_cnt = w.isl_inc_cnt[bb + bt]; }
        } else if (role_lin) {
            // a further step of the launch: the state the last step wrote back (by this very thread; the barrier behind the write-back made
            // it visible), through body_begin's own expressions — what does not change within a launch stayed in registers
            const V3 lin = v3(w.b_linvel[b_gid]); const Q4 rot = q4(w.b_rot[b_gid]);
            const V3 trans = qrot(rot, b_aux) + v3(w.b_pos[b_gid]);
            B_lin[bt] = f4(lin, 0.0f); B_rot[bt] = f4(rot); B_trans[bt] = f4(trans, 0.0f);
        } else if (role_ang) {
            const V3 ang = v3(w.b_angvel[b_gid]);
            const Sym3 ii = load_ii(w, b_gid);
            b_inc = sym_mul(ii, b_aux) * w.prm.dt_sub; // (inca: the world inverse inertia turned with the body)
            B_ang[bt] = f4(ang, 0.0f);
            if (b_fl & RP_BF_GYRO) B_axes[bt] = f4(qmul(q4(w.b_rot[b_gid]), b_pframe));
        }
        if (t == 0) any_bouncy = 0;
        __syncthreads();
        ISL_STAMP(0); // body load + list
        typename std::conditional<COUL, IslSideC, IslSide>::type h; // the lane's half of its manifold: rp_lanepair.h (twist model) / rp_coulomb_pair.h
        h.n = 0; h.id = -1; h.odd = odd; h.cids = 0;
        // ---- generate (S1) by the lane pair, pose stage for the initial poses ----
        if (live) {
            if (isl_generate(w, h, L, m, slot, own_g, own_l, odd, pair_static) && !odd) any_bouncy = 1;
        }
        if (fused && isl == (int)blockIdx.x && t >= v_first) {
            // the wavefronts without a manifold prove, under cover of generate (the longest interval of the
            // kernel), that this island needs neither broad nor narrow phase this step: one item (a body's
            // collider, an active pair, a pair without solver contacts) per lane and round
#ifdef RP_ISL_PROFILE
            long long tv0 = (long long)__builtin_readcyclecounter();
#endif
            int slp = 0;
#ifndef RP_ISL_NOVAL
            bool cross = false;
            if (fused_validate_island<WIDE>(w, isl, t - v_first, THREADS - v_first, (WIDE && w.sleep_enabled) ? pi_stamp_before(w) : 0, slp, 0, 0, ns > 1 ? &cross : nullptr)) s_abort = 1;
            if (cross) { s_cross = 1; if (w.flags[FL_GRID_TIMEOUT] == 0) w.flags[FL_GRID_TIMEOUT] = 2; } // (2: "no launches of several steps for this world", read by settle())
#endif
            if constexpr (WIDE) if (slp) atomicOr(&s_slp, slp);
#ifdef RP_ISL_PROFILE
            if (blockIdx.x == 0 && (t == v_first || t == THREADS - 1)) w.dbg[t == v_first ? 13 : 14] += (long long)__builtin_readcyclecounter() - tv0;
#endif
        }
        if (live) isl_pose_stage(w, h, L, m, 0.0f); // each lane reads back only what it stored itself
        ISL_STAMP(1); // generate + first pose stage

        for (int sub = 0; sub < w.prm.num_substeps; ++sub) {
            float solved_dt = (float)sub * w.prm.dt_sub;
            // S2 increment (worker.rs:235-284) by the body threads, side by side with the warm-start terms of every manifold ...
            V3 inc_v = v3(0, 0, 0);
            if (live) { if constexpr (COUL) isl_ws_terms_coul(w, h, W, ws_row); else isl_ws_terms<true>(w, h, W, ws_row); }
            else if (role_lin) inc_v = v3(B_lin[bt]) + b_inc;
            else if (role_ang) { // (body_increment's angular half: ang + inca, then gyroscopic_corrected_angvel with the hoisted operands)
                inc_v = v3(B_ang[bt]) + b_inc;
                if (b_fl & RP_BF_GYRO) inc_v = gyro_corrected(inc_v, q4(B_axes[bt]), b_pi, b_invpi, w.prm.dt_sub);
            }
            __syncthreads(); // + pose stage read rot/trans; relax sweep of the previous substep done
            if (early_poll && sub == 1 && !decided && s_early == 2) { go = false; decided = true; break; } // (written in substep 0's pose stage, barriers ago; uniform)
            if (fused && sub == 0 && isl == (int)blockIdx.x && t == 0) atomicAdd((unsigned *)&w.flags[FL_ARRIVE], 1u + ((s_abort || (WIDE && fused_sleep_abort(s_slp))) ? ab_one : 0u)); // this workgroup validated all of its islands (for this step)
            ISL_STAMP(sub == 0 ? 12 : 2); // warm-start terms + increment (substep 0: + whatever the validating wavefronts still have to do)
            // ... then the warm start of this body in sweep order
            if (role_lin) {
                if (prm.warmstart_coefficient != 0.0f) { if constexpr (COUL) isl_ws_accumulate_lin_coul(W, inc_begin, inc_cnt, inc_v); else isl_ws_accumulate_lin_dense(W, inc_begin, inc_cnt, inc_v); }
                B_lin[bt] = f4(inc_v, 0.0f);
            } else if (role_ang) {
                if (prm.warmstart_coefficient != 0.0f) { if constexpr (COUL) isl_ws_accumulate_ang_coul(W, inc_begin, inc_cnt, inc_v); else isl_ws_accumulate_ang_dense(W, inc_begin, inc_cnt, inc_v); }
                B_ang[bt] = f4(inc_v, 0.0f);
            }
            __syncthreads();
            ISL_STAMP(3); // increment + body-centric warm start
#ifdef RP_ISL_EXTRA_EMPTY // overhead measurement only: one extra sweep of empty stages (velocity read/write + barrier)
            for (int q = 0; q < nls; ++q) { if (myq == q) { Vel v = isl_vel(L, h.id); isl_set_vel(L, h.id, v); } __syncthreads(); }
            ISL_STAMP(9);
#endif
            for (int it = 0; it < prm.num_internal_pgs_iterations; ++it)
                for (int q = 0; q < nls; ++q) { if (myq == q) isl_solve(h, L, false, fib); __syncthreads(); }
            ISL_STAMP(4); // biased sweep
            if (role_lin) { // S6
                V3 lin = v3(B_lin[bt]), ang = v3(B_ang[bt]), trans = v3(B_trans[bt]); Q4 rot = q4(B_rot[bt]);
                body_integrate(w, b_fl, lin, ang, rot, trans);
                B_lin[bt] = f4(lin, 0.0f); B_ang[bt] = f4(ang, 0.0f); B_rot[bt] = f4(rot); B_trans[bt] = f4(trans, 0.0f);
            }
            __syncthreads();
            ISL_STAMP(5); // integrate
            if (live) isl_pose_stage(w, h, L, m, solved_dt + w.prm.dt_sub);
            else if (role_ang && (b_fl & RP_BF_GYRO)) B_axes[bt] = f4(qmul(q4(B_rot[bt]), b_pframe)); // (the next substep's increment; rot stands until the next integrate)
            else if (fused && !decided && t == THREADS - 1 && (sub == w.prm.num_substeps - 1 || (early_poll && sub == 0)) && isl == (int)blockIdx.x && s_early == 0) {
                // the verdict on this step, asked for EARLY by a lane that has nothing else to do: every workgroup arrived long ago (it does
                // so in substep 0), so the one L2 round trip of the commit poll hides behind the last relaxed sweep instead of following it
                // (early_poll: also in substep 0 — an abort standing by then is final whatever the count, and ends the step at the top of substep 1)
                const unsigned v = __hip_atomic_load((unsigned *)&w.flags[FL_ARRIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (((v >> ab_shift) & 0xffu) != 0) s_early = 2;                        // an abort of THIS step: final
                else if ((v & 0xffffu) >= (unsigned)arrive_target) s_early = 1;         // complete and clean: final
            }
            ISL_STAMP(6); // pose stage
            for (int it = 0; it < prm.num_internal_stabilization_iterations; ++it)
                for (int q = 0; q < nls; ++q) { if (myq == q) isl_solve(h, L, true, true); __syncthreads(); }
            ISL_STAMP(7); // relax sweep
        }
        if (go && has_restitution && any_bouncy)
            for (int q = 0; q < nls; ++q) { if (myq == q) isl_restitution(h, L); __syncthreads(); }
        // ---- write-back (S9, S10, advance_to_final_positions) ----
        if (!decided) { // fused: nothing leaves the workgroup before every workgroup validated its islands
#ifdef RP_ISL_PROFILE
            long long t_w0 = (long long)__builtin_readcyclecounter();
#endif
            if (t == 0 && s_early) s_go = s_early == 1; // (the early answer: see the last pose stage)
            else if (t == 0) {
                int spins = 0; unsigned v;
                while (((v = __hip_atomic_load((unsigned *)&w.flags[FL_ARRIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffffu) < (unsigned)arrive_target) {
                    if (ns > 1 && ((v >> ab_shift) & 0xffu) != 0) break; // (a launch of several steps: an abort of THIS step is final, its arrivals need not be waited for)
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1 << 22)) {
                        // ~1 s: a workgroup never became resident (the GPU is shared).  Turn the wait into an abort of the whole launch:
                        // the abort count is added only while the arrival count is still short (CAS), so a workgroup that later
                        // finds the count complete also finds the abort — either every workgroup writes back or none does.
                        unsigned seen = v;
                        while ((seen & 0xffffu) < (unsigned)arrive_target &&
                               !__hip_atomic_compare_exchange_strong((unsigned *)&w.flags[FL_ARRIVE], &seen, seen + ab_one, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { }
                        if ((seen & 0xffffu) < (unsigned)arrive_target) { v = seen + ab_one; __hip_atomic_store(&w.flags[FL_GRID_TIMEOUT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        v = seen; break; // the last arrival came in meanwhile: decide from the complete count
                    }
                }
                s_go = (v & 0xffffu) >= (unsigned)arrive_target && ((v >> ab_shift) & 0xffu) == 0;
            }
            __syncthreads();
            go = s_go != 0; decided = true;
#ifdef RP_ISL_PROFILE
            if (blockIdx.x == 0 && threadIdx.x == 0) w.dbg[11] += (long long)__builtin_readcyclecounter() - t_w0;
#endif
        }
        if (!go) break;
        if (live && !odd) isl_writeback(w, h, slot);
        if (role_lin) {
            body_writeback(w, b_gid, v3(B_lin[bt]), v3(B_ang[bt]), q4(B_rot[bt]), v3(B_trans[bt]));
            if (ns > 1 && w.b_quar[b_gid]) s_cross = 1; // a body went non-finite and was stopped: the host disables it before the next step (no further step here)
        }
        ISL_STAMP(8); // write-back
#ifdef RP_ISL_PROFILE
        if (blockIdx.x == 0 && threadIdx.x == 0) w.dbg[63] += 1;
#endif
    }
    if (!go) break;
    __syncthreads(); // this step's write-back (global memory, written by this workgroup) is visible to all of it before the next step reads it
    } // steps of this launch
    if (fused) { // the last workgroup to leave retires the step (or not, when aborted) and re-arms the counters
        __syncthreads();
        if (threadIdx.x == 0) {
            if (atomicAdd(&w.flags[FL_DEPART], 1) == (int)gridDim.x - 1) {
                // every workgroup has arrived by now (each one arrives before it can reach this point)
                const unsigned v = __hip_atomic_load((unsigned *)&w.flags[FL_ARRIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the steps every workgroup committed: all of them, or — the launch ended at its first aborted step j, for which every
                // workgroup that left had arrived — the full rounds of arrivals in front of that step
                w.flags[FL_STEP] += (v >> 16) == 0 ? ns : (int)(((v & 0xffffu) - 1u) / gridDim.x);
                if ((v >> 16) != 0) w.flags[FL_FAST_ABORT] = 1; // sticky: later fast launches exit until a full step ran
                __hip_atomic_store(&w.flags[FL_ARRIVE], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&w.flags[FL_DEPART], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ void __launch_bounds__(ISL_THREADS) k_island_solve(DevWorld w, int has_restitution, int fast, int retire, int fused) { island_solve_body<false>(w, has_restitution, fast, retire, fused); }
// ... and as a launch of `nsteps` fused steps (see island_solve_body)
__global__ void __launch_bounds__(ISL_THREADS) k_island_solve_steps(DevWorld w, int has_restitution, int nsteps) { island_solve_body<false>(w, has_restitution, 1, 1, 1, nsteps); }
// the same kernel with the WIDE validators (rp_island_stages.h): worlds with compound bodies or sleeping enabled
__global__ void __launch_bounds__(ISL_THREADS) k_island_solve_wide(DevWorld w, int has_restitution, int fast, int retire, int fused) { island_solve_body<true>(w, has_restitution, fast, retire, fused); }
// FrictionModel::Coulomb in the same kernel (round 6): the lane pair holds ContactWithCoulombFriction (rp_coulomb_pair.h) in registers
// — 16 warm-start rows per lane instead of 11 (82 KB of W), one island per CU like the twist form, the same fused / several-steps protocol
__global__ void __launch_bounds__(ISL_THREADS) k_island_solve_coul(DevWorld w, int has_restitution, int fast, int retire, int fused) { island_solve_body<false, true>(w, has_restitution, fast, retire, fused); }
__global__ void __launch_bounds__(ISL_THREADS) k_island_solve_coul_steps(DevWorld w, int has_restitution, int nsteps) { island_solve_body<false, true>(w, has_restitution, 1, 1, 1, nsteps); }
__global__ void __launch_bounds__(ISL_THREADS) k_island_solve_coul_wide(DevWorld w, int has_restitution, int fast, int retire, int fused) { island_solve_body<true, true>(w, has_restitution, fast, retire, fused); }
// ---- the generic island kernel ----------------------------------------------------------------------------------------------------
// One workgroup = one island, like k_island_solve, but the constraint is the HBM-resident one of the global path (rp_constraint.h /
// rp_coulomb.h through an accessor): thread m owns manifold m for the whole step, so its constraint planes are private to the
// thread (no fence, L2-resident: 87 planes x 16 B under FrictionModel::Coulomb do not fit registers or LDS), and only the solver
// bodies — the one thing manifolds share — live in LDS, ordered by the workgroup barrier between colour stages.  Serves the worlds
// whose constraint model k_island_solve does not hold in registers (FrictionModel::Coulomb): their islands used to be poisoned
// onto the global path, where every colour stage costs a cross-CU hand-off (~8 us) instead of a barrier.
// Same stage order as global_single_block (rp_global.h), so the result is bit-identical to the global path and the oracle.
#define ISL_GEN_THREADS 256 // k_island_generic: one thread per manifold (<= RP_ISL_NC_MAX = 160), FOUR wavefronts = one per SIMD, so a lane may hold up to
                            // 512 registers — room for the rows a stage reads (Acc::PRELOAD below)
template <bool PRE>
struct IslGenAccT {
    // PRELOAD: a stage fetches every row it reads in ONE round trip before its first store instead of one round trip per contact point
    // and part (rolled loops: 8 dependent trips of ~0.7 us per stage = what a stage cost).  At 512 threads per workgroup the preloaded
    // rows lived in scratch (256 registers per lane: 0.57 -> 1.67 ms per step, round 2); at 256 threads they fit the register file.
    static constexpr bool PRELOAD = PRE;
    const DevWorld &w; int pos; const IslLds &L;
    // The thread owns ONE manifold from generate to write-back: what every stage asks for FIRST — the two solver bodies, the point
    // count, the direction / inverse-mass / tangent header planes — stays in registers once generate has stored it (round 5: those
    // seven L2 round trips sat at the head of every stage's dependent chain: C3 under FrictionModel::Coulomb 581 -> 495 us per step).
    // The planes still go to memory (write-back reads them through the same accessor; nothing else reads an island's rows).
    // (Also measured in round 5: the 56 per-point planes of a sweep in LDS instead of L2 — 147 KB of dynamic LDS — 516 us: a stage is
    // bound by the ~2,500 dependent instructions of its four normal + four tangent solves on one wavefront per SIMD, not by its loads.)
    mutable int m_id1, m_id2, m_n, m_cid; mutable float4 m_h0, m_h1, m_h2, m_h6;
    RP_DEV IslGenAccT(const DevWorld &w_, int pos_, const IslLds &L_) : w(w_), pos(pos_), L(L_), m_id1(-1), m_id2(-1), m_n(0), m_cid(0) {
        m_h0 = make_float4(0, 0, 0, 0); m_h1 = m_h0; m_h2 = m_h0; m_h6 = m_h0;
    }
    RP_DEV float4 ld(int plane) const {
        if (plane <= CP_H6) { if (plane == CP_H0) return m_h0; if (plane == CP_H1) return m_h1; if (plane == CP_H2) return m_h2; if (plane == CP_H6) return m_h6; }
        return w.C[(size_t)plane * w.cons_cap + pos];
    }
    RP_DEV void st(int plane, float4 v) const {
        w.C[(size_t)plane * w.cons_cap + pos] = v;
        if (plane <= CP_H6) { if (plane == CP_H0) m_h0 = v; else if (plane == CP_H1) m_h1 = v; else if (plane == CP_H2) m_h2 = v; else if (plane == CP_H6) m_h6 = v; }
    }
    RP_DEV int id1() const { return m_id1; }
    RP_DEV int id2() const { return m_id2; }
    RP_DEV int n() const { return m_n; }
    RP_DEV int cids() const { return m_cid; }
    RP_DEV void set_meta(int a, int b, int cnt, int cid) const { w.k_b1[pos] = a; w.k_b2[pos] = b; w.k_n[pos] = cnt; w.k_cid[pos] = cid; m_id1 = a; m_id2 = b; m_n = cnt; m_cid = cid; }
    RP_DEV Vel vel(int id) const { return isl_vel(L, id); }
    RP_DEV void set_vel(int id, const Vel &v) const { isl_set_vel(L, id, v); }
    RP_DEV Xf xf(int id) const { return isl_xf(L, id); }
};
typedef IslGenAccT<true> IslGenAcc;
template <bool COUL>
__global__ void __launch_bounds__(ISL_GEN_THREADS) k_island_generic(DevWorld w, int has_restitution, int fast, int retire) {
    const bool aborted = fast && w.flags[FL_FAST_ABORT];
    if (retire && blockIdx.x == 0) { // SINGLE mode: workgroup 0 retires the step and publishes the scalars (as k_island_solve does)
        if (threadIdx.x == 0) { w.flags[FL_SEQ] += 1; if (!aborted) w.flags[FL_STEP] += 1; }
        __threadfence(); __syncthreads();
        publish_flags(w);
    }
    if (aborted) return;
    __shared__ float4 B_lin[RP_ISL_NB_MAX], B_ang[RP_ISL_NB_MAX], B_rot[RP_ISL_NB_MAX], B_trans[RP_ISL_NB_MAX];
    __shared__ int S_a[RP_ISL_NC_MAX], S_b[RP_ISL_NC_MAX], S_c[RP_ISL_NC_MAX], S_d[RP_ISL_NC_MAX];
    __shared__ int any_bouncy;
    const int t = threadIdx.x;
    const int n_islands = w.flags[FL_N_ISLANDS];
    const int nst_global = w.flags[FL_N_STAGES];
    const rp_integration_params &prm = w.prm.p;
    const bool fib = prm.friction_in_bias_pass || prm.num_internal_stabilization_iterations == 0;
    IslLds L;
    L.lin = B_lin; L.ang = B_ang; L.rot = B_rot; L.trans = B_trans; L.E = nullptr; L.F = nullptr; L.B0 = nullptr; L.B1 = nullptr;
    for (int isl = blockIdx.x; isl < n_islands; isl += gridDim.x) {
        const int nb = w.isl_nb[isl], nc = w.isl_nc[isl];
        const int bb = w.isl_body_begin[isl], cb = w.isl_cons_begin[isl];
        __syncthreads(); // previous island of this workgroup fully written back
        if (!w.isl_sorted[isl]) island_sort(w, isl, nc, cb, nst_global, S_a, S_b, S_c, S_d);
        int b_gid = -1, b_fl = 0;
        V3 b_incl = v3(0, 0, 0), b_inca = b_incl, b_invpi = b_incl; Q4 b_pframe = q4(0, 0, 0, 1);
        if (t < nb) { // S0: thread t owns solver body t
            int g = w.isl_bodies[bb + t];
            V3 lin, ang, trans; Q4 rot;
            body_begin(w, g, lin, ang, rot, trans, b_incl, b_inca);
            b_gid = g; b_fl = w.b_flags[g];
            B_lin[t] = f4(lin, 0.0f); B_ang[t] = f4(ang, 0.0f); B_rot[t] = f4(rot); B_trans[t] = f4(trans, 0.0f);
            b_invpi = v3(w.b_invpi[g]); b_pframe = q4(w.b_pframe[g]);
        }
        if (t == 0) any_bouncy = 0;
        const int nls = w.isl_nstages[isl];
        const bool live = t < nc;
        int slot = -1, myq = -1;
        // the island's manifolds take the constraint rows from the top of the planes down; the global path's own rows grow from 0
        // (together they are at most the live pairs, and cons_cap = pool_cap)
        const IslGenAcc A(w, w.cons_cap - 1 - (cb + (live ? t : 0)), L);
        if (live) { slot = w.isl_cons[cb + t]; myq = w.isl_cstage[cb + t]; }
        __syncthreads();
        if (live) { // S1
            const int g1 = w.isl_cg1[cb + t], g2 = w.isl_cg2[cb + t], l1 = w.isl_cl1[cb + t], l2 = w.isl_cl2[cb + t];
            const bool bouncy = COUL ? coul_generate(w, A, slot, g1, g2, l1, l2) : cons_generate(w, A, slot, g1, g2, l1, l2);
            if (bouncy) any_bouncy = 1;
        }
        __syncthreads();
        // (the barrier orders the LDS velocities only: a thread's constraint rows are its own, so its global stores may stay in flight
        // instead of being drained as __syncthreads() would — measured worth little, 575 -> 571 us: a stage is bound by the dependent
        // arithmetic of its four normal + four tangent solves, not by the stores)
#define ISLGEN_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define ISLGEN_SWEEP(MODE, SDT) for (int q = 0; q < nls; ++q) { if (myq == q) cons_apply_model<COUL>(w, A, MODE, fib, SDT); ISLGEN_BARRIER(); }
        for (int sub = 0; sub < w.prm.num_substeps; ++sub) {
            const float solved_dt = (float)sub * w.prm.dt_sub;
            if (t < nb) { // S2
                V3 lin = v3(B_lin[t]), ang = v3(B_ang[t]);
                body_increment(w, b_fl, lin, ang, q4(B_rot[t]), b_incl, b_inca, b_invpi, b_pframe);
                B_lin[t] = f4(lin, 0.0f); B_ang[t] = f4(ang, 0.0f);
            }
            __syncthreads();
            ISLGEN_SWEEP(MODE_WARMSTART, solved_dt)
            for (int it = 0; it < prm.num_internal_pgs_iterations; ++it) ISLGEN_SWEEP(MODE_BIAS, solved_dt)
            if (t < nb) { // S6
                V3 lin = v3(B_lin[t]), ang = v3(B_ang[t]), trans = v3(B_trans[t]); Q4 rot = q4(B_rot[t]);
                body_integrate(w, b_fl, lin, ang, rot, trans);
                B_lin[t] = f4(lin, 0.0f); B_ang[t] = f4(ang, 0.0f); B_rot[t] = f4(rot); B_trans[t] = f4(trans, 0.0f);
            }
            __syncthreads();
            for (int it = 0; it < prm.num_internal_stabilization_iterations; ++it) ISLGEN_SWEEP(MODE_RELAX, solved_dt + w.prm.dt_sub)
        }
        if (has_restitution && any_bouncy) ISLGEN_SWEEP(MODE_RESTITUTION, 0.0f)
#undef ISLGEN_SWEEP
#undef ISLGEN_BARRIER
        if (live) { if (COUL) coul_writeback(w, A, slot); else cons_writeback(w, A, slot); }
        if (t < nb) body_writeback(w, b_gid, v3(B_lin[t]), v3(B_ang[t]), q4(B_rot[t]), v3(B_trans[t]));
    }
}

void rp_launch_islands_build(const DevWorld &w, hipStream_t st) {
    // every workgroup must be resident (grid barriers): at most DevWorld::gbar_blocks workgroups of 1024 threads (rp_gridbar.h)
    int n = w.n_bodies > w.pool_cap ? w.n_bodies : w.pool_cap;
    int blocks = (n + 255) / 256; if (blocks > w.gbar_blocks) blocks = w.gbar_blocks; if (blocks < 1) blocks = 1; // ~4 wavefronts of items per workgroup
    hipLaunchKernelGGL(k_layout_rebuild, dim3(blocks), dim3(1024), 0, st, w);
}
// Most workgroups a fused fast step may launch: its arrival barrier needs every workgroup resident at once, so the cap comes from
// the device (CU count x the occupancy of k_island_solve) with 1/16 of the CUs left free for whatever else the GPU is running
// (256 CUs, one 512-thread workgroup with ~90 KB of LDS per CU -> 240).  A launch that still meets a non-resident workgroup
// aborts the step after ~1 s (FL_GRID_TIMEOUT) and the world falls back to the two-kernel fast graph.
int rp_fused_grid(int device) {
    static int cached[64] = {0};
    if (device >= 0 && device < 64 && cached[device]) return cached[device];
    hipDeviceProp_t prop;
    int per_cu = 0, cus = 0;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_island_solve, ISL_THREADS, 0) != hipSuccess) per_cu = 0;
    { int pw = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pw, k_island_solve_wide, ISL_THREADS, 0) != hipSuccess) pw = 0; if (pw < per_cu) per_cu = pw; }
    { int pc = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, k_island_solve_coul_wide, ISL_THREADS, 0) != hipSuccess) pc = 0; if (pc < per_cu) per_cu = pc; } // (111 KB of LDS: one per CU, like the twist forms)
    if (per_cu < 1 || cus < 1) return 0;
    if (per_cu > 2) per_cu = 2; // more co-resident islands per CU than this only slow each other down
    int g = cus * per_cu - (cus + 15) / 16;
    if (g < 1) g = 1;
    if (device >= 0 && device < 64) cached[device] = g;
    return g;
}
void rp_launch_island_solve_dense(const DevWorld &w, hipStream_t st, int grid, int has_restitution, int fast, int retire, int fused, int wide);
void rp_launch_island_solve_dense_steps(const DevWorld &w, hipStream_t st, int grid, int has_restitution, int nsteps); // rp_islands_lean.hip
void rp_launch_island_solve_steps(const DevWorld &w, hipStream_t st, int grid, int has_restitution, int nsteps, int dense) {
    if (w.prm.p.friction_model == RP_FRICTION_COULOMB) { hipLaunchKernelGGL(k_island_solve_coul_steps, dim3(grid < 1 ? 1 : grid), dim3(ISL_THREADS), 0, st, w, has_restitution, nsteps); return; }
    if (dense) { rp_launch_island_solve_dense_steps(w, st, grid, has_restitution, nsteps); return; }
    hipLaunchKernelGGL(k_island_solve_steps, dim3(grid < 1 ? 1 : grid), dim3(ISL_THREADS), 0, st, w, has_restitution, nsteps);
}
void rp_launch_island_solve(const DevWorld &w, hipStream_t st, int grid, int has_restitution, int fast, int retire, int fused, int dense, int wide) {
    if (grid < 1) grid = 1;
    if (w.prm.p.friction_model == RP_FRICTION_COULOMB) { // (no register-lean form of the Coulomb pair: `dense` is never planned for this model, rp_api_step.inc)
        if (w.isl_generic) hipLaunchKernelGGL(k_island_generic<true>, dim3(grid), dim3(ISL_GEN_THREADS), 0, st, w, has_restitution, fast, retire); // RP_ISL_GENERIC=1: rows in HBM, one lane per manifold (A/B, tests)
        else if (wide) hipLaunchKernelGGL(k_island_solve_coul_wide, dim3(grid), dim3(ISL_THREADS), 0, st, w, has_restitution, fast, retire, fused);
        else hipLaunchKernelGGL(k_island_solve_coul, dim3(grid), dim3(ISL_THREADS), 0, st, w, has_restitution, fast, retire, fused);
        return;
    }
    if (w.isl_generic) { hipLaunchKernelGGL(k_island_generic<false>, dim3(grid), dim3(ISL_GEN_THREADS), 0, st, w, has_restitution, fast, retire); return; } // RP_ISL_GENERIC=1: the twist model through the generic kernel (tests)
    if (dense) rp_launch_island_solve_dense(w, st, grid, has_restitution, fast, retire, fused, wide); // rp_islands_lean.hip
    else if (wide) hipLaunchKernelGGL(k_island_solve_wide, dim3(grid), dim3(ISL_THREADS), 0, st, w, has_restitution, fast, retire, fused);
    else hipLaunchKernelGGL(k_island_solve, dim3(grid), dim3(ISL_THREADS), 0, st, w, has_restitution, fast, retire, fused);
}

// workgroups of k_layout_rebuild (1024 threads) one CU holds at once (0 = the query failed): input of DevWorld::gbar_blocks (rp_api.hip)
int rp_occ_layout_rebuild(void) { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_layout_rebuild, 1024, 0) != hipSuccess) n = 0; return n; }
