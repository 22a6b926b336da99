// rp_tiles.hip — a whole colour sweep of a giant island inside one CU per tile.
//
// What it replaces: on the global path (rp_solver.hip) every colour stage of every sweep is a kernel launch — StagedIslandSolver's
// stage machine (/root/reference/src/dynamics/solver/staged_island_solver/{worker.rs:438-734, solve.rs:12-92, sync.rs:39-186}) pays a
// worker barrier at the same places.  On MI355X any dependent cross-CU exchange (launch boundary, grid barrier, per-body hand-off:
// all three measured) costs 5-9 us, so a step of b3d_large_pyramid (one island of 20,100 bodies, 4 substeps x 2
// sweeps x 7 colours) spent ~0.5 ms on 64 stage launches.  A workgroup barrier costs ~0.2 us.  So the island is cut into TILES:
//
//   * the global-path bodies are ordered along a Morton curve of their centres of mass (counting sort by cell) and cut into runs of T
//     bodies: tile = the bodies one workgroup OWNS;
//   * a tile's CONE is everything the state of its owned bodies after a whole sweep depends on: walking the sweep's stages backwards,
//     every constraint of stage s that touches a body already needed is taken in and its bodies become needed for the stages before s.
//     Dependencies only run along paths of strictly decreasing stage, so the cone is a thin halo, not the H-ring neighbourhood;
//   * one workgroup loads the velocities of its cone bodies into LDS, runs ALL stages of the sweep over its cone constraints with a
//     workgroup barrier between stages, and writes back the bodies it owns.  Halo constraints are solved redundantly by every tile
//     whose cone holds them — on identical inputs, in the reference's order per body, hence with identical bits (-ffp-contract=off);
//     only the tile that owns a constraint's first body stores its rows;
//   * the velocities are double-buffered (s_lin / s_ang <-> t_lin / t_ang): a tile reads its halo from the buffer the previous kernel
//     left and writes its owned bodies to the other one, so no tile ever sees another tile's result of the same sweep.
//
// A sweep is then ONE launch (8 per step instead of 64) whose critical path is stages x (row fetch + ~100 dependent flops + barrier).
// The arithmetic is cons_solve / cons_restitution of rp_constraint.h through an accessor: same operands, same order per accumulator as
// the per-stage launches, the dataflow launch and the oracle.  Tilings are rebuilt on the device when the constraint layout changes
// (FL_FLOW_DIRTY) and verified there: a world the tiling cannot hold (overflow colour in use, cone larger than the LDS budget, fewer
// than RP_TILE_MIN_BODIES bodies) publishes FL_N_TILES = 0 and keeps the per-stage launches; a sweep kernel that finds no tiling runs
// the sweep in one workgroup (correct, slow) until the host has read the hint.
#include "rp_global.h"
#include "rp_lanepair.h"
#include "rp_gridbar.h"
#include <cstdlib>

#define RP_TILE_THREADS 256
#define RP_TILE_HASH 2048 // slots of the body -> local id table of one tile (2 x RP_TILE_BCAP)

// floats as order-preserving unsigned keys (atomicMin / atomicMax over the centres of mass)
RP_DEV unsigned tile_ord(float f) { unsigned u = (unsigned)__float_as_int(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
RP_DEV float tile_unord(unsigned u) { u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; return __int_as_float((int)u); }
// joint colour stages of a sweep as the tiles see them: the parallel colours, then the small ones (k_joint_layout)
RP_DEV int tile_joint_stages(const DevWorld &w) { return w.n_joints > 0 ? w.j_stage_count[RP_NUM_COLORS] : 0; }
RP_DEV unsigned tile_hash(int g) { return ((unsigned)g * 2654435761u) >> 21; } // 11 bits

// ---- tiling (one launch behind grid barriers, only when the layout changed) --------------------------------------------------------
// k_tiles_sort orders the bodies along a Morton curve and produces two things:
//   * tl_owned / tl_body_tile: the global-path bodies in curve order, cut into tiles of T;
//   * b_order: the curve rank of EVERY body (a permutation of the body indices).  The NEXT layout rebuild ranks the manifolds of a colour
//     stage by the order value of their owner body (k_layout_rebuild, rp_islands.hip) instead of its arena index, so the rows a tile
//     fetches in one stage are runs of consecutive positions — with index order a tile's 16-byte rows each sat in a cache line of their
//     own and the sweeps were bound by the CU's L2 bandwidth (128 B fetched per 16 B used; measured: 4 us per relaxed stage).
//     Positions inside a stage never matter for the result (a colour is body-disjoint), and any injective order is valid: bodies
//     inserted after a sort keep b_order[i] = i, which lies above every rank handed out before.
__global__ void __launch_bounds__(1024) k_tiles_sort(DevWorld w) {
    if (!w.flags[FL_FLOW_DIRTY]) return; // (cleared by k_begin_generate, which runs after this kernel)
    const int gid = gbar_item(), gstride = gridDim.x * blockDim.x, t = threadIdx.x;
    if (gid == 0) w.flags[FL_TILE_JMAX] = 0; // (k_tiles_cones, the next launch, takes the maximum over the cones it builds)
    for (int idx = gid; idx < w.tile_cap * RP_TILE_NBR_WORDS(w.tile_cap); idx += gstride) w.tl_nbr[idx] = 0u; // (... and notes which tiles exchange bodies)
    int M = w.flags[FL_N_CONS]; if (M > w.cons_cap) M = w.cons_cap;
    const int nst = w.flags[FL_N_STAGES], nb = w.n_bodies;
    const int njl = w.n_joints > 0 ? w.flags[FL_NJ_OVF_BEGIN] + w.flags[FL_NJ_OVF_COUNT] : 0, njs = tile_joint_stages(w); // live joints, their colour stages (small colours included)
    const bool joint_overflow = w.n_joints > 0 && w.j_stage_begin[RP_NUM_COLORS] > 0; // joints in the one colour that is not body-disjoint: not tiled
    if (w.flags[FL_N_GLOB_BODIES] < w.tile_min || M + njl < w.tile_min || w.flags[FL_HAS_OVERFLOW_COLOR] || joint_overflow || nst + njs > RP_TILE_STAGES || nst + njs < 1) {
        if (gid == 0) { w.flags[FL_N_TILES] = 0; w.tl_bbox[8] = 0u; w.dbg[900] += 1; w.dbg[901] = 1; } // nothing worth tiling / a serial overflow colour: colour stages stay launches
        return;
    }
    GridBar bar = gbar_begin(w, 4);
    // pass R: can the last partition stay?  A tiling only schedules the work — any partition of the global-path bodies gives the same
    // bits — so the five passes of the sort are only needed when the SET of global-path bodies changed (or bodies came / went); a
    // layout change of a settling pile (pairs begin / end to touch) leaves it alone.  One pass and one barrier: every body compares what
    // it is now with what the last sort made of it.  Every 32nd time the sort runs anyway (bodies drift along the curve).
    // tl_bbox[11]: scratch of this pass, [12]: tiles of the last sort that fitted, [13]: reuses in a row
    const bool reusable = w.tl_bbox[8] != 0u && (int)w.tl_bbox[9] == nb && w.tl_bbox[12] != 0u && w.tl_bbox[13] < 32u; // (written behind the last barrier of a launch: uniform)
    if (reusable) {
        bool differs = false;
        for (int i = gid; i < nb; i += gstride) differs |= (w.tl_body_tile[i] >= 0) != global_body(w, i);
        if (__any(differs) && (t & 63) == 0) atomicOr(&w.tl_bbox[11], 1u);
        for (int pos = gid; pos < M; pos += gstride) { int a, b; flow_ids(w, pos, a, b); w.fk_ids[pos] = make_int2(a, b); } // (what pass 0 of the sort would have done for the cones)
        GBAR_SYNC(bar);
        if (__hip_atomic_load(&w.tl_bbox[11], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            if (gid == 0) { w.flags[FL_N_TILES] = (int)w.tl_bbox[12]; w.tl_bbox[13] += 1u; w.dbg[909] += 1; w.dbg[905] = 0; w.dbg[906] = 0; w.dbg[907] = 0; w.dbg[908] = 0; }
            gbar_end(bar);
            return;
        }
        GBAR_SYNC(bar); // (every workgroup has read the verdict before it is cleared)
        if (gid == 0) w.tl_bbox[11] = 0u;
    }
    // pass 0: the solver bodies of every position; bounding box of the centres of mass (tl_bbox rests at min = ~0, max = 0)
    for (int pos = gid; pos < M; pos += gstride) { int a, b; flow_ids(w, pos, a, b); w.fk_ids[pos] = make_int2(a, b); }
    for (int i0 = gid - (t & 63); i0 < nb; i0 += gstride) { // (wave-uniform trip count: one atomic per wavefront and bound, not per body)
        const int i = i0 + (t & 63);
        unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
        if (i < nb) {
            const float4 c = w.b_wcom[i];
            if (isfinite(c.x) && isfinite(c.y) && isfinite(c.z)) { lo[0] = hi[0] = tile_ord(c.x); lo[1] = hi[1] = tile_ord(c.y); lo[2] = hi[2] = tile_ord(c.z); }
        }
        for (int off = 32; off > 0; off >>= 1)
            for (int a = 0; a < 3; ++a) {
                const unsigned l2 = (unsigned)__shfl_xor((int)lo[a], off, 64), h2 = (unsigned)__shfl_xor((int)hi[a], off, 64);
                lo[a] = l2 < lo[a] ? l2 : lo[a]; hi[a] = h2 > hi[a] ? h2 : hi[a];
            }
        if ((t & 63) == 0 && hi[0] != 0u)
            for (int a = 0; a < 3; ++a) { atomicMin(&w.tl_bbox[a], lo[a]); atomicMax(&w.tl_bbox[3 + a], hi[a]); }
    }
    GBAR_SYNC(bar);
    // pass 1: Morton cell of every body.  The 12 cell bits go to the axes greedily (always to the axis whose cells are the longest), so
    // a flat scene spends none on its thin axis; the interleaving follows the same order, most significant bit first.
    {
        float mn[3], ex[3], cs[3];
        for (int a = 0; a < 3; ++a) { mn[a] = tile_unord(w.tl_bbox[a]); ex[a] = tile_unord(w.tl_bbox[3 + a]) - mn[a]; if (!(ex[a] > 0.0f)) ex[a] = 0.0f; cs[a] = ex[a]; }
        int bits[3] = {0, 0, 0};
        unsigned ordw = 0;
        for (int k = 0; k < 12; ++k) {
            int a = (cs[1] > cs[0]) ? 1 : 0; if (cs[2] > cs[a]) a = 2;
            ordw |= (unsigned)a << (2 * k); bits[a]++; cs[a] *= 0.5f;
        }
        for (int i = gid; i < nb; i += gstride) {
            const float4 c = w.b_wcom[i];
            const float p[3] = {c.x, c.y, c.z};
            int q[3], rem[3];
            for (int a = 0; a < 3; ++a) {
                const int n = 1 << bits[a];
                int v = (ex[a] > 0.0f && isfinite(p[a])) ? (int)((p[a] - mn[a]) / ex[a] * (float)n) : 0;
                q[a] = v < 0 ? 0 : (v > n - 1 ? n - 1 : v); rem[a] = bits[a];
            }
            int cell = 0;
            for (int k = 0; k < 12; ++k) { const int a = (ordw >> (2 * k)) & 3; rem[a]--; cell = (cell << 1) | ((q[a] >> rem[a]) & 1); }
            const bool glob = global_body(w, i);
            atomicAdd(&w.tl_hist[cell], 1);
            if (glob) atomicAdd(&w.tl_hist[RP_TILE_CELLS + cell], 1);
            w.tl_cell[i] = glob ? cell : -1 - cell; // (the sign carries "on the global path")
        }
    }
    GBAR_SYNC(bar);
    // pass 2: exclusive scans of the two cell histograms (workgroup 0; 4 cells per thread): every body / global-path bodies; the
    // histograms go back to zero and serve as fill cursors in pass 3
    if (blockIdx.x == 0) {
        __shared__ int sc[1024];
        for (int which = 0; which < 2; ++which) {
            int *hist = w.tl_hist + which * RP_TILE_CELLS, *ofs = w.tl_cellofs + which * RP_TILE_CELLS;
            const int base = t * (RP_TILE_CELLS / 1024);
            int sum = 0;
            for (int k = 0; k < RP_TILE_CELLS / 1024; ++k) sum += hist[base + k];
            __syncthreads();
            sc[t] = sum;
            __syncthreads();
            for (int off = 1; off < 1024; off <<= 1) { int v = t >= off ? sc[t - off] : 0; __syncthreads(); sc[t] += v; __syncthreads(); }
            int run = sc[t] - sum;
            for (int k = 0; k < RP_TILE_CELLS / 1024; ++k) { int h = hist[base + k]; ofs[base + k] = run; run += h; hist[base + k] = 0; }
            if (which == 1 && t == 1023) w.tl_bbox[6] = (unsigned)sc[1023]; // global-path bodies
        }
    }
    GBAR_SYNC(bar);
    // pass 3: every body takes a slot of its cell's segment (tl_sorted: the members of every cell, in whatever order the atomics make it)
    const int NG = (int)w.tl_bbox[6];
    int T = (NG + w.tile_target - 1) / w.tile_target; T = T < 64 ? 64 : (T > 256 ? 256 : T);
    const int NT = (NG + T - 1) / T;
    const bool fits = NT >= 1 && NT <= w.tile_cap;
    for (int i = gid; i < nb; i += gstride) {
        const int cc = w.tl_cell[i], cell = cc >= 0 ? cc : -1 - cc;
        w.tl_sorted[w.tl_cellofs[cell] + atomicAdd(&w.tl_hist[cell], 1)] = i;
    }
    GBAR_SYNC(bar);
    // pass 4: a global-path body's rank among the global-path bodies -> owner tile.  DETERMINISTIC: begin of its cell's segment + the
    // global-path bodies of that cell with a smaller index (a cell holds a handful of bodies), so the same world always gets the same
    // tiles and cones (round 3 ranked inside a cell by atomics: any partition gives the same bits, but a failure could not be replayed)
    for (int i = gid; i < nb; i += gstride) {
        const int cc = w.tl_cell[i];
        int tile = -1;
        if (cc >= 0 && fits) {
            const int beg = w.tl_cellofs[cc], end = cc + 1 < RP_TILE_CELLS ? w.tl_cellofs[cc + 1] : nb;
            int r = w.tl_cellofs[RP_TILE_CELLS + cc];
            for (int k = beg; k < end; ++k) { const int m = w.tl_sorted[k]; r += (m < i && w.tl_cell[m] >= 0) ? 1 : 0; }
            w.tl_owned[r] = i; tile = r / T;
        }
        w.tl_body_tile[i] = tile;
    }
    if (gid == 0) { // (dbg[900..]: statistics of the last tiling, tools/tile_diag.py)
        w.flags[FL_N_TILES] = fits ? NT : 0;
        w.tl_bbox[7] = (unsigned)T; w.tl_bbox[8] = 1u; w.tl_bbox[9] = (unsigned)nb; // [8]: the sort ran — k_tiles_cones derives b_order from it
        w.tl_bbox[12] = fits ? (unsigned)NT : 0u; w.tl_bbox[13] = 0u;
        w.dbg[900] += 1; w.dbg[901] = fits ? 0 : 2; w.dbg[902] = NG; w.dbg[903] = NT; w.dbg[904] = T; w.dbg[905] = 0; w.dbg[906] = 0; w.dbg[907] = 0; w.dbg[908] = 0;
    }
    // the scratch of the sort goes back to its rest state once the ranks are out: by the cone kernel (next launch: a kernel boundary)
    gbar_end(bar);
}

// The cone of every tile (see the file header): one workgroup per tile, launched behind k_tiles_sort.
// The needed-bodies table of a tile lives in LDS: arena index -> begin of the body's sweep-ordered toucher list, and one packed word
// (tile-local id | list entries still ahead | the stage at which the body joined the cone: it takes part in the stages BEFORE that one,
// owned bodies in all of them).
#define TW_LID(wd) ((wd) & 0x7ff)
#define TW_CUR(wd) (((wd) >> 11) & 0x1fff)
#define TW_SINCE(wd) ((wd) >> 24)
#define TW_PACK(since, cur, lid) (((since) << 24) | ((cur) << 11) | (lid))
// hk: arena index; hw: TW_PACK(stage at which the body joined, contact-list entries still ahead, tile-local id); hbeg / hjbeg: begin of
// the body's contact / joint toucher list; hj: joint-list entries still ahead; live[tile-local id] = slot (the walkers go over the
// bodies, not over the sparse table)
struct TileTab { int *hk, *hw, *hbeg, *hj, *hjbeg, *live, *nloc, *bad; };
RP_DEV void tile_insert(const TileTab &T, int g, int since, int cursor, int begin, int jcursor, int jbegin) {
    unsigned h = tile_hash(g) & (RP_TILE_HASH - 1);
    for (int probes = 0; probes < RP_TILE_HASH; ++probes) {
        int old = atomicCAS(&T.hk[h], -1, g);
        if (old == -1) {
            const int lid = atomicAdd(T.nloc, 1);
            if (lid >= RP_TILE_BCAP || cursor > 0x1fff) { *T.bad = 1; return; } // (the word keeps its rest value: since = -1, never walked)
            T.hbeg[h] = begin; T.hj[h] = jcursor; T.hjbeg[h] = jbegin; T.hw[h] = TW_PACK(since, cursor, lid); T.live[lid] = (int)h;
            return;
        }
        if (old == g) return;
        if (*T.bad) return; // over budget: the tiling is being abandoned, do not fill the table
        h = (h + 1) & (RP_TILE_HASH - 1);
    }
    *T.bad = 1;
}
RP_DEV int tile_find(const TileTab &T, int g) { // slot of body g or -1
    unsigned h = tile_hash(g) & (RP_TILE_HASH - 1);
    for (int probes = 0; probes < RP_TILE_HASH; ++probes) {
        int k = T.hk[h];
        if (k == g) return (int)h;
        if (k == -1) return -1;
        h = (h + 1) & (RP_TILE_HASH - 1);
    }
    return -1;
}
// entries of `list` (ascending keys, `key(v)` = v >> shift) that lie below `limit`, of the first `n`: the cursor of a body that joins
// the cone for the stages before the one that begins at `limit`
RP_DEV int tile_cursor_below(const int *list, int begin, int n, int shift, int limit) {
    int c = n;
    int e8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e8[k] = k < c ? (list[begin + c - 1 - k] >> shift) : -1; // the last eight entries in one round trip
    int drop = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (e8[k] >= limit) ++drop; // (entries descend from the end: the ones >= limit are a prefix of e8)
    c -= drop;
    if (drop == 8) while (c > 0 && (list[begin + c - 1] >> shift) >= limit) --c;
    return c;
}
#define RP_CONE_THREADS 512
#ifdef RP_CONES_CHECK // diagnosis build: an index outside its array is recorded (dbg[890] = site, dbg[891] = index, dbg[892] = bound) and skipped
#define CCHK(site, idx, bound) (((long long)(idx) >= 0 && (long long)(idx) < (long long)(bound)) ? true : (atomicMax((unsigned long long *)&w.dbg[890], (unsigned long long)(site)), w.dbg[891] = (long long)(idx), w.dbg[892] = (long long)(bound), false))
#else
#define CCHK(site, idx, bound) true
#endif
__global__ void __launch_bounds__(RP_CONE_THREADS) k_tiles_cones(DevWorld w) {
    if (!w.flags[FL_FLOW_DIRTY]) return;
    const int t = threadIdx.x, nt = blockDim.x, gid = blockIdx.x * nt + t, gstride = gridDim.x * nt;
    for (int idx = gid; idx < 2 * RP_TILE_CELLS; idx += gstride) w.tl_hist[idx] = 0; // rest state of the sort scratch
    if (gid == 0) { w.tl_bbox[0] = w.tl_bbox[1] = w.tl_bbox[2] = 0xffffffffu; w.tl_bbox[3] = w.tl_bbox[4] = w.tl_bbox[5] = 0u; }
    if (w.tl_bbox[8]) {
        // the curve rank of every body, deterministic: begin of its cell's segment + the bodies of that cell with a smaller index (the
        // slots inside a segment were handed out by atomics; a cell holds a handful of bodies)
        const int nb = (int)w.tl_bbox[9]; // bodies the sort covered
        for (int i = gid; i < nb; i += gstride) {
            if (!CCHK(1, i, w.n_bodies)) continue;
            const int cc = w.tl_cell[i], cell = cc >= 0 ? cc : -1 - cc;
            if (!CCHK(2, cell, RP_TILE_CELLS)) continue;
            const int beg = w.tl_cellofs[cell], end = cell + 1 < RP_TILE_CELLS ? w.tl_cellofs[cell + 1] : nb;
            if (!CCHK(3, beg, w.n_bodies + 1) || !CCHK(4, end, w.n_bodies + 1)) continue;
            int r = 0;
            for (int k = beg; k < end; ++k) r += w.tl_sorted[k] < i;
            w.b_order[i] = beg + r;
        }
        // the first sort of a world: its layout was still built in arena-index order — have the next step rebuild it along the curve
        if (gid == 0) { if (w.tl_bbox[10] == 0u) w.flags[FL_LAYOUT_DIRTY] = 1; w.tl_bbox[10] += 1u; }
    }
    // ONE read of the tile count per workgroup: a workgroup whose cone does not fit zeroes FL_N_TILES while others are still starting, and
    // waves of one workgroup that read different counts would leave the loop (and its barriers) at different tiles — the waves that
    // stay then walk hash slots the others never initialised (found in round 6 on a collapsing pyramid of base 400: a memory fault)
    __shared__ int nt_sh;
    if (t == 0) nt_sh = __hip_atomic_load(&w.flags[FL_N_TILES], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int NT = nt_sh;
    if (NT <= 0) return;
    const int nst = w.flags[FL_N_STAGES], NG = (int)w.tl_bbox[6], T = (int)w.tl_bbox[7];
    const int njs = tile_joint_stages(w), nall = njs + nst; // a sweep = every joint stage, then every contact stage (solve.rs:89-92)
    __shared__ int hk[RP_TILE_HASH], hw[RP_TILE_HASH], hbeg[RP_TILE_HASH], hj[RP_TILE_HASH], hjbeg[RP_TILE_HASH];
    __shared__ int live[RP_TILE_BCAP], cpos[RP_TILE_CCAP / 2], ssoff[RP_TILE_STAGES + 2];
    __shared__ int nloc, ncons, bad, nsnap;
    const TileTab Tb = {hk, hw, hbeg, hj, hjbeg, live, &nloc, &bad};
    for (int tile = blockIdx.x; tile < NT; tile += gridDim.x) {
        __syncthreads();
        for (int h = t; h < RP_TILE_HASH; h += nt) { hk[h] = -1; hw[h] = (int)0xff000000; } // (a slot claimed during a stage shows since = -1 until its owner has filled it in: skipped)
        if (t == 0) { nloc = 0; ncons = 0; bad = 0; }
        __syncthreads();
        const int ob = tile * T, oc = (NG - ob) < T ? (NG - ob) : T;
        for (int k = t; k < oc; k += nt) { if (!CCHK(5, ob + k, w.n_bodies)) continue; const int g = w.tl_owned[ob + k]; if (!CCHK(6, g, w.n_bodies)) continue; const int2 d = w.fb_deg[g], bg = w.fb_begin[g]; tile_insert(Tb, g, nall, d.x, bg.x, njs ? d.y : 0, bg.y); }
        int4 *cons = w.tl_cons + (size_t)tile * RP_TILE_CCAP;
        int *soff = w.tl_soff + (size_t)tile * (RP_TILE_STAGES + 1);
        __syncthreads();
        if (t == 0) { soff[nall] = RP_TILE_CCAP; nsnap = nloc; }
        __syncthreads();
        // Backwards over the sweep.  Every needed body walks its own toucher lists (k_flow_ranks: the manifolds / joints that touch it in
        // sweep order, and the body on the other side of each) from the end: the items of a stage are one contiguous range of positions
        // (joint sweep indices) and a body has at most one item per stage (a colour is body-disjoint), so "my item of stage S" is the
        // list entry under the cursor or nothing.  The item joins the cone; its other body, if it was not needed yet, becomes needed for
        // the stages before S — nothing else of stage S can touch that body, so lookups and insertions of one stage may interleave
        // freely.  When both bodies were needed already, both find the item: the one with the smaller index reports it.
        for (int S = nall - 1; S >= 0; --S) {
            const bool jst = S < njs;
            const int beg = jst ? w.j_stage_begin[S] : w.stage_begin[S - njs];
            const int n0 = nsnap < RP_TILE_BCAP ? nsnap : RP_TILE_BCAP; // the bodies needed before this stage began
            for (int lid = t; lid < n0; lid += nt) {
                const int h = live[lid];
                const int a = hk[h], wd = hw[h];
                if (TW_SINCE(wd) <= S) continue;
                const int c = jst ? hj[h] : TW_CUR(wd);
                if (c <= 0) continue;
                const int at = (jst ? hjbeg[h] : hbeg[h]) + c - 1;
                if (!CCHK(7, at, 2 * (size_t)w.cons_cap)) {
#ifdef RP_CONES_CHECK
                    if (atomicCAS((unsigned long long *)&w.dbg[888], 0ull, 1ull) == 0ull) {
                    w.dbg[893] = hbeg[h]; w.dbg[894] = c; w.dbg[895] = h; w.dbg[896] = lid; w.dbg[897] = a; w.dbg[898] = wd; w.dbg[899] = (long long)S * 100000 + n0; w.dbg[889] = (long long)nloc * 100000 + tile; w.dbg[887] = (long long)nsnap * 100000 + oc; w.dbg[886] = (long long)bad * 100000 + ncons; }
#endif
                    continue;
                }
                const int item = jst ? w.f_jsorted[at] : (w.f_sorted[at] >> 1), b = jst ? w.f_jother[at] : w.f_other[at];
                if (b >= 0 && !CCHK(8, b, w.n_bodies)) continue;
                if (item < beg) continue; // this body has no item of stage S
                if (jst) hj[h] = c - 1; else hw[h] = wd - (1 << 11);
                if (b >= 0) {
                    const int hb = tile_find(Tb, b);
                    if (hb >= 0 && TW_SINCE(hw[hb]) > S) { if (b < a) continue; } // needed on both sides: b's walker reports it
                    else { // b joins the cone for the stages before S: its cursors skip the entries of stage S and later
                        const int2 db = w.fb_deg[b], bb = w.fb_begin[b];
                        // (joining at a joint stage: only joint stages lie before it; at a contact stage: every joint stage does)
                        const int cc = jst ? 0 : tile_cursor_below(w.f_sorted, bb.x, db.x, 1, beg);
                        const int cj = !njs ? 0 : (jst ? tile_cursor_below(w.f_jsorted, bb.y, db.y, 0, beg) : db.y);
                        tile_insert(Tb, b, S, cc, bb.x, cj, bb.y);
                    }
                }
                const int k = atomicAdd(&ncons, 1);
                if (k < RP_TILE_CCAP) cons[RP_TILE_CCAP - 1 - k] = make_int4(jst ? -1 - item : item, 0, 0, 0); else bad = 1;
            }
            __syncthreads();
            if (t == 0) { soff[S] = RP_TILE_CCAP - (ncons < RP_TILE_CCAP ? ncons : RP_TILE_CCAP); nsnap = nloc; } // the list is filled from its end: ascending stages once read forwards
            __syncthreads();
        }
        if (t == 0 && njs > 0) atomicMax(&w.flags[FL_TILE_JMAX], soff[njs] - soff[0]); // the most joints one cone holds (k_joint_net_step: one per thread)
        if (t == 0) { // statistics of the tiling (tools/tile_diag.py)
            atomicMax((unsigned long long *)&w.dbg[905], (unsigned long long)nloc); atomicMax((unsigned long long *)&w.dbg[906], (unsigned long long)ncons);
            atomicAdd((unsigned long long *)&w.dbg[907], (unsigned long long)nloc); atomicAdd((unsigned long long *)&w.dbg[908], (unsigned long long)ncons);
        }
        if (bad || nloc > RP_TILE_BCAP || ncons > RP_TILE_CCAP) { if (t == 0) { atomicExch(&w.flags[FL_N_TILES], 0); w.dbg[901] = 3; } continue; } // cone over the LDS budget
        for (int h = t; h < RP_TILE_HASH; h += nt) if (hk[h] >= 0) {
            const int lid = TW_LID(hw[h]);
            w.tl_bodies[(size_t)tile * RP_TILE_BCAP + lid] = hk[h];
            if (lid >= oc) { // a halo body: its owner and this tile exchange bodies between sweeps (k_joint_net_step waits for exactly those)
                const int a = w.tl_body_tile[hk[h]], nw = RP_TILE_NBR_WORDS(w.tile_cap);
                if (a >= 0 && a != tile && a < w.tile_cap) { atomicOr(&w.tl_nbr[(size_t)tile * nw + (a >> 5)], 1u << (a & 31)); atomicOr(&w.tl_nbr[(size_t)a * nw + (tile >> 5)], 1u << (tile & 31)); }
            }
        }
        if (t == 0) w.tl_flag[(size_t)tile * 32] = 16u * (unsigned)w.flags[FL_SEQ]; // (every later launch counts from a larger FL_SEQ)
        const int nc = ncons;
        // arena indices -> tile-local ids; which tile stores the rows.  A cone of at most half the list's capacity is also packed to the
        // front of the list with every stage sorted by position: neighbouring lanes of a sweep then fetch neighbouring rows (the
        // positions of a stage follow the owners' curve order: runs of consecutive positions, one per tile that owns part of the cone)
        const bool pack = nc <= RP_TILE_CCAP / 2;
        const int tail0 = RP_TILE_CCAP - nc;
        if (pack) for (int k = t; k < nc; k += nt) cpos[k] = cons[tail0 + k].x;
        for (int s2 = t; s2 <= nall; s2 += nt) ssoff[s2] = soff[s2] - tail0; // stage begins relative to the cone's first entry
        __syncthreads();
        for (int k = t; k < nc; k += nt) {
            const int item = pack ? cpos[k] : cons[tail0 + k].x; // a manifold position, or -1 - (joint sweep index)
            int dst = tail0 + k;
            if (pack) {
                int sg = 0; while (sg + 1 < nall && ssoff[sg + 1] <= k) ++sg;
                int r = 0;
                if (item >= 0) { for (int q = ssoff[sg]; q < ssoff[sg + 1]; ++q) r += cpos[q] < item; }
                else { for (int q = ssoff[sg]; q < ssoff[sg + 1]; ++q) r += cpos[q] > item; } // (-1 - index: ascending index = descending code)
                dst = ssoff[sg] + r;
            }
            int2 ab; int code;
            if (item >= 0 && !CCHK(9, item, w.cons_cap)) continue;
            if (item >= 0) { ab = w.fk_ids[item]; code = item; }
            else { const int j = w.j_order[-1 - item]; ab = make_int2(w.j_b1[j], w.j_b2[j]); code = -1 - j; } // (the sweep wants the joint itself)
            const int first = ab.x >= 0 ? ab.x : ab.y;
            if (!CCHK(10, first, w.n_bodies)) continue;
            const int own = w.tl_body_tile[first] == tile ? 1 : 0;
            const int l1 = ab.x >= 0 ? TW_LID(hw[tile_find(Tb, ab.x)]) : -1, l2 = ab.y >= 0 ? TW_LID(hw[tile_find(Tb, ab.y)]) : -1;
            cons[dst] = make_int4(code, l1, l2, own);
        }
        if (pack) for (int s2 = t; s2 <= nall; s2 += nt) soff[s2] = ssoff[s2];
        if (t == 0) w.tl_hdr[tile] = make_int4(nloc, nc, oc, 0);
    }
}

// ---- the sweep -------------------------------------------------------------------------------------------------------------------------
// Constraint rows in registers (fetched in one go, before the first dependent instruction), solver velocities in LDS, poses (read-only
// during a sweep) fetched with the rows.
struct TileAcc {
    static constexpr bool PRELOAD = false; // the rows already sit in registers
    const DevWorld &w; const float4 *v; int pos, l1, l2, nn; bool own; float4 *Ll, *La; Xf X1, X2;
    RP_DEV float4 ld(int plane) const { return v[plane]; }
    RP_DEV void st(int plane, float4 x) const { if (own) w.C[(size_t)cplane(plane, w.c_par ^ 1) * w.cons_cap + pos] = x; } // (only mutable planes are ever stored: the other copy)
    RP_DEV int id1() const { return l1; }
    RP_DEV int id2() const { return l2; }
    RP_DEV int n() const { return nn; }
    RP_DEV Vel vel(int id) const {
        Vel r;
        if (id < 0) { r.lin = v3(0, 0, 0); r.ang = v3(0, 0, 0); } else { r.lin = v3(Ll[id]); r.ang = v3(La[id]); }
        return r;
    }
    RP_DEV void set_vel(int id, const Vel &x) const { if (id >= 0) { Ll[id] = f4(x.lin, 0.0f); La[id] = f4(x.ang, 0.0f); } }
    RP_DEV Xf xf(int id) const {
        if (id < 0) { Xf x; x.r = q4(0, 0, 0, 1); x.t = v3(0, 0, 0); return x; }
        return id == l1 ? X1 : X2;
    }
};

// one cone constraint of one stage: its rows come in with a single round trip, then cons_solve / cons_restitution over LDS velocities.
// Left to itself the compiler sinks every row load into the branch that consumes it (if (k < n) ..., if (friction) ...) and keeps only
// ~10 in flight: 5-6 exposed round trips per stage (measured: 4.0 us per relaxed stage).  The empty asm statements "use" every fetched
// component right after the loads, so nothing can sink, and the scheduling barrier asks for the loads as one group.
template <int MODE>
RP_DEV void tile_apply(const DevWorld &w, const int4 e, const int n, const int *Lg, float4 *Ll, float4 *La, bool fib, bool friction, float solved_dt) {
    const int pos = e.x;
    float4 v[CP_COUNT];
#define TL_LD(p) v[p] = w.C[(size_t)cplane(p, w.c_par) * w.cons_cap + pos]
    TL_LD(CP_H0); TL_LD(CP_H1); TL_LD(CP_H2);
#pragma unroll
    for (int k = 0; k < 4; ++k) { // (all four points: no load waits for the point count; the planes of unused points are never read)
        TL_LD(NPL(k, NP_M)); TL_LD(NPL(k, NP_A)); TL_LD(NPL(k, NP_B)); TL_LD(NPL(k, NP_C)); TL_LD(NPL(k, NP_D));
        if (MODE == MODE_RELAX) { TL_LD(NPL(k, NP_E)); TL_LD(NPL(k, NP_F)); }
    }
    TL_LD(CP_HM0); TL_LD(CP_HM1); // (always: an owner instance carries them over to the other copy even when the sweep leaves them alone)
    if (friction) {
        TL_LD(CP_H3); TL_LD(CP_H4); TL_LD(CP_H5); TL_LD(CP_H6); TL_LD(CP_H7); TL_LD(CP_H8);
        TL_LD(CP_T0); TL_LD(CP_T1); TL_LD(CP_T2); TL_LD(CP_T3); TL_LD(CP_T4); TL_LD(CP_T5); TL_LD(CP_T6); TL_LD(CP_T7);
    }
    Xf X1, X2; X1.r = q4(0, 0, 0, 1); X1.t = v3(0, 0, 0); X2 = X1;
    if (MODE == MODE_RELAX) { // (poses fetched unconditionally — of cone body 0 for a world-attached side, whose pose is never looked at: TileAcc::xf)
        TL_LD(CP_B2);
        const int g1 = Lg[e.y >= 0 ? e.y : 0], g2 = Lg[e.z >= 0 ? e.z : 0];
        X1.r = q4(w.s_rot[g1]); X1.t = v3(w.s_trans[g1]); X2.r = q4(w.s_rot[g2]); X2.t = v3(w.s_trans[g2]);
    }
#undef TL_LD
    __builtin_amdgcn_sched_group_barrier(0x020, 64, 0); // every VMEM read above as one group, ahead of whatever follows
#define TL_PIN4(p) asm volatile("" : "+v"(v[p].x), "+v"(v[p].y), "+v"(v[p].z), "+v"(v[p].w))
#define TL_PIN3(p) asm volatile("" : "+v"(v[p].x), "+v"(v[p].y), "+v"(v[p].z))
    TL_PIN4(CP_H0); TL_PIN4(CP_H1); TL_PIN4(CP_H2); TL_PIN4(CP_HM0); TL_PIN4(CP_HM1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        TL_PIN4(NPL(k, NP_M)); TL_PIN4(NPL(k, NP_A)); TL_PIN4(NPL(k, NP_C)); TL_PIN3(NPL(k, NP_D));
        if (MODE == MODE_RESTITUTION) TL_PIN4(NPL(k, NP_B)); else TL_PIN3(NPL(k, NP_B));
        if (MODE == MODE_RELAX) { TL_PIN3(NPL(k, NP_E)); TL_PIN3(NPL(k, NP_F)); }
    }
    if (friction) {
        TL_PIN4(CP_H3); TL_PIN4(CP_H4); TL_PIN4(CP_H5); TL_PIN4(CP_H6); TL_PIN3(CP_H7); TL_PIN4(CP_H8);
        TL_PIN3(CP_T0); TL_PIN3(CP_T1); TL_PIN3(CP_T2); TL_PIN3(CP_T3); TL_PIN3(CP_T4); TL_PIN3(CP_T5); TL_PIN3(CP_T6); TL_PIN3(CP_T7);
    }
    if (MODE == MODE_RELAX) {
        TL_PIN3(CP_B2);
        asm volatile("" : "+v"(X1.r.x), "+v"(X1.r.y), "+v"(X1.r.z), "+v"(X1.r.w), "+v"(X1.t.x), "+v"(X1.t.y), "+v"(X1.t.z));
        asm volatile("" : "+v"(X2.r.x), "+v"(X2.r.y), "+v"(X2.r.z), "+v"(X2.r.w), "+v"(X2.t.x), "+v"(X2.t.y), "+v"(X2.t.z));
    }
#undef TL_PIN4
#undef TL_PIN3
    const TileAcc A = {w, v, pos, e.y, e.z, n, e.w != 0, Ll, La, X1, X2};
    if (MODE == MODE_BIAS) cons_solve(w, A, false, fib, solved_dt);
    else if (MODE == MODE_RELAX) cons_solve(w, A, true, true, solved_dt);
    else cons_restitution(w, A);
    // the other copy of the mutable planes must be complete after every sweep: what this sweep did not store is carried over
    // (cons_solve stores NP_M of every live point; CP_HM0 only with friction, CP_HM1 only when it refreshes the rhs)
    if (!friction) A.st(CP_HM0, v[CP_HM0]);
    if (MODE != MODE_RELAX) A.st(CP_HM1, v[CP_HM1]);
}

// The same constraint by a PAIR of adjacent lanes (rp_lanepair.h: the island kernel's form): the even lane fetches and holds body 1's
// half of every row, the odd lane body 2's, the halves of each relative velocity meet through DPP quad permutes.  A tile stage is
// bound by the instruction issue of its one wavefront per SIMD (~2,000 instructions for a relaxed 4-point solve in the one-lane form):
// the pair halves the stream and the row loads per lane.  Same operations on the same operands as cons_solve (the island kernel is
// compared with the oracle bit for bit over the same rows): identical results.
// (SC1, cpar, rot, trans: k_tile_step — the copy of the mutable planes and of the poses that is current inside a launch that runs many
// sweeps, and write-through stores for what other tiles read behind a flag instead of a kernel boundary)
#ifndef RP_TILE_RECOMP
#define RP_TILE_RECOMP true
#endif
template <int MODE, bool SC1 = false, bool SC1LD = false>
RP_DEV void tile_apply2(const DevWorld &w, const int4 e, const int n, const bool odd, const int *Lg, float4 *Ll, float4 *La, bool friction, float solved_dt, const int cpar, const float4 *rot, const float4 *trans) {
    const int pos = e.x;
    const size_t cap = w.cons_cap;
#define TL2(pe, po) w.C[(size_t)(odd ? (po) : (pe)) * cap + pos]   // an immutable plane per lane parity
#define TLM(p) ld16_t<SC1LD>(w.C + (size_t)cplane(p, cpar) * cap, (unsigned)pos) // a mutable plane (even lane's business; both lanes fetch it)
    float4 h0 = TL2(CP_H0, CP_H0), h6 = TL2(CP_H6, CP_H6), imr = TL2(CP_H1, CP_H2), h2 = TL2(CP_H2, CP_H2), hm0 = TLM(CP_HM0), hm1 = TLM(CP_HM1);
    float4 pa[4], pc[4], pm[4], lp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { // (all four points: no load waits for the point count)
        pa[k] = TL2(NPL(k, NP_A), NPL(k, NP_B)); pm[k] = TLM(NPL(k, NP_M));
        if (!RP_TILE_RECOMP) pc[k] = TL2(NPL(k, NP_C), NPL(k, NP_D));
        if (MODE == MODE_RELAX) lp[k] = TL2(NPL(k, NP_E), NPL(k, NP_F));
    }
    float4 iiA = h0, iiB = h0, td0 = h0, td1 = h0, itd0 = h0, itd1 = h0, h7 = h0, h8 = h0, b2 = h0, xr = h0, xt = h0;
    if (friction || RP_TILE_RECOMP) { iiA = TL2(CP_H3, CP_H5); iiB = TL2(CP_H4, CP_H4); }
    if (friction) {
        td0 = TL2(CP_T0, CP_T2); td1 = TL2(CP_T1, CP_T3);
        if (!RP_TILE_RECOMP) { itd0 = TL2(CP_T4, CP_T6); itd1 = TL2(CP_T5, CP_T7); }
        h7 = TL2(CP_H7, CP_H7); h8 = TL2(CP_H8, CP_H8);
    }
    const int lid = odd ? e.z : e.y;
    if (MODE == MODE_RELAX) { b2 = TL2(CP_B2, CP_B2); const int g = Lg[lid >= 0 ? lid : 0]; xr = ld16_t<SC1LD>(rot, (unsigned)g); xt = ld16_t<SC1LD>(trans, (unsigned)g); }
#undef TL2
#undef TLM
    __builtin_amdgcn_sched_group_barrier(0x020, 48, 0); // every VMEM read above as one group
#define P4(r_) asm volatile("" : "+v"((r_).x), "+v"((r_).y), "+v"((r_).z), "+v"((r_).w))
    P4(h0); P4(h6); P4(imr); P4(h2); P4(hm0); P4(hm1);
#pragma unroll
    for (int k = 0; k < 4; ++k) { P4(pa[k]); if (!RP_TILE_RECOMP) P4(pc[k]); P4(pm[k]); if (MODE == MODE_RELAX) P4(lp[k]); }
    if (friction || RP_TILE_RECOMP) { P4(iiA); P4(iiB); }
    if (friction) { P4(td0); P4(td1); if (!RP_TILE_RECOMP) { P4(itd0); P4(itd1); } P4(h7); P4(h8); }
    if (MODE == MODE_RELAX) { P4(b2); P4(xr); P4(xt); }
#undef P4
    IslSide h;
    h.odd = odd; h.id = lid; h.n = n; h.cids = 0;
    h.dir = v3(h0); h.t0 = v3(h6); h.t1 = cross(h.dir, h.t0);
    h.im = v3(imr);
    { const V3 dim = cmul(h.dir, h.im); h.sdim = odd ? -dim : dim; }
    // RP_TILE_RECOMP: the ii_torque_dir rows (NP_C / NP_D of every point, T4 .. T7) are not fetched — sym_mul of the inertia (H3 .. H5, fetched
    // for the twist row anyway) with the torque_dir rows is what cons_generate stored there, operand for operand (rp_constraint.h:138-172):
    // 6 of a lane's 33 row loads in a relaxed stage, 4 of 18 (+ the two inertia rows) in a biased one, for ~90 multiply-adds.  The builder
    // distance NP_C carried in its spare word comes from the copies cons_generate leaves in T0.w, T1.w, B2.w, H7.w.
    const Sym3 ii = odd ? Sym3{iiB.z, iiB.w, iiA.x, iiA.y, iiA.z, iiA.w} : Sym3{iiA.x, iiA.y, iiA.z, iiA.w, iiB.x, iiB.y};
    {
        const V3 tw = sym_mul(ii, h.dir);
        h.stw = odd ? -tw : tw;
    }
    h.td0 = v3(td0); h.td1 = v3(td1);
    if (RP_TILE_RECOMP) { h.itd0 = sym_mul(ii, h.td0); h.itd1 = sym_mul(ii, h.td1); } else { h.itd0 = v3(itd0); h.itd1 = v3(itd1); }
    h.mu = h0.w; h.twist_r = imr.w; h.rhs_wo0 = h6.w; h.rhs_wo1 = h7.x; h.k11 = h7.y; h.k22 = h7.z; h.k12 = h2.w * 0.5f;
    h.inv_det = rp_inv(h.k11 * h.k22 - h.k12 * h.k12);
    h.td[0] = h8.x; h.td[1] = h8.y; h.td[2] = h8.z; h.td[3] = h8.w;
    h.tw_imp = hm0.x; h.tw_acc = hm0.y; h.t_imp0 = hm0.z; h.t_imp1 = hm0.w; h.t_acc0 = hm1.x; h.t_acc1 = hm1.y; h.t_rhs0 = hm1.z; h.t_rhs1 = hm1.w;
    h.tb0 = 0.0f; h.tb1 = 0.0f; h.cfm_factor = 0.0f; h.erp_inv_dt = 0.0f;
    Xf x; x.r = q4(xr); x.t = v3(xt);
    if (lid < 0) { x.r = q4(0, 0, 0, 1); x.t = v3(0, 0, 0); }
    const V3 tangent_delta = v3(b2) * solved_dt;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        SidePoint &q = h.P[k];
        q.pa = v3(pa[k]); q.r = pa[k].w; q.seed = 0.0f;
        if (RP_TILE_RECOMP) { q.pc = sym_mul(ii, q.pa); q.d0 = k == 0 ? td0.w : (k == 1 ? td1.w : (k == 2 ? b2.w : h7.w)); } // (d0: read by the relaxed form only, whose even lane holds T0 / T1)
        else { q.pc = v3(pc[k]); q.d0 = pc[k].w; }
        q.rhs = pm[k].x; q.cfm = pm[k].y; q.lam = pm[k].z; q.acc = pm[k].w; q.rhsR = 0.0f; q.rhsB = 0.0f; q.cfmB = 1.0f;
        if (MODE == MODE_RELAX) { // refresh_rhs_wo_bias (:529-554): p1 = T1 lp1 + delta on the even lane, p2 = T2 lp2 on the odd one
            V3 pw = xf_tp(x, v3(lp[k]));
            pw = sel(odd, pw, pw + tangent_delta);
            const V3 p2 = dppv<DPP_FROM_ODD>(pw);
            const float dist = q.d0 + dot(pw - p2, h.dir);
            q.rhsR = rp_max(dist, 0.0f) * w.prm.inv_dt_sub;
        }
    }
    IslLds L; L.lin = Ll; L.ang = La; L.rot = nullptr; L.trans = nullptr; L.E = nullptr; L.F = nullptr; L.B0 = nullptr; L.B1 = nullptr;
    isl_solve(h, L, MODE == MODE_RELAX, friction);
    if (!odd && e.w != 0) { // the owner's even lane stores the manifold's mutable planes into the other copy (all of them: see tile_apply)
        const int par = cpar ^ 1;
#define TLS(p, v) do { if (SC1) st16_sc1(w.C + (size_t)cplane(p, par) * cap, (unsigned)pos, v); else w.C[(size_t)cplane(p, par) * cap + pos] = v; } while (0)
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (k >= n) break; const SidePoint &q = h.P[k]; TLS(NPL(k, NP_M), make_float4(q.rhs, q.cfm, q.lam, q.acc)); }
        TLS(CP_HM0, make_float4(h.tw_imp, h.tw_acc, h.t_imp0, h.t_imp1));
        TLS(CP_HM1, make_float4(h.t_acc0, h.t_acc1, h.t_rhs0, h.t_rhs1));
#undef TLS
    }
}

// one cone JOINT of one joint stage: joint_solve_one_t of rp_joints.h (the rows of k_joint_update, [remove bias] [warm start] solve)
// over LDS velocities; the two words a sweep changes per row live in DevWorld::jm (read copy c_par, owner instances write the other)
struct TileJointIO {
    float4 *Ll, *La; int l1, l2, out; bool own;
    RP_DEV void load_vel(int side, int b, V3 &l, V3 &a) const { const int lid = side ? l2 : l1; l = v3(Ll[lid]); a = v3(La[lid]); }
    RP_DEV void store_vel(int side, int b, V3 l, V3 a) const { const int lid = side ? l2 : l1; Ll[lid] = f4(l, 0.0f); La[lid] = f4(a, 0.0f); }
    RP_DEV int jm_out(const DevWorld &) const { return out; }
    RP_DEV bool jm_store() const { return own; }
};
// Everything a joint solve reads that is not a velocity: the axis masks (-> row count), the inverse masses and the first three rows,
// fetched in ONE round trip whatever the count turns out to be (every joint owns twelve row slots; a spherical joint — b3d_joint_grid —
// has exactly three).  None of it changes during a sweep (the rows are rebuilt between sweeps, the sweep's own words go to the other
// copy of DevWorld::jm), so the fetch of a thread's NEXT joint stage is issued a whole stage ahead.
struct TileJointPre { int locked, limited, motor, b1, b2; V3 im1, im2; JointRowsT<3> R; };
RP_DEV void tile_joint_fetch(const DevWorld &w, const int4 e, const int *Lg, TileJointPre &P) {
    const int j = -1 - e.x;
    P.locked = w.j_locked[j]; P.limited = w.j_limited[j]; P.motor = w.j_motor[j]; P.b1 = e.y >= 0 ? Lg[e.y] : -1; P.b2 = e.z >= 0 ? Lg[e.z] : -1;
    P.im1 = v3(JRP(JR_IM1, j)); P.im2 = v3(JRP(JR_IM2, j));
#pragma unroll
    for (int q = 0; q < 3; ++q) jrow_load(w, j, q, P.R.c[q]);
}
// The first biased sweep of a substep in a world whose joints are all spherical (DevWorld::joints_spherical) rebuilds the rows itself:
// JointConstraintBuilder::update straight from the poses (14 loads against the 20 of the built rows), the rows go from the registers
// into the solve, the owner instance stores them for the sweeps that follow.  k_ws_prepare then has no joint work: in a world without
// contacts (b3d_joint_grid) it becomes an empty launch.  (The words a sweep changes: the rebuilt impulse seeds the solve, whose result
// goes to the other copy of jm as ever; the copy being read is NOT written — halo instances of other tiles still read it.)
struct TilePoseIO {
    const DevWorld &w; int b1, b2; // the joint's bodies (arena indices, -1 = world-attached side) from the cone entry's tile-local ids
    RP_DEV void bodies(const DevWorld &, int, int &o1, int &o2) const { o1 = b1; o2 = b2; }
    RP_DEV void pose(int side, int b, Pose &p) const { p.r = q4(w.s_rot[b]); p.t = v3(w.s_trans[b]); }
};
struct TileJointBuild {
    TileJointPre &P; bool own;
    RP_DEV void take3(const DevWorld &w, int j, JointRow (&r3)[3], V3 im1, V3 im2) const {
        P.im1 = im1; P.im2 = im2;
#pragma unroll
        for (int k = 0; k < 3; ++k) P.R.c[k] = r3[k];
        if (own) {
#pragma unroll
            for (int k = 0; k < 3; ++k) jrow_store_planes(w, j, k, r3[k]);
            JRP(JR_IM1, j) = f4(im1, 0.0f); JRP(JR_IM2, j) = f4(im2, 0.0f);
        }
    }
};
RP_DEV void tile_joint_build(const DevWorld &w, const int4 e, const int *Lg, int substep, TileJointPre &P) {
    P.locked = 0x7; P.limited = 0; P.motor = 0; P.b1 = e.y >= 0 ? Lg[e.y] : -1; P.b2 = e.z >= 0 ? Lg[e.z] : -1; // (no dependent load of j_b1 / j_b2)
    const TilePoseIO io = {w, P.b1, P.b2};
    const TileJointBuild sink = {P, e.w != 0};
    joint_update_one_t<TilePoseIO, TileJointBuild, true>(w, io, -1 - e.x, substep, sink);
}
// a prepared joint parked in LDS, one word per field and slot ([field][slot]: neighbouring threads, neighbouring banks)
#define RP_TILE_JX 64
#define TJP_WORDS (5 + 6 + 3 * 22)
RP_DEV void tile_joint_park(float *Lj, int slot, const TileJointPre &P) {
    int f = 0;
#define TJ_PUT(x) Lj[(f++) * RP_TILE_JX + slot] = (x)
    TJ_PUT(__int_as_float(P.locked)); TJ_PUT(__int_as_float(P.limited)); TJ_PUT(__int_as_float(P.motor)); TJ_PUT(__int_as_float(P.b1)); TJ_PUT(__int_as_float(P.b2));
    TJ_PUT(P.im1.x); TJ_PUT(P.im1.y); TJ_PUT(P.im1.z); TJ_PUT(P.im2.x); TJ_PUT(P.im2.y); TJ_PUT(P.im2.z);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const JointRow &c = P.R.c[k];
        TJ_PUT(c.lin_jac.x); TJ_PUT(c.lin_jac.y); TJ_PUT(c.lin_jac.z); TJ_PUT(c.ang_jac1.x); TJ_PUT(c.ang_jac1.y); TJ_PUT(c.ang_jac1.z);
        TJ_PUT(c.ang_jac2.x); TJ_PUT(c.ang_jac2.y); TJ_PUT(c.ang_jac2.z); TJ_PUT(c.ii1.x); TJ_PUT(c.ii1.y); TJ_PUT(c.ii1.z); TJ_PUT(c.ii2.x); TJ_PUT(c.ii2.y); TJ_PUT(c.ii2.z);
        TJ_PUT(c.impulse); TJ_PUT(c.inv_lhs); TJ_PUT(c.rhs); TJ_PUT(c.rhs_wo_bias); TJ_PUT(c.cfm_gain); TJ_PUT(c.bmin); TJ_PUT(c.bmax);
    }
#undef TJ_PUT
}
RP_DEV void tile_joint_unpark(const float *Lj, int slot, TileJointPre &P) {
    int f = 0;
#define TJ_GET() Lj[(f++) * RP_TILE_JX + slot]
    P.locked = __float_as_int(TJ_GET()); P.limited = __float_as_int(TJ_GET()); P.motor = __float_as_int(TJ_GET()); P.b1 = __float_as_int(TJ_GET()); P.b2 = __float_as_int(TJ_GET());
    P.im1.x = TJ_GET(); P.im1.y = TJ_GET(); P.im1.z = TJ_GET(); P.im2.x = TJ_GET(); P.im2.y = TJ_GET(); P.im2.z = TJ_GET();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        JointRow &c = P.R.c[k];
        c.lin_jac.x = TJ_GET(); c.lin_jac.y = TJ_GET(); c.lin_jac.z = TJ_GET(); c.ang_jac1.x = TJ_GET(); c.ang_jac1.y = TJ_GET(); c.ang_jac1.z = TJ_GET();
        c.ang_jac2.x = TJ_GET(); c.ang_jac2.y = TJ_GET(); c.ang_jac2.z = TJ_GET(); c.ii1.x = TJ_GET(); c.ii1.y = TJ_GET(); c.ii1.z = TJ_GET(); c.ii2.x = TJ_GET(); c.ii2.y = TJ_GET(); c.ii2.z = TJ_GET();
        c.impulse = TJ_GET(); c.inv_lhs = TJ_GET(); c.rhs = TJ_GET(); c.rhs_wo_bias = TJ_GET(); c.cfm_gain = TJ_GET(); c.cfm_coeff = 0.0f; c.bmin = TJ_GET(); c.bmax = TJ_GET();
    }
#undef TJ_GET
}
RP_DEV void tile_apply_joint(const DevWorld &w, const int4 e, TileJointPre &P, float4 *Ll, float4 *La, bool wo_bias, bool warmstart) {
    const TileJointIO io = {Ll, La, e.y, e.z, w.c_par ^ 1, e.w != 0};
    joint_solve_fetched<TileJointIO, 3>(w, io, -1 - e.x, P.b1, P.b2, joint_row_count(P.locked, P.limited, P.motor), P.im1, P.im2, P.R, wo_bias, warmstart);
}

// fuse bit 2: the sweep rebuilds the rows of its (spherical) joints itself — tile_joint_build; the substep index rides in fuse >> 8
// fuse bit 0: the sweep starts the substep — every cone body is incremented and warm-started on its way into LDS (k_increment_ws folded
//             in; halo bodies redundantly);  bit 1: the sweep ends the biased phase — owned bodies are integrated on their way out
//             (k_integrate folded in): velocities AND poses then go to the other buffers (t_lin / t_ang / t_rot / t_trans).
// LP: two lanes per manifold (tile_apply2) instead of one (tile_apply); RP_TILE_LANES=1 keeps the one-lane form
template <int MODE, bool LP>
__global__ void __launch_bounds__(RP_TILE_THREADS) k_tile_sweep(DevWorld w, int friction_in_bias, float solved_dt, int fuse, int joint_warmstart) {
    if (lean_dead(w)) return; // (rp_world.h "lean step graphs")
    const int NT = w.flags[FL_N_TILES];
    const int t = threadIdx.x, nt = blockDim.x;
    const bool fib = friction_in_bias != 0;
    if (NT <= 0) {
        // no valid tiling (the host planned on a stale hint): the whole sweep in workgroup 0, then the result moves to the other buffers
        if (blockIdx.x != 0) return;
        if (fuse & 1) {
            for (int i = t; i < w.n_bodies; i += nt) if (global_body(w, i)) { V3 lin, ang; body_increment_ws(w, i, lin, ang); w.s_lin[i] = f4(lin, 0.0f); w.s_ang[i] = f4(ang, 0.0f); }
            __threadfence(); __syncthreads();
        }
        if (MODE == MODE_BIAS && (fuse & 4)) { // the rows this sweep was to rebuild on its tiles
            for (int j = t; j < w.n_joints; j += nt) if (joint_live(w, j)) { const PlainBodyIO io = {w}; joint_update_one_t<PlainBodyIO, JointRowsToPlanes, true>(w, io, j, fuse >> 8); } // (all spherical: fuse bit 2)
            __threadfence(); __syncthreads();
        }
        if (MODE != MODE_RESTITUTION) joint_tail_sweep(w, 0, MODE == MODE_RELAX, joint_warmstart != 0); // every joint before any contact
        tail_sweep<MODE, false>(w, 0, fib, solved_dt);
        if (fuse & 2) {
            for (int i = t; i < w.n_bodies; i += nt) if (global_body(w, i)) g_body_integrate(w, i);
            __threadfence(); __syncthreads();
        }
        for (int i = t; i < w.n_bodies; i += nt) if (global_body(w, i)) {
            w.t_lin[i] = w.s_lin[i]; w.t_ang[i] = w.s_ang[i];
            if (fuse & 2) { w.t_rot[i] = w.s_rot[i]; w.t_trans[i] = w.s_trans[i]; }
        }
        int M = w.flags[FL_N_CONS]; if (M > w.cons_cap) M = w.cons_cap;
        for (int pos = t; pos < M; pos += nt)
            for (int m = 0; m < CP_SHADOW_COUNT; ++m) {
                const int plane = m < 4 ? NPL(m, NP_M) : (m == 4 ? CP_HM0 : CP_HM1);
                w.C[(size_t)cplane(plane, w.c_par ^ 1) * w.cons_cap + pos] = w.C[(size_t)cplane(plane, w.c_par) * w.cons_cap + pos];
            }
        if (w.jm) for (size_t k = t; k < (size_t)JR_MAX_ROWS * w.n_joints; k += nt) w.jm[(size_t)(w.c_par ^ 1) * JR_MAX_ROWS * w.n_joints + k] = w.jm[(size_t)w.c_par * JR_MAX_ROWS * w.n_joints + k];
        return;
    }
    __shared__ float4 Ll[RP_TILE_BCAP], La[RP_TILE_BCAP];
    __shared__ int Lg[RP_TILE_BCAP];
    __shared__ float Lj[TJP_WORDS * RP_TILE_JX]; // the prepared second joints of the first RP_TILE_JX threads (tile_joint_park)
    __shared__ int Soff[RP_TILE_STAGES + 4];
    const int njs = tile_joint_stages(w); // joint stages come first in a sweep
    const int nst = njs + w.flags[FL_N_STAGES];
    const bool friction = MODE == MODE_RELAX || (MODE == MODE_BIAS && fib);
#ifdef RP_TILE_PROFILE // thread 0 of tile 0 accumulates wall-clock ticks (10 ns) per phase into dbg[920 + 24 * MODE ..] (tools/tile_diag.py)
#define TP_STAMP(k) do { if (blockIdx.x == 0 && t == 0) { const long long n_ = (long long)wall_clock64(); w.dbg[920 + 24 * MODE + (k)] += n_ - tp_; tp_ = n_; } } while (0)
    long long tp_ = (long long)wall_clock64();
    const long long tp0_ = tp_;
#else
#define TP_STAMP(k) do { } while (0)
#endif
    // (XCD-aware placement, as in k_tile_step below: workgroup b runs on XCD b % 8 — every XCD takes a contiguous run of the curve, so tiles
    // that share halo rows share an L2; grids smaller than the tiling keep the strided loop)
    const int per_xcd_ = (NT + 7) >> 3;
    const bool xmap_ = 8 * per_xcd_ <= (int)gridDim.x;
    const int tile0_ = xmap_ ? (((int)blockIdx.x >> 3) < per_xcd_ ? ((int)blockIdx.x & 7) * per_xcd_ + ((int)blockIdx.x >> 3) : NT) : (int)blockIdx.x;
    for (int tile = tile0_; tile < NT; tile += xmap_ ? NT : (int)gridDim.x) {
        __syncthreads();
        // one round trip for everything that only depends on the tile: the header, the body list (every thread asks for its
        // RP_TILE_BCAP / 256 slots whatever the count turns out to be: the list is allocated in full), the stage offsets and — cones
        // packed to the front of their list (k_tiles_cones: all but the very largest) — the thread's first list entry
        const int4 *cons = w.tl_cons + (size_t)tile * RP_TILE_CCAP;
        const int4 hdr = w.tl_hdr[tile];
        int gl[RP_TILE_BCAP / RP_TILE_THREADS];
#pragma unroll
        for (int k = 0; k < RP_TILE_BCAP / RP_TILE_THREADS; ++k) gl[k] = w.tl_bodies[(size_t)tile * RP_TILE_BCAP + t + k * RP_TILE_THREADS];
        const int4 e_first = cons[t];
        for (int s = t; s <= nst + 3; s += nt) Soff[s] = w.tl_soff[(size_t)tile * (RP_TILE_STAGES + 1) + (s < nst ? s : nst)];
        const int nb = hdr.x, n_owned = hdr.z; // (tile-local ids [0, n_owned) are the bodies the tile owns: k_tiles_cones inserts them first)
#pragma unroll
        for (int k = 0; k < RP_TILE_BCAP / RP_TILE_THREADS; ++k) {
            const int l = t + k * RP_TILE_THREADS;
            if (l >= nb) break;
            const int g = gl[k];
            Lg[l] = g;
            if (fuse & 1) { V3 lin, ang; body_increment_ws(w, g, lin, ang); Ll[l] = f4(lin, 0.0f); La[l] = f4(ang, 0.0f); }
            else { Ll[l] = w.s_lin[g]; La[l] = w.s_ang[g]; }
        }
        __syncthreads();
        // the list entry (and point count) of a thread's next stage is fetched while it works on the current one: a stage then costs one
        // round trip (the rows) instead of two (branch-free: a load inside a conditional block is waited for at the end of the block).
        // (A deeper pipeline — entries two stages ahead, the rows of the next JOINT stage one ahead — was built and measured: the
        // registers it holds across the contact stages cost b3d_large_pyramid 0.544 -> 0.623 ms and the joint stages gained nothing.)
        const int my = LP ? (t >> 1) : t, per = LP ? (nt >> 1) : nt; // a stage's entries go to lanes (one-lane form) or lane pairs
        const bool odd = LP && (t & 1);
        // (Asking for the rows of stage s + 1 a stage ahead — one dword per row through global_load_lds into a junk LDS slot — was built
        // and measured: every stage got ~50 % SLOWER.  The sweeps already pull 2.3-4.6 TB/s from HBM (PMC FETCH_SIZE: 100 MB per relaxed
        // launch in 43 us): what bounds a stage is the traffic the tiling demands — rows x 1.95 instances — not the latency of one miss.)
        // ---- the joint stages (they come first in a sweep) ----
        // A cone that holds at most one joint per thread (b3d_joint_grid: ~200 per tile) hands thread k the k-th joint entry for the whole
        // sweep: what a joint solve reads besides velocities — the rows, rebuilt from the poses by the first biased sweep of a substep
        // (~600 dependent operations) or fetched (20 loads) — is independent of the sweep's velocities, so ALL of a tile's joints are
        // prepared at once, before the first stage, by as many lanes as there are joints; a stage is then LDS velocities + three row
        // solves.  (Round 3 prepared inside the stage: 45 busy lanes and a 2 us dependent chain in every one of the five stages.)
        // Larger cones keep the per-stage form.
        if (njs > 0) {
            const int j0 = Soff[0], nje = Soff[njs] - j0;
            if (nje <= nt + RP_TILE_JX) { // (tile-uniform)
                // (a cone of a few more joints than threads — b3d_joint_grid's tiles hold 200-270 — gives its first threads a second
                // joint, prepared in a second round and parked in LDS until its stage: the cone sizes depend on where the Morton cuts
                // fall, and a tile that dropped to the per-stage form cost the whole launch 5 us)
                int4 je = make_int4(0, 0, 0, 0), je2 = je; int jstage = -1, jstage2 = -1; TileJointPre JP;
                TP_STAMP(0);
                if (t < nje - nt) {
                    je2 = cons[j0 + nt + t];
                    jstage2 = 0; while (Soff[jstage2 + 1] - j0 <= nt + t) ++jstage2;
                    TileJointPre P2;
                    if (MODE == MODE_BIAS && (fuse & 4)) tile_joint_build(w, je2, Lg, fuse >> 8, P2);
                    else tile_joint_fetch(w, je2, Lg, P2);
                    tile_joint_park(Lj, t, P2);
                }
                if (t < nje) {
                    je = j0 == 0 ? e_first : cons[j0 + t];
                    jstage = 0; while (Soff[jstage + 1] - j0 <= t) ++jstage;
                    if (MODE == MODE_BIAS && (fuse & 4)) tile_joint_build(w, je, Lg, fuse >> 8, JP);
                    else tile_joint_fetch(w, je, Lg, JP);
                }
#ifdef RP_TILE_PROFILE
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                TP_STAMP(3); // every joint of the cone prepared (rows rebuilt / fetched)
                for (int s = 0; s < njs; ++s) {
                    if (jstage == s) tile_apply_joint(w, je, JP, Ll, La, MODE == MODE_RELAX, joint_warmstart != 0);
                    if (jstage2 == s) { // (a colour is body-disjoint: the thread's two joints of one stage do not share a body)
                        TileJointPre P2; tile_joint_unpark(Lj, t, P2);
                        tile_apply_joint(w, je2, P2, Ll, La, MODE == MODE_RELAX, joint_warmstart != 0);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    TP_STAMP(4 + (s < 16 ? s : 16));
                }
            } else {
                TP_STAMP(0);
                for (int s = 0; s < njs; ++s) {
                    const int end = Soff[s + 1];
                    for (int i = Soff[s] + (LP ? (t >> 1) : t); i < end; i += (LP ? (nt >> 1) : nt)) {
                        if (LP && (t & 1)) continue; // one lane per joint (the even lane of a pair)
                        const int4 e = cons[i];
                        TileJointPre P;
                        if (MODE == MODE_BIAS && (fuse & 4)) tile_joint_build(w, e, Lg, fuse >> 8, P);
                        else tile_joint_fetch(w, e, Lg, P);
                        tile_apply_joint(w, e, P, Ll, La, MODE == MODE_RELAX, joint_warmstart != 0);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    TP_STAMP(4 + (s < 16 ? s : 16));
                }
            }
        } else TP_STAMP(0);
        // ---- the contact stages ----
        int4 e_next; int n_next; bool have_next;
        { const int i0 = Soff[njs] + my; have_next = i0 < Soff[njs + 1]; e_next = cons[have_next ? i0 : 0]; n_next = w.k_n[e_next.x > 0 ? e_next.x : 0]; }
        for (int s = njs; s < nst; ++s) {
            const int end = Soff[s + 1];
            int i = Soff[s] + my;
            int4 e = e_next; int n = n_next;
            bool have = have_next;
            { const int i1 = end + my; have_next = i1 < Soff[s + 2]; e_next = cons[have_next ? i1 : 0]; } // (Soff[nst + 1] = Soff[nst]: nothing behind the last stage)
            while (have) {
                if (LP) tile_apply2<MODE>(w, e, n, odd, Lg, Ll, La, friction, solved_dt, w.c_par, w.s_rot, w.s_trans);
                else tile_apply<MODE>(w, e, n, Lg, Ll, La, fib, friction, solved_dt);
                i += per; have = i < end;
                if (have) { e = cons[i]; n = w.k_n[e.x]; }
            }
            n_next = w.k_n[e_next.x > 0 ? e_next.x : 0];
            // the barrier orders the LDS velocities only: a thread's row stores may stay in flight (nothing reads them before the next kernel)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TP_STAMP(4 + (s < 16 ? s : 16));
        }
        for (int l = t; l < n_owned; l += nt) {
            const int g = Lg[l];
            if (fuse & 2) {
                V3 lin = v3(Ll[l]), ang = v3(La[l]), trans = v3(w.s_trans[g]); Q4 rot = q4(w.s_rot[g]);
                body_integrate(w, w.b_flags[g], lin, ang, rot, trans);
                w.t_lin[g] = f4(lin, 0.0f); w.t_ang[g] = f4(ang, 0.0f); w.t_rot[g] = f4(rot); w.t_trans[g] = f4(trans, 0.0f);
            } else { w.t_lin[g] = Ll[l]; w.t_ang[g] = La[l]; }
        }
#ifdef RP_TILE_PROFILE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TP_STAMP(1);
        if (blockIdx.x == 0 && t == 0) w.dbg[920 + 24 * MODE + 2] += 1;
        if (t == 0 && tile < 256) { w.dbg[(MODE == MODE_BIAS ? 300 : 560) + tile] += (long long)wall_clock64() - tp0_; if (MODE == MODE_BIAS && tile < 70) w.dbg[830 + tile] = Soff[njs] - Soff[0]; } // every tile: its whole sweep (tools/tile_diag.py)
#endif
    }
#undef TP_STAMP
}

// ---- b3d_joint_grid: the TGS loop of a net of spherical joints as ONE launch, the joints in registers from the first substep to the last
// Measured on the sweeps above (tools/tile_diag.py, every tile stamped): of the 12.4 / 10.5 us a tile spends in a biased / relaxed
// sweep only 4.5 are its colour stages — the rest is fetching what the previous launch of the same tile already held (its lists, its
// bodies, its joints' rows and impulses), and the launch itself takes 22.6 / 16.6 us.  Here a workgroup keeps its tile for the whole
// step: the lists are read once, every cone joint lives in ONE thread's registers (rows rebuilt from the poses at the head of a substep,
// impulses carried from sweep to sweep: a cone joint is solved on correct inputs in every sweep, so every instance holds the owner's
// impulse), and between two sweeps only what another tile may need crosses memory — the owned bodies' velocities (poses behind the
// biased sweep) go out, a grid barrier (rp_gridbar.h), the cone bodies' come in.  The arithmetic is the sweeps': body_increment,
// joint_update_one_t, joint_solve_fetched, body_integrate on the same operands in the same order.  Row planes are not written (nothing
// reads them before the next rebuild); the impulses and right-hand sides go to DevWorld::jm once, at the end, for the write-back.
// Where it runs: bare lean graphs (DevWorld::lean bits 1 and 2; lean_dead verifies no manifold, no LDS island, a valid tiling that fits
// the grid, no contact stage, no cone above RP_JN_THREADS joints), all joints spherical, one PGS and one stabilisation iteration.
struct JnPoseIO {
    const float4 *rot, *trans; int b1, b2;
    RP_DEV void bodies(const DevWorld &, int, int &o1, int &o2) const { o1 = b1; o2 = b2; }
    RP_DEV void pose(int side, int b, Pose &p) const { p.r = q4(rot[b]); p.t = v3(trans[b]); }
};
struct JnSink {
    TileJointPre &P;
    RP_DEV void take3(const DevWorld &, int, JointRow (&r3)[3], V3 im1, V3 im2) const {
        P.im1 = im1; P.im2 = im2;
#pragma unroll
        for (int k = 0; k < 3; ++k) P.R.c[k] = r3[k];
    }
};
struct JnVelIO { // LDS velocities; the sweep's words stay in the registers (no store)
    float4 *Ll, *La; int l1, l2;
    RP_DEV void load_vel(int side, int b, V3 &l, V3 &a) const { const int lid = side ? l2 : l1; l = v3(Ll[lid]); a = v3(La[lid]); }
    RP_DEV void store_vel(int side, int b, V3 l, V3 a) const { const int lid = side ? l2 : l1; Ll[lid] = f4(l, 0.0f); La[lid] = f4(a, 0.0f); }
    RP_DEV int jm_out(const DevWorld &w) const { return w.c_par; }
    RP_DEV bool jm_store() const { return false; }
};
// What one tile hands to the others between two sweeps (its owned bodies' velocities, poses behind the biased sweep) is stored WRITE-THROUGH
// (16-byte sc1 stores: nothing of it stays dirty in the XCD's L2), so the barrier needs no release fence — the write-back of an L2 with
// freshly dirtied lines is what made rp_gridbar.h's barrier cost ~10 us here (MI355X guide, Guideline 16 R1: sc1 payload, every storing
// wave drains, ONE lane arrives; the consumer polls relaxed, ONE agent acquire, then plain loads).
typedef float jn_v4f __attribute__((ext_vector_type(4)));
RP_DEV void jn_store_sc1(float4 *p, float4 v) {
    const jn_v4f x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(x) : "memory");
}
// Between two sweeps a tile waits for the tiles it exchanges bodies with — not for the grid: every tile publishes "sweep k done" in a
// word of its own (tl_flag, one 128-byte line each; the value counts from 16 x FL_SEQ, which no launch shares with another) and polls
// the words of its neighbours (tl_nbr: who holds a body I own, whose bodies I hold; symmetric, so a tile is never more than one sweep
// ahead of a neighbour — the other copy of the double buffers is safe to overwrite).  handoff-flag of the MI355X guide (sc1 payload,
// every storing wave drains, one lane stores the flag; relaxed polls, ONE agent acquire) in place of a grid barrier: ~7 us -> ~2 us
// per sweep boundary, and a slow tile only holds up its neighbours.
// true = the launch is dead: a neighbour never arrived (a workgroup that is not resident — another process or stream holds CUs).  Nothing
// is committed by this kernel (the write-back is a launch of its own and looks at lean_dead): FL_JN_TIMEOUT makes the step die like any
// lean step, the full graph resumes it, and settle() stops planning the joint-net form for this world (rp_counters.joint_net_disabled)
template <bool ACQ = true>
RP_DEV bool jn_sync(const DevWorld &w, int tile, unsigned epoch, const int *Lnbr, int nn) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every wave: its write-through stores have completed
    __syncthreads();
    int dead = 0;
    if (threadIdx.x == 0) __hip_atomic_store(&w.tl_flag[(size_t)tile * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)threadIdx.x < nn) {
        const unsigned *f = &w.tl_flag[(size_t)Lnbr[threadIdx.x] * 32];
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 21)) { __hip_atomic_store(&w.flags[FL_JN_TIMEOUT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = 1; break; }
            if ((spins & 1023u) == 0 && __hip_atomic_load(&w.flags[FL_JN_TIMEOUT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { dead = 1; break; } // (some tile gave up: so does this one)
        }
    }
    __syncthreads();
    if (ACQ && threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // (!ACQ: the caller reads what its neighbours published with sc1 loads)
    return __syncthreads_or(dead) != 0;
}
__global__ void __launch_bounds__(RP_JN_THREADS) k_joint_net_step(DevWorld w, int joint_warmstart, int stall_tile) {
    if (lean_dead(w)) return; // (the same answer in every workgroup: nothing it reads changes while a lean graph runs)
    if ((int)blockIdx.x == stall_tile) return; // (test hook, testing build only: a workgroup that never becomes resident)
    __shared__ float4 Ll[RP_TILE_BCAP], La[RP_TILE_BCAP];
    __shared__ int Lg[RP_TILE_BCAP];
    __shared__ int Soff[RP_TILE_STAGES + 2];
    const int t = threadIdx.x, nt = blockDim.x;
    // (XCD-aware placement: workgroup b runs on XCD b % 8; every XCD takes a contiguous run of the curve — the tiles a tile exchanges bodies with sit behind the same L2)
    const int ntiles_ = w.flags[FL_N_TILES], per_xcd_ = (ntiles_ + 7) >> 3;
    const bool xmap_ = 8 * per_xcd_ <= (int)gridDim.x; // (A/B on one box, 3,000 steps of b3d_joint_grid twice: 10,993 / 11,045 steps/s with the map, 10,841 / 10,770 without)
    if (xmap_ && ((int)blockIdx.x >> 3) >= per_xcd_) return; // (workgroups beyond the map)
    const int tile = xmap_ ? ((int)blockIdx.x & 7) * per_xcd_ + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int njs = tile_joint_stages(w), substeps = w.prm.num_substeps;
#ifdef RP_TILE_PROFILE // thread 0 of tile JN_PROF_TILE accumulates wall-clock ticks (10 ns) per phase into dbg[260 ..] (tools/tile_diag.py)
#define JN_STAMP(k) do { if (blockIdx.x == 0 && t == 0) { const long long n_ = (long long)wall_clock64(); w.dbg[260 + (k)] += n_ - jp_; jp_ = n_; } } while (0)
    long long jp_ = (long long)wall_clock64();
#else
#define JN_STAMP(k) do { } while (0)
#endif
    if (tile >= w.flags[FL_N_TILES]) return; // (workgroups beyond the tiling: nobody waits for them)
    const bool have = true;
    __shared__ int Lnbr[RP_JN_THREADS];
    __shared__ int nn_sh;
    if (t == 0) nn_sh = 0;
    const unsigned e0 = 16u * (unsigned)w.flags[FL_SEQ]; // (FL_SEQ moves once per step graph, behind this launch: the same in every workgroup)
    int nb = 0, n_owned = 0;
    __syncthreads();
    if (t < w.tile_cap && ((w.tl_nbr[(size_t)tile * RP_TILE_NBR_WORDS(w.tile_cap) + (t >> 5)] >> (t & 31)) & 1u)) Lnbr[atomicAdd(&nn_sh, 1)] = t; // (lean_dead: at most RP_JN_THREADS tiles)
    if (have) {
        const int4 hdr = w.tl_hdr[tile];
        nb = hdr.x; n_owned = hdr.z;
        for (int s = t; s <= njs; s += nt) Soff[s] = w.tl_soff[(size_t)tile * (RP_TILE_STAGES + 1) + s];
        for (int l = t; l < nb; l += nt) Lg[l] = w.tl_bodies[(size_t)tile * RP_TILE_BCAP + l];
    }
    __syncthreads();
    const int nn = nn_sh;
    int4 je = make_int4(0, -1, -1, 0); int jstage = -1; bool mine = false;
    TileJointPre JP;
    if (have) {
        const int j0 = Soff[0], nje = Soff[njs] - j0;
        if (t < nje) {
            mine = true;
            je = w.tl_cons[(size_t)tile * RP_TILE_CCAP + j0 + t];
            jstage = 0; while (Soff[jstage + 1] - j0 <= t) ++jstage;
        }
    }
    JP.locked = 0x7; JP.limited = 0; JP.motor = 0; JP.b1 = (mine && je.y >= 0) ? Lg[je.y] : -1; JP.b2 = (mine && je.z >= 0) ? Lg[je.z] : -1;
    const int j = -1 - je.x;
    float4 *vs = w.s_lin, *as = w.s_ang, *vt = w.t_lin, *at = w.t_ang, *rs = w.s_rot, *ts = w.s_trans, *rt = w.t_rot, *tt = w.t_trans;
    const bool ws = w.prm.p.warmstart_joints != 0;
    const float ws_coeff = w.prm.p.warmstart_coefficient;
    JN_STAMP(0);
    for (int s = 0; s < substeps; ++s) {
        // S2: every cone body is incremented on its way into LDS (halo bodies redundantly, on the same operands)
        for (int l = t; l < nb; l += nt) {
            const int g = Lg[l];
            V3 lin = v3(vs[g]), ang = v3(as[g]);
            body_increment(w, w.b_flags[g], lin, ang, q4(rs[g]), v3(w.s_incl[g]), v3(w.s_inca[g]), v3(w.b_invpi[g]), q4(w.b_pframe[g]));
            Ll[l] = f4(lin, 0.0f); La[l] = f4(ang, 0.0f);
        }
        JN_STAMP(1);
        // the rows of this substep from the poses (JointConstraintBuilder::update); the impulse of the last sweep seeds the next
        if (mine) {
            const float i0 = JP.R.c[0].impulse, i1 = JP.R.c[1].impulse, i2 = JP.R.c[2].impulse;
            const JnPoseIO io = {rs, ts, JP.b1, JP.b2};
            const JnSink sink = {JP};
            joint_update_one_t<JnPoseIO, JnSink, true>(w, io, j, 0, sink); // (substep 0: seeded from the joint's stored impulses)
            if (s > 0 && ws) { JP.R.c[0].impulse = i0 * ws_coeff; JP.R.c[1].impulse = i1 * ws_coeff; JP.R.c[2].impulse = i2 * ws_coeff; }
        }
        __syncthreads();
        JN_STAMP(2);
        const JnVelIO vio = {Ll, La, je.y, je.z};
        for (int st = 0; st < njs; ++st) {
            if (jstage == st) joint_solve_fetched<JnVelIO, 3>(w, vio, j, JP.b1, JP.b2, 3, JP.im1, JP.im2, JP.R, false, joint_warmstart != 0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        JN_STAMP(3);
        // S6: the owned bodies are integrated on their way out (velocities AND poses to the other copies)
        for (int l = t; l < n_owned; l += nt) {
            const int g = Lg[l];
            V3 lin = v3(Ll[l]), ang = v3(La[l]), trans = v3(ts[g]); Q4 rot = q4(rs[g]);
            body_integrate(w, w.b_flags[g], lin, ang, rot, trans);
            jn_store_sc1(vt + g, f4(lin, 0.0f)); jn_store_sc1(at + g, f4(ang, 0.0f)); jn_store_sc1(rt + g, f4(rot)); jn_store_sc1(tt + g, f4(trans, 0.0f));
        }
        { float4 *a = vs; vs = vt; vt = a; a = as; as = at; at = a; a = rs; rs = rt; rt = a; a = ts; ts = tt; tt = a; }
        JN_STAMP(4);
        if (jn_sync(w, tile, e0 + 2u * (unsigned)s + 1u, Lnbr, nn)) return;
        JN_STAMP(5);
        for (int l = t; l < nb; l += nt) { const int g = Lg[l]; Ll[l] = vs[g]; La[l] = as[g]; }
        __syncthreads();
        JN_STAMP(6);
        for (int st = 0; st < njs; ++st) {
            if (jstage == st) joint_solve_fetched<JnVelIO, 3>(w, vio, j, JP.b1, JP.b2, 3, JP.im1, JP.im2, JP.R, true, false);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        JN_STAMP(7);
        for (int l = t; l < n_owned; l += nt) { const int g = Lg[l]; jn_store_sc1(vt + g, Ll[l]); jn_store_sc1(at + g, La[l]); }
        { float4 *a = vs; vs = vt; vt = a; a = as; as = at; at = a; }
        JN_STAMP(8);
        if (s + 1 < substeps) if (jn_sync(w, tile, e0 + 2u * (unsigned)s + 2u, Lnbr, nn)) return;
        JN_STAMP(9);
    }
    // what the write-back reads: the owner instance's impulses (and right-hand sides, as the sweeps leave them) in the current copy of jm
    if (mine && je.w != 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) jm_put(w, j, k, w.c_par, JP.R.c[k].impulse, JP.R.c[k].rhs);
    }
    JN_STAMP(10);
#ifdef RP_TILE_PROFILE
    if (blockIdx.x == 0 && t == 0) w.dbg[260 + 11] += 1;
#endif
#undef JN_STAMP
}
int rp_test_jn_stall_tile = -1; // (RP_TEST_JN_STALL=<tile>, testing build only: rp_api.hip)
void rp_launch_joint_net_step(const DevWorld &w, hipStream_t st, int grid, int joint_warmstart) {
    hipLaunchKernelGGL(k_joint_net_step, dim3(grid < 1 ? 1 : grid), dim3(RP_JN_THREADS), 0, st, w, joint_warmstart, rp_test_jn_stall_tile);
}
// most workgroups a k_joint_net_step launch may use on the current device (all of them resident at once: grid barriers), 0 = none
int rp_joint_net_cap(void) {
    static int cached[64] = {0};
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return 0;
    if (device >= 0 && device < 64 && cached[device]) return cached[device] > 0 ? cached[device] : 0;
    hipDeviceProp_t prop;
    int per_cu = 0, cus = 0;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_joint_net_step, RP_JN_THREADS, 0) != hipSuccess) per_cu = 0;
    if (per_cu > 1) per_cu = 1;
    int g = cus * per_cu - (cus + 15) / 16; // (a sixteenth left free, like the fused island step)
    if (g < 1) g = -1;
    if (device >= 0 && device < 64) cached[device] = g;
    return g > 0 ? g : 0;
}

// ---- b3d_large_pyramid: the TGS loop of a tiled contact world as ONE launch (round 6) ------------------------------------
// Measured on the sweep launches (rocprofv3, 1,060 steps of b3d_large_pyramid): a substep is k_ws_prepare 12.9 us + k_increment_ws 12.2 us
// + biased sweep 27.6 us + relaxed sweep 38.2 us, while a tile is busy for 19.3 / 28.8 us of a sweep and the two preparation launches are
// one or two round trips each: a third of the solver loop is launch, prologue and the wait for the slowest of 240 tiles — sixteen times a
// step.  Here a workgroup keeps its tile for the whole step (the joint-net recipe, k_joint_net_step, without its register residency: a
// tile's ~480 manifold instances x 816 B are streamed every sweep as before) and the four launches of a substep become four phases:
//   A  update + warm-start terms of the manifolds the tile OWNS (ws_prepare_one, k_ws_prepare's body)            -> flag
//   B  increment + body-centric warm start of the bodies it owns (body_increment_ws, k_increment_ws's body)      -> flag
//   C  biased sweep over the cone (tile_apply2), owned bodies integrated on their way out                        -> flag
//   D  relaxed sweep over the cone
// What another tile reads next — terms, owned bodies' velocities and poses, the owner's copy of the six mutable planes — is stored
// write-through (sc1), a tile waits for the flags of the tiles it exchanges bodies with (jn_sync: a constraint instance in another cone
// puts one of my bodies there, a toucher of my body owned elsewhere puts my body into its owner's cone: both are neighbours in tl_nbr),
// and the double buffers of the sweeps keep a tile that is one phase ahead from overwriting what a neighbour still reads: D reads the
// copies C wrote and writes the ones A / B / C read, and between my next C and a neighbour's D lie two flags it has to have passed.
// Same device functions on the same operands in the same order as the launches: the same bits.  Where it runs: lean graphs
// (DevWorld::lean bit 3, the grid in bits 8 and up; lean_dead verifies a valid tiling that fits the grid), no impulse joints, one PGS
// and one stabilisation iteration, at most five substeps (a launch owns 16 flag values).  A tile that waits in vain raises
// FL_JN_TIMEOUT: nothing is committed by this kernel, the step dies like any lean step and the world keeps the sweep launches.
#ifndef RP_TS_SC1LD
#define RP_TS_SC1LD true // what another tile stored in this launch is read past the L2 (sc1 loads) and a flag is not followed by an agent-scope acquire: the L2 the XCD's tiles share keeps their rows
#endif
struct TsPrepAcc { // ws_prepare_one's view of one owned manifold: rows (of the current copy) in registers, write-through stores, poses fetched with the rows
    static constexpr bool PRELOAD = false;
    const DevWorld &w; const float4 *v; int pos, par, b1, b2, nn; Xf X1, X2;
    RP_DEV float4 ld(int plane) const { return v[plane]; }
    RP_DEV void st(int plane, float4 x) const { st16_sc1(w.C + (size_t)cplane(plane, par) * w.cons_cap, (unsigned)pos, x); }
    RP_DEV int id1() const { return b1; }
    RP_DEV int id2() const { return b2; }
    RP_DEV int n() const { return nn; }
    RP_DEV Xf xf(int id) const {
        if (id < 0) { Xf x; x.r = q4(0, 0, 0, 1); x.t = v3(0, 0, 0); return x; }
        return id == b1 ? X1 : X2;
    }
};
// phase A for one owned manifold: everything ws_prepare_one reads comes in with ONE round trip (k_ws_prepare hides its dependent loads
// behind thousands of wavefronts; a tile has four)
RP_DEV void ts_prepare(const DevWorld &w, const int pos, const int b1, const int b2, const int n, const int par, const float4 *rot, const float4 *trans, const float solved_dt) {
    float4 v[CP_COUNT];
#define TP_LD(p) v[p] = (RP_TS_SC1LD && cplane_mut_index(p) >= 0) ? ld16_sc1(w.C + (size_t)cplane(p, par) * w.cons_cap, (unsigned)pos) : w.C[(size_t)cplane(p, par) * w.cons_cap + pos]
    TP_LD(CP_H0); TP_LD(CP_H1); TP_LD(CP_H2); TP_LD(CP_H3); TP_LD(CP_H4); TP_LD(CP_H5); TP_LD(CP_H6); TP_LD(CP_H7); TP_LD(CP_HM0); TP_LD(CP_HM1);
    TP_LD(CP_T4); TP_LD(CP_T5); TP_LD(CP_T6); TP_LD(CP_T7); TP_LD(CP_B0); TP_LD(CP_B1); TP_LD(CP_B2);
#pragma unroll
    for (int k = 0; k < 4; ++k) { TP_LD(NPL(k, NP_M)); TP_LD(NPL(k, NP_C)); TP_LD(NPL(k, NP_D)); TP_LD(NPL(k, NP_E)); TP_LD(NPL(k, NP_F)); }
#undef TP_LD
    const int g1 = b1 >= 0 ? b1 : 0, g2 = b2 >= 0 ? b2 : 0;
    float4 r1 = ld16_t<RP_TS_SC1LD>(rot, (unsigned)g1), t1 = ld16_t<RP_TS_SC1LD>(trans, (unsigned)g1), r2 = ld16_t<RP_TS_SC1LD>(rot, (unsigned)g2), t2 = ld16_t<RP_TS_SC1LD>(trans, (unsigned)g2);
    __builtin_amdgcn_sched_group_barrier(0x020, 48, 0); // every VMEM read above as one group
#define TP_PIN(r_) asm volatile("" : "+v"((r_).x), "+v"((r_).y), "+v"((r_).z), "+v"((r_).w))
    TP_PIN(v[CP_H0]); TP_PIN(v[CP_H1]); TP_PIN(v[CP_H2]); TP_PIN(v[CP_H3]); TP_PIN(v[CP_H4]); TP_PIN(v[CP_H5]); TP_PIN(v[CP_H6]); TP_PIN(v[CP_H7]); TP_PIN(v[CP_HM0]); TP_PIN(v[CP_HM1]);
    TP_PIN(v[CP_T4]); TP_PIN(v[CP_T5]); TP_PIN(v[CP_T6]); TP_PIN(v[CP_T7]); TP_PIN(v[CP_B0]); TP_PIN(v[CP_B1]); TP_PIN(v[CP_B2]);
#pragma unroll
    for (int k = 0; k < 4; ++k) { TP_PIN(v[NPL(k, NP_M)]); TP_PIN(v[NPL(k, NP_C)]); TP_PIN(v[NPL(k, NP_D)]); TP_PIN(v[NPL(k, NP_E)]); TP_PIN(v[NPL(k, NP_F)]); }
    TP_PIN(r1); TP_PIN(t1); TP_PIN(r2); TP_PIN(t2);
#undef TP_PIN
    Xf X1, X2; X1.r = q4(r1); X1.t = v3(t1); X2.r = q4(r2); X2.t = v3(t2);
    const TsPrepAcc A = {w, v, pos, par, b1, b2, n, X1, X2};
    ws_prepare_one<TsPrepAcc, true>(w, A, pos, solved_dt);
}
// the contact stages of one sweep over the cone (the stage loop of k_tile_sweep: the list entry and point count of a thread's next
// stage are fetched while it works on the current one)
// (Measured and removed, round 6: the wavefronts of a stage that hold no manifold — a stage of b3d_large_pyramid keeps ~53 of 128 lane
// pairs busy — touching the NEXT stage's rows, one dword per row and manifold through a live sink register, so that the busy wavefronts
// would find them in the L2 a stage later.  Every stage got 2.2x SLOWER (biased 60 -> 132, relaxed 103 -> 224 us per launch): a stage is
// bound by the row traffic it already causes, not by the latency of its first miss — the verdict of round 3's global_load_lds experiment
// with the SAME lanes asking ahead, now confirmed with the loads in wavefronts that have nothing else to wait for.)
template <int MODE>
RP_DEV void ts_stages(const DevWorld &w, const int4 *cons, const int *Soff, const int nst, const int my, const int per, const bool odd, const int *Lg, float4 *Ll, float4 *La,
                      const bool friction, const float solved_dt, const int cpar, const float4 *rot, const float4 *trans) {
    int4 e_next; int n_next; bool have_next;
    { const int i0 = Soff[0] + my; have_next = i0 < Soff[1]; e_next = cons[have_next ? i0 : 0]; n_next = w.k_n[e_next.x > 0 ? e_next.x : 0]; }
    for (int s = 0; s < nst; ++s) {
        const int end = Soff[s + 1];
        int i = Soff[s] + my;
        int4 e = e_next; int n = n_next;
        bool have = have_next;
        { const int i1 = end + my; have_next = i1 < Soff[s + 2]; e_next = cons[have_next ? i1 : 0]; } // (Soff[nst + 1] = Soff[nst]: nothing behind the last stage)
        while (have) {
            tile_apply2<MODE, true, RP_TS_SC1LD>(w, e, n, odd, Lg, Ll, La, friction, solved_dt, cpar, rot, trans);
            i += per; have = i < end;
            if (have) { e = cons[i]; n = w.k_n[e.x]; }
        }
        n_next = w.k_n[e_next.x > 0 ? e_next.x : 0];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // (orders the LDS velocities only: the row stores stay in flight until the next flag)
    }
}
#define RP_TS_THREADS 256 // (512 threads — every owned manifold of phase A in one round — was measured: 256 registers per lane put the preloaded rows into scratch, phase A 10 -> 30 us)
__global__ void __launch_bounds__(RP_TS_THREADS) k_tile_step(DevWorld w, int friction_in_bias, int stall_tile) {
    if (lean_dead(w)) return; // (the same answer in every workgroup: nothing it reads changes while a lean graph runs)
    if ((int)blockIdx.x == stall_tile) return; // (test hook, testing build only: a workgroup that never becomes resident)
    __shared__ float4 Ll[RP_TILE_BCAP], La[RP_TILE_BCAP];
    __shared__ int Lg[RP_TILE_BCAP];
    __shared__ int Soff[RP_TILE_STAGES + 4];
    __shared__ int Lnbr[RP_TS_THREADS];
    __shared__ int Lown[RP_TILE_CCAP]; // positions of the manifolds this tile owns
    __shared__ int nn_sh, nown_sh;
    const int t = threadIdx.x, nt = blockDim.x;
    // XCD-aware placement: workgroup b runs on XCD b % 8 (observed dispatch order: a speed matter only, any map is correct), and tiles
    // that are neighbours on the curve share halo rows — every XCD takes a contiguous run of the curve, so that the second tile to ask
    // for a row finds it in the L2 they share instead of on the fabric (consecutive workgroups used to put neighbouring tiles on different XCDs)
    const int ntiles_ = w.flags[FL_N_TILES], per_xcd_ = (ntiles_ + 7) >> 3;
    const int tile = (8 * per_xcd_ <= (int)gridDim.x) ? ((int)blockIdx.x & 7) * per_xcd_ + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    if (8 * per_xcd_ <= (int)gridDim.x && ((int)blockIdx.x >> 3) >= per_xcd_) return; // (workgroups beyond the map)
#ifdef RP_TILE_PROFILE // thread 0 of a tile in the middle of the curve accumulates wall-clock ticks (10 ns) per phase into dbg[272 ..] (tools/tile_diag.py)
#define TS_STAMP(k) do { if (blockIdx.x == 100 && t == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const long long n_ = (long long)wall_clock64(); w.dbg[272 + (k)] += n_ - tsp_; tsp_ = n_; } } while (0)
    long long tsp_ = (long long)wall_clock64();
#else
#define TS_STAMP(k) do { } while (0)
#endif
    if (tile >= w.flags[FL_N_TILES]) return; // (workgroups beyond the tiling: nobody waits for them)
    const int nst = w.flags[FL_N_STAGES], substeps = w.prm.num_substeps;
    const bool fib = friction_in_bias != 0;
    if (t == 0) { nn_sh = 0; nown_sh = 0; }
    const unsigned e0 = 16u * (unsigned)w.flags[FL_SEQ]; // (FL_SEQ moves once per step graph, behind this launch: the same in every workgroup)
    __syncthreads();
    for (int a = t; a < w.tile_cap; a += nt)
        if ((w.tl_nbr[(size_t)tile * RP_TILE_NBR_WORDS(w.tile_cap) + (a >> 5)] >> (a & 31)) & 1u) { const int k = atomicAdd(&nn_sh, 1); if (k < RP_TS_THREADS) Lnbr[k] = a; }
    const int4 *cons = w.tl_cons + (size_t)tile * RP_TILE_CCAP;
    const int4 hdr = w.tl_hdr[tile];
    const int nb = hdr.x, nc = hdr.y, n_owned = hdr.z;
    for (int i = t; i < nc; i += nt) { const int4 e = cons[i]; if (e.w != 0 && e.x >= 0) Lown[atomicAdd(&nown_sh, 1)] = e.x; }
    for (int s = t; s <= nst + 3; s += nt) Soff[s] = w.tl_soff[(size_t)tile * (RP_TILE_STAGES + 1) + (s < nst ? s : nst)];
    for (int l = t; l < nb; l += nt) Lg[l] = w.tl_bodies[(size_t)tile * RP_TILE_BCAP + l];
    __syncthreads();
    const int nn = nn_sh < RP_TS_THREADS ? nn_sh : RP_TS_THREADS;
    const int n_own = nown_sh;
    // a thread's first owned manifold stays in its registers for the whole step (position, solver bodies, point count: what phase A asks for first)
    const int own_pos = t < n_own ? Lown[t] : -1;
    const int own_b1 = own_pos >= 0 ? w.k_b1[own_pos] : -1, own_b2 = own_pos >= 0 ? w.k_b2[own_pos] : -1, own_n = own_pos >= 0 ? w.k_n[own_pos] : 0;
    // (... and its second one: a tile of b3d_large_pyramid owns ~250 manifolds, some tiles a few more than the workgroup has threads)
    const int own2_pos = t + nt < n_own ? Lown[t + nt] : -1;
    const int own2_b1 = own2_pos >= 0 ? w.k_b1[own2_pos] : -1, own2_b2 = own2_pos >= 0 ? w.k_b2[own2_pos] : -1, own2_n = own2_pos >= 0 ? w.k_n[own2_pos] : 0;
    float4 *vs = w.s_lin, *as = w.s_ang, *vt = w.t_lin, *at = w.t_ang, *rs = w.s_rot, *ts = w.s_trans, *rt = w.t_rot, *tt = w.t_trans;
    const int par = w.c_par;
    const int my = t >> 1, per = nt >> 1;
    const bool odd = (t & 1) != 0;
    const int half = nt >> 1; // phase B: the first half of the workgroup adds the linear chains, the second the angular ones (as k_increment_ws's two grid halves)
    TS_STAMP(0);
    for (int s = 0; s < substeps; ++s) {
        const float solved_dt = (float)s * w.prm.dt_sub;
        // A: the manifolds this tile owns
        if (own_pos >= 0) ts_prepare(w, own_pos, own_b1, own_b2, own_n, par, rs, ts, solved_dt);
        if (own2_pos >= 0) ts_prepare(w, own2_pos, own2_b1, own2_b2, own2_n, par, rs, ts, solved_dt);
        for (int k = t + 2 * nt; k < n_own; k += nt) { const int pos = Lown[k]; ts_prepare(w, pos, w.k_b1[pos], w.k_b2[pos], w.k_n[pos], par, rs, ts, solved_dt); }
        TS_STAMP(1);
        if (jn_sync<!RP_TS_SC1LD>(w, tile, e0 + 3u * (unsigned)s + 1u, Lnbr, nn)) return;
        TS_STAMP(2);
        // B: the bodies this tile owns
        if (t < half) {
            for (int l = t; l < n_owned; l += half) { const int g = Lg[l]; V3 lin, ang; body_increment_ws_at<RP_TS_SC1LD>(w, g, vs, as, rs, lin, ang); Ll[l] = f4(lin, 0.0f); st16_sc1(vs, (unsigned)g, f4(lin, 0.0f)); }
        } else {
            for (int l = t - half; l < n_owned; l += half) { const int g = Lg[l]; V3 lin, ang; body_increment_ws_at<RP_TS_SC1LD>(w, g, vs, as, rs, lin, ang); La[l] = f4(ang, 0.0f); st16_sc1(as, (unsigned)g, f4(ang, 0.0f)); }
        }
        TS_STAMP(3);
        if (jn_sync<!RP_TS_SC1LD>(w, tile, e0 + 3u * (unsigned)s + 2u, Lnbr, nn)) return;
        TS_STAMP(4);
        // C: biased sweep; the halo bodies come in, the owned ones leave integrated (velocities AND poses to the other copies)
        for (int l = n_owned + t; l < nb; l += nt) { const int g = Lg[l]; Ll[l] = ld16_t<RP_TS_SC1LD>(vs, (unsigned)g); La[l] = ld16_t<RP_TS_SC1LD>(as, (unsigned)g); }
        __syncthreads();
        TS_STAMP(5);
        ts_stages<MODE_BIAS>(w, cons, Soff, nst, my, per, odd, Lg, Ll, La, fib, solved_dt, par, rs, ts);
        TS_STAMP(6);
        for (int l = t; l < n_owned; l += nt) {
            const int g = Lg[l];
            V3 lin = v3(Ll[l]), ang = v3(La[l]), trans = v3(ld16_t<RP_TS_SC1LD>(ts, (unsigned)g)); Q4 rot = q4(ld16_t<RP_TS_SC1LD>(rs, (unsigned)g));
            body_integrate(w, w.b_flags[g], lin, ang, rot, trans);
            Ll[l] = f4(lin, 0.0f); La[l] = f4(ang, 0.0f);
            st16_sc1(vt, (unsigned)g, f4(lin, 0.0f)); st16_sc1(at, (unsigned)g, f4(ang, 0.0f)); st16_sc1(rt, (unsigned)g, f4(rot)); st16_sc1(tt, (unsigned)g, f4(trans, 0.0f));
        }
        { float4 *a = vs; vs = vt; vt = a; a = as; as = at; at = a; a = rs; rs = rt; rt = a; a = ts; ts = tt; tt = a; }
        TS_STAMP(7);
        if (jn_sync<!RP_TS_SC1LD>(w, tile, e0 + 3u * (unsigned)s + 3u, Lnbr, nn)) return;
        TS_STAMP(8);
        // D: relaxed sweep (the other copy of the mutable planes, the new poses)
        for (int l = n_owned + t; l < nb; l += nt) { const int g = Lg[l]; Ll[l] = ld16_t<RP_TS_SC1LD>(vs, (unsigned)g); La[l] = ld16_t<RP_TS_SC1LD>(as, (unsigned)g); }
        __syncthreads();
        TS_STAMP(9);
        ts_stages<MODE_RELAX>(w, cons, Soff, nst, my, per, odd, Lg, Ll, La, true, solved_dt + w.prm.dt_sub, par ^ 1, rs, ts);
        TS_STAMP(10);
        for (int l = t; l < n_owned; l += nt) { const int g = Lg[l]; st16_sc1(vt, (unsigned)g, Ll[l]); st16_sc1(at, (unsigned)g, La[l]); }
        { float4 *a = vs; vs = vt; vt = a; a = as; as = at; at = a; }
        __syncthreads(); // (every store of this phase has completed: phase A reads the rows this tile just wrote, phase B its own bodies)
        TS_STAMP(11);
    }
#ifdef RP_TILE_PROFILE
    if (blockIdx.x == 100 && t == 0) w.dbg[272 + 12] += 1;
#endif
#undef TS_STAMP
}
int rp_test_ts_stall_tile = -1; // (RP_TEST_TS_STALL=<tile>, testing build only: rp_api.hip)
// returns the parity for rp_launch_solver_writeback (velocities and mutable planes where they began, poses in the other copy after an odd number of substeps)
void rp_launch_tile_step(const DevWorld &w, hipStream_t st, int grid, int friction_in_bias) {
    hipLaunchKernelGGL(k_tile_step, dim3(grid < 1 ? 1 : grid), dim3(RP_TS_THREADS), 0, st, w, friction_in_bias, rp_test_ts_stall_tile);
}
// most workgroups a k_tile_step launch may use on the current device (all of them resident at once), 0 = none
int rp_tile_step_cap(void) {
    static int cached[64] = {0};
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return 0;
    if (device >= 0 && device < 64 && cached[device]) return cached[device] > 0 ? cached[device] : 0;
    hipDeviceProp_t prop;
    int per_cu = 0, cus = 0;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tile_step, RP_TS_THREADS, 0) != hipSuccess) per_cu = 0;
    if (per_cu > 1) per_cu = 1;
    int g = cus * per_cu - (cus + 15) / 16; // (a sixteenth left free, like the fused island step)
    if (g < 1) g = -1;
    if (device >= 0 && device < 64) cached[device] = g;
    return g > 0 ? g : 0;
}

void rp_launch_tiles_build(const DevWorld &w, hipStream_t st) {
    if (!w.tile_cap) return;
    int n = w.cons_cap > w.n_bodies ? w.cons_cap : w.n_bodies;
    int blocks = (n + 1023) / 1024; if (blocks > w.gbar_blocks) blocks = w.gbar_blocks; if (blocks < 1) blocks = 1; // all resident (grid barriers)
    hipLaunchKernelGGL(k_tiles_sort, dim3(blocks), dim3(1024), 0, st, w);
    hipLaunchKernelGGL(k_tiles_cones, dim3(w.tile_target < w.tile_cap ? w.tile_target : w.tile_cap), dim3(RP_CONE_THREADS), 0, st, w); // (no grid barrier: any grid will do, workgroups loop over tiles)
}
// one sweep over every tile: reads w.s_lin / w.s_ang, leaves the result in w.t_lin / w.t_ang (the caller swaps the pointers)
void rp_launch_tile_sweep(const DevWorld &w, hipStream_t st, int mode, int grid, int friction_in_bias, float solved_dt, int fuse, int joint_warmstart) {
    if (grid < 1) grid = 1;
    static const bool pairs = !(getenv("RP_TILE_LANES") && atoi(getenv("RP_TILE_LANES")) == 1);
    if (pairs) {
        if (mode == MODE_BIAS) hipLaunchKernelGGL((k_tile_sweep<MODE_BIAS, true>), dim3(grid), dim3(RP_TILE_THREADS), 0, st, w, friction_in_bias, solved_dt, fuse, joint_warmstart);
        else hipLaunchKernelGGL((k_tile_sweep<MODE_RELAX, true>), dim3(grid), dim3(RP_TILE_THREADS), 0, st, w, friction_in_bias, solved_dt, fuse, joint_warmstart);
    } else {
        if (mode == MODE_BIAS) hipLaunchKernelGGL((k_tile_sweep<MODE_BIAS, false>), dim3(grid), dim3(RP_TILE_THREADS), 0, st, w, friction_in_bias, solved_dt, fuse, joint_warmstart);
        else hipLaunchKernelGGL((k_tile_sweep<MODE_RELAX, false>), dim3(grid), dim3(RP_TILE_THREADS), 0, st, w, friction_in_bias, solved_dt, fuse, joint_warmstart);
    }
    // (the restitution sweep, rare, stays on the per-stage launches: rp_solver.hip)
}
// workgroups of k_tiles_sort (1024 threads) one CU holds at once (0 = the query failed): input of DevWorld::gbar_blocks (rp_api.hip)
int rp_occ_tiles_build(void) { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_tiles_sort, 1024, 0) != hipSuccess) n = 0; return n; }
