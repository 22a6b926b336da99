// rp_islands.hip — contact islands and the LDS-resident per-island TGS megakernel.
//
// The reference solves all awake bodies as ONE colour-ordered Gauss-Seidel system
// (/root/reference/src/dynamics/island_manager/manager.rs:20-39, staged_island_solver/worker.rs) and
// pays a barrier per colour per sweep.  Constraints of different connected components never share a
// body, so sweeping each component through the same colour sequence independently is bit-identical
// to the global sweep (SURVEY Appendix B.3 applied across components).  On MI355X that turns ~100
// dependent kernel launches per step into ONE launch: each workgroup owns one island, stages its
// bodies (13 floats each) and all of its constraint planes (800 B per manifold) in the CU's 160 KiB
// LDS, and runs generate -> 4 x (increment, update+warmstart, biased solve, integrate, relaxed
// solve) -> restitution -> write-back with workgroup barriers between colours.  HBM is touched once
// per step per manifold (pair data in, impulses out) instead of 12 sweeps.
// Islands that do not fit (more than RP_ISL_NB_MAX bodies or RP_ISL_NC_MAX manifolds) and bodies
// without contacts stay on the global per-colour path (rp_solver.hip).
//
// Island discovery (only when the active-manifold set changed): union-find hooking with atomicMin
// over the active dynamic-dynamic pairs inside one workgroup, then island numbering and list filling.
// persistent.rs keeps comparable connected components for sleeping; here they drive scheduling only.
#include "rp_constraint.h"

RP_DEV int ld_i32(int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RP_DEV int uf_find(int *label, int x) {
    int p = ld_i32(&label[x]);
    while (p != x) { x = p; p = ld_i32(&label[x]); }
    return x;
}
RP_DEV bool is_dyn(const DevWorld &w, int b) { return b >= 0 && (w.b_flags[b] & RP_BF_TYPE_MASK) == RP_BODY_DYNAMIC; }

__global__ void __launch_bounds__(1024) k_islands_build(DevWorld w) {
    if (!w.flags[FL_LAYOUT_DIRTY]) return;
    __shared__ int changed;
    int tid = threadIdx.x, nt = blockDim.x;
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    int nb = w.n_bodies;
    for (int b = tid; b < nb; b += nt) { w.b_label[b] = b; w.r_nb[b] = 0; w.r_nc[b] = 0; w.r_island[b] = -1; w.b_island[b] = -1; w.b_local[b] = -1; }
    if (tid == 0) { w.flags[FL_N_ISLANDS] = 0; w.flags[FL_N_GLOB_BODIES] = 0; w.flags[FL_ISL_BODY_CURSOR] = 0; w.flags[FL_ISL_CONS_CURSOR] = 0; }
    __threadfence(); __syncthreads();
    // (b) connected components over active pairs whose two sides are dynamic
    for (int iter = 0; iter < 4096; ++iter) {
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int s = tid; s < top; s += nt) {
            if (w.p_c1[s] < 0 || w.p_nsc[s] == 0) continue;
            int b1 = w.c_parent[w.p_c1[s]], b2 = w.c_parent[w.p_c2[s]];
            if (!is_dyn(w, b1) || !is_dyn(w, b2)) continue;
            int r1 = uf_find(w.b_label, b1), r2 = uf_find(w.b_label, b2);
            if (r1 != r2) { int hi = r1 > r2 ? r1 : r2, lo = r1 > r2 ? r2 : r1; atomicMin(&w.b_label[hi], lo); changed = 1; }
        }
        __threadfence(); __syncthreads();
        for (int b = tid; b < nb; b += nt) { int r = uf_find(w.b_label, b); __hip_atomic_store(&w.b_label[b], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        __threadfence(); __syncthreads();
        int c = changed;
        __syncthreads();
        if (!c) break;
    }
    // (c) per-root sizes
    for (int b = tid; b < nb; b += nt) if (is_dyn(w, b)) atomicAdd(&w.r_nb[ld_i32(&w.b_label[b])], 1);
    for (int s = tid; s < top; s += nt) {
        if (w.p_c1[s] < 0 || w.p_nsc[s] == 0) continue;
        int b1 = w.c_parent[w.p_c1[s]], b2 = w.c_parent[w.p_c2[s]];
        int b = is_dyn(w, b1) ? b1 : b2;
        if (is_dyn(w, b)) atomicAdd(&w.r_nc[ld_i32(&w.b_label[b])], 1);
    }
    __threadfence(); __syncthreads();
    // (d) number the islands that fit in LDS
    for (int b = tid; b < nb; b += nt) {
        if (!is_dyn(w, b) || ld_i32(&w.b_label[b]) != b) continue;
        int cnb = ld_i32(&w.r_nb[b]), cnc = ld_i32(&w.r_nc[b]);
        if (cnc > 0 && cnb <= RP_ISL_NB_MAX && cnc <= RP_ISL_NC_MAX) {
            int id = atomicAdd(&w.flags[FL_N_ISLANDS], 1);
            w.isl_body_begin[id] = atomicAdd(&w.flags[FL_ISL_BODY_CURSOR], cnb);
            w.isl_cons_begin[id] = atomicAdd(&w.flags[FL_ISL_CONS_CURSOR], cnc);
            w.isl_nb[id] = cnb; w.isl_nc[id] = cnc; w.isl_fill_b[id] = 0; w.isl_fill_c[id] = 0;
            w.r_island[b] = id;
        } else {
            atomicAdd(&w.flags[FL_N_GLOB_BODIES], cnb);
        }
    }
    __threadfence(); __syncthreads();
    // (e) fill the island lists
    for (int b = tid; b < nb; b += nt) {
        if (!is_dyn(w, b)) continue;
        int id = ld_i32(&w.r_island[ld_i32(&w.b_label[b])]);
        if (id >= 0) {
            int k = atomicAdd(&w.isl_fill_b[id], 1);
            w.isl_bodies[w.isl_body_begin[id] + k] = b;
            w.b_island[b] = id; w.b_local[b] = k;
        }
    }
    for (int s = tid; s < top; s += nt) {
        if (w.p_c1[s] < 0 || w.p_nsc[s] == 0) { w.p_island[s] = -1; continue; }
        int b1 = w.c_parent[w.p_c1[s]], b2 = w.c_parent[w.p_c2[s]];
        int b = is_dyn(w, b1) ? b1 : b2;
        int id = is_dyn(w, b) ? ld_i32(&w.r_island[ld_i32(&w.b_label[b])]) : -1;
        w.p_island[s] = id;
        if (id >= 0) { int k = atomicAdd(&w.isl_fill_c[id], 1); w.isl_cons[w.isl_cons_begin[id] + k] = s; }
        else { int color = w.p_color[s]; if (color <= RP_COLOR_OVERFLOW) atomicAdd(&w.color_count_glob[color], 1); }
    }
}

// ---- accessor over LDS ------------------------------------------------------------------------
struct LdsAcc {
    float4 *C; int t;
    int *kb1, *kb2, *kn, *kcid;
    float4 *lin, *ang, *rot, *trans;
    RP_DEV float4 ld(int plane) const { return C[plane * RP_ISL_NC_MAX + t]; }
    RP_DEV void st(int plane, float4 v) const { C[plane * RP_ISL_NC_MAX + t] = v; }
    RP_DEV int id1() const { return kb1[t]; }
    RP_DEV int id2() const { return kb2[t]; }
    RP_DEV int n() const { return kn[t]; }
    RP_DEV int cids() const { return kcid[t]; }
    RP_DEV void set_meta(int a, int b, int cnt, int cid) const { kb1[t] = a; kb2[t] = b; kn[t] = cnt; kcid[t] = cid; }
    RP_DEV Vel vel(int id) const {
        Vel v;
        if (id < 0) { v.lin = v3(0, 0, 0); v.ang = v3(0, 0, 0); } else { v.lin = v3(lin[id]); v.ang = v3(ang[id]); }
        return v;
    }
    RP_DEV void set_vel(int id, const Vel &v) const { if (id >= 0) { lin[id] = f4(v.lin, 0.0f); ang[id] = f4(v.ang, 0.0f); } }
    RP_DEV Xf xf(int id) const {
        Xf x;
        if (id < 0) { x.r = q4(0, 0, 0, 1); x.t = v3(0, 0, 0); } else { x.r = q4(rot[id]); x.t = v3(trans[id]); }
        return x;
    }
};

#define ISL_THREADS 192

// One workgroup = one island, everything in LDS.
__global__ void __launch_bounds__(ISL_THREADS) k_island_solve(DevWorld w, int has_restitution) {
    __shared__ float4 C[CP_COUNT * RP_ISL_NC_MAX];
    __shared__ float4 B_lin[RP_ISL_NB_MAX], B_ang[RP_ISL_NB_MAX], B_rot[RP_ISL_NB_MAX], B_trans[RP_ISL_NB_MAX];
    __shared__ float4 B_incl[RP_ISL_NB_MAX], B_inca[RP_ISL_NB_MAX], B_invpi[RP_ISL_NB_MAX], B_pframe[RP_ISL_NB_MAX];
    __shared__ int B_gid[RP_ISL_NB_MAX], B_fl[RP_ISL_NB_MAX];
    __shared__ int K_b1[RP_ISL_NC_MAX], K_b2[RP_ISL_NC_MAX], K_n[RP_ISL_NC_MAX], K_cid[RP_ISL_NC_MAX], K_slot[RP_ISL_NC_MAX], K_rank[RP_ISL_NC_MAX];
    __shared__ int T_slot[RP_ISL_NC_MAX], T_rank[RP_ISL_NC_MAX];
    __shared__ int st_lo[RP_NUM_COLORS + 2], st_hi[RP_NUM_COLORS + 2], n_local_stages, ov_lo, ov_hi, any_bouncy;

    const int t = threadIdx.x;
    const int n_islands = w.flags[FL_N_ISLANDS];
    const int nst_global = w.flags[FL_N_STAGES];
    const rp_integration_params &prm = w.prm.p;
    const bool fib = prm.friction_in_bias_pass || prm.num_internal_stabilization_iterations == 0;

    for (int isl = blockIdx.x; isl < n_islands; isl += gridDim.x) {
        const int nb = w.isl_nb[isl], nc = w.isl_nc[isl];
        const int bb = w.isl_body_begin[isl], cb = w.isl_cons_begin[isl];
        __syncthreads(); // previous island of this workgroup fully written back
        // ---- bodies -> LDS (S0) ----
        if (t < nb) {
            int g = w.isl_bodies[bb + t];
            V3 lin, ang, trans, incl, inca; Q4 rot;
            body_begin(w, g, lin, ang, rot, trans, incl, inca);
            B_gid[t] = g; B_fl[t] = w.b_flags[g];
            B_lin[t] = f4(lin, 0.0f); B_ang[t] = f4(ang, 0.0f); B_rot[t] = f4(rot); B_trans[t] = f4(trans, 0.0f);
            B_incl[t] = f4(incl, 0.0f); B_inca[t] = f4(inca, 0.0f); B_invpi[t] = w.b_invpi[g]; B_pframe[t] = w.b_pframe[g];
        }
        // ---- constraint list, ordered by sweep stage (rank of the pair's colour) ----
        if (t < nc) {
            int s = w.isl_cons[cb + t];
            int color = w.p_color[s];
            T_slot[t] = s;
            T_rank[t] = color >= RP_COLOR_OVERFLOW ? nst_global : w.color_rank[color];
        }
        if (t == 0) any_bouncy = 0;
        __syncthreads();
        if (t < nc) {
            int r = T_rank[t], posn = 0;
            for (int j = 0; j < nc; ++j) { int rj = T_rank[j]; posn += (rj < r) || (rj == r && j < t); }
            K_slot[posn] = T_slot[t]; K_rank[posn] = r;
        }
        __syncthreads();
        if (t == 0) { // compact list of the stages present in this island
            int ns = 0; ov_lo = nc; ov_hi = nc;
            int i = 0;
            while (i < nc) {
                int r = K_rank[i], j = i;
                while (j < nc && K_rank[j] == r) ++j;
                if (r >= nst_global) { ov_lo = i; ov_hi = j; } else { st_lo[ns] = i; st_hi[ns] = j; ns++; }
                i = j;
            }
            n_local_stages = ns;
        }
        LdsAcc A; A.C = C; A.t = t; A.kb1 = K_b1; A.kb2 = K_b2; A.kn = K_n; A.kcid = K_cid;
        A.lin = B_lin; A.ang = B_ang; A.rot = B_rot; A.trans = B_trans;
        // ---- generate (S1) ----
        if (t < nc) {
            int s = K_slot[t];
            int rb1 = w.c_parent[w.p_c1[s]], rb2 = w.c_parent[w.p_c2[s]];
            int rel_dom = w.p_reldom[s];
            int g1 = (is_dyn(w, rb1) && rel_dom <= 0) ? rb1 : -1;
            int g2 = (is_dyn(w, rb2) && rel_dom >= 0) ? rb2 : -1;
            int l1 = g1 >= 0 ? w.b_local[g1] : -1, l2 = g2 >= 0 ? w.b_local[g2] : -1;
            if (cons_generate(w, A, s, g1, g2, l1, l2)) any_bouncy = 1;
        }
        __syncthreads();
        const int nls = n_local_stages, olo = ov_lo, ohi = ov_hi;

#define ISL_SWEEP(MODE, SDT)                                                                             \
        for (int q = 0; q < nls; ++q) {                                                                  \
            int lo = st_lo[q], hi = st_hi[q];                                                            \
            if (t >= lo && t < hi) cons_apply(w, A, MODE, fib, SDT);                                     \
            __syncthreads();                                                                             \
        }                                                                                                \
        if (ohi > olo) {                                                                                 \
            if (t == 0) { LdsAcc O = A; for (int i = olo; i < ohi; ++i) { O.t = i; cons_apply(w, O, MODE, fib, SDT); } } \
            __syncthreads();                                                                             \
        }

        for (int sub = 0; sub < w.prm.num_substeps; ++sub) {
            float solved_dt = (float)sub * w.prm.dt_sub;
            if (t < nb) { // S2
                V3 lin = v3(B_lin[t]), ang = v3(B_ang[t]);
                body_increment(w, B_fl[t], lin, ang, q4(B_rot[t]), v3(B_incl[t]), v3(B_inca[t]), v3(B_invpi[t]), q4(B_pframe[t]));
                B_lin[t] = f4(lin, 0.0f); B_ang[t] = f4(ang, 0.0f);
            }
            __syncthreads();
            ISL_SWEEP(MODE_WARMSTART, solved_dt)
            for (int it = 0; it < prm.num_internal_pgs_iterations; ++it) { ISL_SWEEP(MODE_BIAS, solved_dt) }
            if (t < nb) { // S6
                V3 lin = v3(B_lin[t]), ang = v3(B_ang[t]), trans = v3(B_trans[t]); Q4 rot = q4(B_rot[t]);
                body_integrate(w, B_fl[t], lin, ang, rot, trans);
                B_lin[t] = f4(lin, 0.0f); B_ang[t] = f4(ang, 0.0f); B_rot[t] = f4(rot); B_trans[t] = f4(trans, 0.0f);
            }
            __syncthreads();
            for (int it = 0; it < prm.num_internal_stabilization_iterations; ++it) { ISL_SWEEP(MODE_RELAX, solved_dt + w.prm.dt_sub) }
        }
        if (has_restitution && any_bouncy) { ISL_SWEEP(MODE_RESTITUTION, 0.0f) }
#undef ISL_SWEEP
        // ---- write-back (S9, S10, advance_to_final_positions) ----
        if (t < nc) cons_writeback(w, A, K_slot[t]);
        if (t < nb) body_writeback(w, B_gid[t], v3(B_lin[t]), v3(B_ang[t]), q4(B_rot[t]), v3(B_trans[t]));
    }
}

void rp_launch_islands_build(const DevWorld &w, hipStream_t st) {
    hipLaunchKernelGGL(k_islands_build, dim3(1), dim3(1024), 0, st, w);
}
void rp_launch_island_solve(const DevWorld &w, hipStream_t st, int grid, int has_restitution) {
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_island_solve, dim3(grid), dim3(ISL_THREADS), 0, st, w, has_restitution);
}
