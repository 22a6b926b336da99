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
            w.isl_nb[id] = cnb; w.isl_nc[id] = cnc; w.isl_fill_b[id] = 0; w.isl_fill_c[id] = 0; w.isl_sorted[id] = 0; w.isl_nstages[id] = 0;
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

// ---- register-resident constraint of one island thread ------------------------------------------
// Thread t owns solver manifold t of its island for the whole step.  Everything the colour-ordered
// sweeps touch stays in the thread's VGPRs; the solver bodies shared between manifolds (velocity +
// pose, 64 B each) and the builder's body-local points (read once per substep) live in LDS.
//
// The reference fuses the pose-dependent `update` / `refresh_rhs_wo_bias` into the colour sweeps
// (contact_with_twist_friction.rs:426-554).  Those parts read poses only, and poses change only in the
// integrate stage, so here they run ONCE per substep for all manifolds in parallel (isl_pose_stage,
// right after integrate) instead of once per colour stage on the Gauss-Seidel critical path.  The
// distance computed there serves both the relax sweep of substep s and the biased sweeps of substep
// s+1 (same expression on the same poses; tangent_velocity is identically zero without contact
// modification hooks, which are outside this ABI).  Every f32 expression is evaluated exactly as in
// rp_constraint.h, so the result stays bit-identical to the per-colour launch path and the oracle.
struct IslPoint {
    V3 a, b, c, d;             // torque_dir1, torque_dir2, ii_torque_dir1, ii_torque_dir2
    float r, seed, d0;         // projected mass, restitution seed, builder dist
    float lam, acc, rhs, cfm;  // impulse, impulse accumulator, active rhs / cfm factor
    float rhsR, rhsB, cfmB;    // rhs of the relax sweep; rhs / cfm of the next biased sweeps
};
struct IslCons {
    int id1, id2, n, cids;
    V3 dir, dim1, dim2, im1, im2, t0, t1, twa, twb;
    float mu, twist_r, k11, k22, k12, inv_det, rhs_wo0, rhs_wo1;
    float td[4];
    float tw_imp, tw_acc, t_imp0, t_imp1, t_acc0, t_acc1, t_rhs0, t_rhs1, tb0, tb1;
    V3 T[8];
    IslPoint P[4];
    float cfm_factor, erp_inv_dt;
};
struct IslLds {
    float4 *lin, *ang, *rot, *trans;   // [RP_ISL_NB_MAX] solver bodies
    float4 *E, *F;                     // [4][RP_ISL_NC_MAX] builder local_p1 / local_p2
    float4 *B0, *B1;                   // [RP_ISL_NC_MAX] builder local friction centres
};
// scratch accessor cons_generate writes into (all plane indices are compile-time constants)
struct GenAcc {
    float4 (&R)[CP_COUNT];
    int &m1, &m2, &mn, &mcid;
    const IslLds &L;
    RP_DEV GenAcc(float4 (&R_)[CP_COUNT], int &a, int &b, int &c, int &d, const IslLds &L_) : R(R_), m1(a), m2(b), mn(c), mcid(d), L(L_) {}
    RP_DEV void st(int plane, float4 v) const { R[plane] = v; }
    RP_DEV void set_meta(int a, int b, int cnt, int cid) const { m1 = a; m2 = b; mn = cnt; mcid = cid; }
    RP_DEV Vel vel(int id) const {
        Vel v;
        if (id < 0) { v.lin = v3(0, 0, 0); v.ang = v3(0, 0, 0); } else { v.lin = v3(L.lin[id]); v.ang = v3(L.ang[id]); }
        return v;
    }
    RP_DEV Xf xf(int id) const {
        Xf x;
        if (id < 0) { x.r = q4(0, 0, 0, 1); x.t = v3(0, 0, 0); } else { x.r = q4(L.rot[id]); x.t = v3(L.trans[id]); }
        return x;
    }
};
RP_DEV Vel isl_vel(const IslLds &L, int id) {
    Vel v;
    if (id < 0) { v.lin = v3(0, 0, 0); v.ang = v3(0, 0, 0); } else { v.lin = v3(L.lin[id]); v.ang = v3(L.ang[id]); }
    return v;
}
RP_DEV void isl_set_vel(const IslLds &L, int id, const Vel &v) { if (id >= 0) { L.lin[id] = f4(v.lin, 0.0f); L.ang[id] = f4(v.ang, 0.0f); } }
RP_DEV Xf isl_xf(const IslLds &L, int id) {
    Xf x;
    if (id < 0) { x.r = q4(0, 0, 0, 1); x.t = v3(0, 0, 0); } else { x.r = q4(L.rot[id]); x.t = v3(L.trans[id]); }
    return x;
}

// generate (S1) + unpack into registers / LDS
RP_DEV bool isl_generate(const DevWorld &w, IslCons &c, const IslLds &L, int t, int slot, int g1, int g2, int l1, int l2) {
    float4 R[CP_COUNT];
    GenAcc G(R, c.id1, c.id2, c.n, c.cids, L);
    bool bouncy = cons_generate(w, G, slot, g1, g2, l1, l2);
    c.dir = v3(R[CP_H0]); c.mu = R[CP_H0].w;
    c.im1 = v3(R[CP_H1]); c.twist_r = R[CP_H1].w;
    c.im2 = v3(R[CP_H2]);
    Sym3 ii1 = {R[CP_H3].x, R[CP_H3].y, R[CP_H3].z, R[CP_H3].w, R[CP_H4].x, R[CP_H4].y};
    Sym3 ii2 = {R[CP_H4].z, R[CP_H4].w, R[CP_H5].x, R[CP_H5].y, R[CP_H5].z, R[CP_H5].w};
    c.t0 = v3(R[CP_H6]); c.rhs_wo0 = R[CP_H6].w; c.rhs_wo1 = R[CP_H7].x;
    c.k11 = R[CP_H7].y; c.k22 = R[CP_H7].z;
    c.td[0] = R[CP_H8].x; c.td[1] = R[CP_H8].y; c.td[2] = R[CP_H8].z; c.td[3] = R[CP_H8].w;
    c.tw_imp = R[CP_HM0].x; c.tw_acc = R[CP_HM0].y; c.t_imp0 = R[CP_HM0].z; c.t_imp1 = R[CP_HM0].w;
    c.t_acc0 = R[CP_HM1].x; c.t_acc1 = R[CP_HM1].y; c.t_rhs0 = R[CP_HM1].z; c.t_rhs1 = R[CP_HM1].w;
#pragma unroll
    for (int q = 0; q < 8; ++q) c.T[q] = v3(R[CP_T0 + q]);
    L.B0[t] = R[CP_B0]; L.B1[t] = R[CP_B1];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= c.n) break;
        IslPoint &p = c.P[k];
        p.rhs = R[NPL(k, NP_M)].x; p.cfm = R[NPL(k, NP_M)].y; p.lam = R[NPL(k, NP_M)].z; p.acc = R[NPL(k, NP_M)].w;
        p.a = v3(R[NPL(k, NP_A)]); p.r = R[NPL(k, NP_A)].w;
        p.b = v3(R[NPL(k, NP_B)]); p.seed = R[NPL(k, NP_B)].w;
        p.c = v3(R[NPL(k, NP_C)]); p.d0 = R[NPL(k, NP_C)].w;
        p.d = v3(R[NPL(k, NP_D)]);
        L.E[k * RP_ISL_NC_MAX + t] = R[NPL(k, NP_E)];
        L.F[k * RP_ISL_NC_MAX + t] = R[NPL(k, NP_F)];
    }
    // loop invariants of the sweeps (same expressions the per-colour path re-evaluates every sweep)
    c.t1 = cross(c.dir, c.t0);
    c.dim1 = cmul(c.dir, c.im1); c.dim2 = cmul(c.dir, c.im2);
    c.twa = sym_mul(ii1, c.dir); c.twb = sym_mul(ii2, c.dir);
    c.k12 = R[CP_H2].w * 0.5f;
    c.inv_det = rp_inv(c.k11 * c.k22 - c.k12 * c.k12);
    bool is_static = c.id1 < 0 || c.id2 < 0;
    float fstatic = is_static ? 1.0f : 0.0f;
    c.cfm_factor = w.prm.dyn_cfm + fstatic * (w.prm.static_cfm - w.prm.dyn_cfm);
    c.erp_inv_dt = w.prm.dyn_erp_inv_dt + fstatic * (w.prm.static_erp_inv_dt - w.prm.dyn_erp_inv_dt);
    return bouncy;
}

// Pose-dependent half of update / refresh_rhs_wo_bias (contact_with_twist_friction.rs:426-554), for the
// poses currently in LDS.  `solved_dt` only scales the (zero) tangent velocity.
RP_DEV void isl_pose_stage(const DevWorld &w, IslCons &c, const IslLds &L, int t, float solved_dt) {
    Xf x1 = isl_xf(L, c.id1), x2 = isl_xf(L, c.id2);
    V3 tangent_delta = v3(0.0f, 0.0f, 0.0f) * solved_dt;
    float inv_dt = w.prm.inv_dt_sub, maxcv = w.prm.max_corrective_velocity;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= c.n) break;
        IslPoint &p = c.P[k];
        V3 p1 = xf_tp(x1, v3(L.E[k * RP_ISL_NC_MAX + t])) + tangent_delta;
        V3 p2 = xf_tp(x2, v3(L.F[k * RP_ISL_NC_MAX + t]));
        float dist = p.d0 + dot(p1 - p2, c.dir);
        float rhs_wo_bias = rp_max(dist, 0.0f) * inv_dt;
        float rhs_bias = rp_clamp(dist * c.erp_inv_dt, -maxcv, 0.0f);
        p.rhsR = rhs_wo_bias;
        p.rhsB = rhs_wo_bias + rhs_bias;
        p.cfmB = dist <= 0.0f ? c.cfm_factor : 1.0f;
    }
    V3 p1 = xf_tp(x1, v3(L.B0[t])) + tangent_delta;
    V3 p2 = xf_tp(x2, v3(L.B1[t]));
    c.tb0 = dot(p1 - p2, c.t0) * inv_dt; c.tb1 = dot(p1 - p2, c.t1) * inv_dt;
}

// Velocity-dependent half of update + warmstart (:426-522, :633-678), colour-ordered.
RP_DEV void isl_warmstart(const DevWorld &w, IslCons &c, const IslLds &L) {
    float wc = w.prm.p.warmstart_coefficient;
    bool ws = wc != 0.0f;
    Vel v1 = isl_vel(L, c.id1), v2 = isl_vel(L, c.id2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= c.n) break;
        IslPoint &p = c.P[k];
        p.rhs = p.rhsB; p.cfm = p.cfmB;
        p.acc += p.lam;
        p.lam *= wc;
        if (ws) {
            v1.lin = v1.lin + c.dim1 * p.lam;
            v1.ang = v1.ang + p.c * p.lam;
            v2.lin = v2.lin + c.dim2 * (-p.lam);
            v2.ang = v2.ang + p.d * p.lam;
        }
    }
    c.t_rhs0 = c.rhs_wo0 + c.tb0; c.t_rhs1 = c.rhs_wo1 + c.tb1;
    c.t_acc0 += c.t_imp0; c.t_acc1 += c.t_imp1;
    c.t_imp0 *= wc; c.t_imp1 *= wc;
    c.tw_acc += c.tw_imp;
    c.tw_imp *= wc;
    if (ws) {
        float i0 = c.t_imp0, i1 = c.t_imp1;
        v1.lin = v1.lin + cmul(c.t0 * i0 + c.t1 * i1, c.im1);
        v1.ang = v1.ang + (c.T[4] * i0 + c.T[5] * i1);
        v2.lin = v2.lin + cmul(c.t0 * (-i0) + c.t1 * (-i1), c.im2);
        v2.ang = v2.ang + (c.T[6] * i0 + c.T[7] * i1);
        if (c.n > 1) {
            v1.ang = v1.ang + c.twa * c.tw_imp;
            v2.ang = v2.ang - c.twb * c.tw_imp;
        }
        isl_set_vel(L, c.id1, v1); isl_set_vel(L, c.id2, v2);
    }
}

// solve (:680-781); `relax` first switches to the bias-free right-hand sides of isl_pose_stage.
RP_DEV void isl_solve(IslCons &c, const IslLds &L, bool relax, bool friction) {
    Vel v1 = isl_vel(L, c.id1), v2 = isl_vel(L, c.id2);
    float imp[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= c.n) break;
        IslPoint &p = c.P[k];
        if (relax) { p.rhs = p.rhsR; p.cfm = 1.0f; }
        float dvel = dot(c.dir, v1.lin) + dot(p.a, v1.ang) - dot(c.dir, v2.lin) + dot(p.b, v2.ang) + p.rhs;
        float new_impulse = p.cfm * rp_max(p.lam - p.r * dvel, 0.0f);
        float dl = new_impulse - p.lam;
        p.lam = new_impulse;
        imp[k] = new_impulse;
        v1.lin = v1.lin + c.dim1 * dl;
        v1.ang = v1.ang + p.c * dl;
        v2.lin = v2.lin + c.dim2 * (-dl);
        v2.ang = v2.ang + p.d * dl;
    }
    if (friction) {
        if (relax) { c.t_rhs0 = c.rhs_wo0; c.t_rhs1 = c.rhs_wo1; }
        float tangent_limit = 0.0f, twist_limit = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (k >= c.n) break; tangent_limit += imp[k]; twist_limit += imp[k] * c.td[k]; }
        tangent_limit *= c.mu; twist_limit *= c.mu;
        if (c.n > 1) {
            float dvel = dot(c.dir, v1.ang - v2.ang) + 0.0f;
            float new_impulse = rp_clamp(c.tw_imp - c.twist_r * dvel, -twist_limit, twist_limit);
            float dl = new_impulse - c.tw_imp;
            c.tw_imp = new_impulse;
            v1.ang = v1.ang + c.twa * dl;
            v2.ang = v2.ang - c.twb * dl;
        }
        {
            float dvel_0 = dot(c.t0, v1.lin) + dot(c.T[0], v1.ang) - dot(c.t0, v2.lin) + dot(c.T[2], v2.ang) + c.t_rhs0;
            float dvel_1 = dot(c.t1, v1.lin) + dot(c.T[1], v1.ang) - dot(c.t1, v2.lin) + dot(c.T[3], v2.ang) + c.t_rhs1;
            float d0 = (c.k22 * dvel_0 - c.k12 * dvel_1) * c.inv_det;
            float d1 = (c.k11 * dvel_1 - c.k12 * dvel_0) * c.inv_det;
            float n0 = c.t_imp0 - d0, n1 = c.t_imp1 - d1;
            float l = sqrtf(n0 * n0 + n1 * n1);
            if (l > tangent_limit) { float sc = tangent_limit / l; n0 *= sc; n1 *= sc; }
            float dl0 = n0 - c.t_imp0, dl1 = n1 - c.t_imp1;
            c.t_imp0 = n0; c.t_imp1 = n1;
            v1.lin = v1.lin + cmul(c.t0 * dl0 + c.t1 * dl1, c.im1);
            v1.ang = v1.ang + (c.T[4] * dl0 + c.T[5] * dl1);
            v2.lin = v2.lin + cmul(c.t0 * (-dl0) + c.t1 * (-dl1), c.im2);
            v2.ang = v2.ang + (c.T[6] * dl0 + c.T[7] * dl1);
        }
    }
    isl_set_vel(L, c.id1, v1); isl_set_vel(L, c.id2, v2);
}

// apply_restitution (:568-597)
RP_DEV void isl_restitution(IslCons &c, const IslLds &L) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (k >= c.n) break; any |= c.P[k].seed < 0.0f; }
    if (!any) return;
    Vel v1 = isl_vel(L, c.id1), v2 = isl_vel(L, c.id2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= c.n) break;
        IslPoint &p = c.P[k];
        float dvel = dot(c.dir, v1.lin) + dot(p.a, v1.ang) - dot(c.dir, v2.lin) + dot(p.b, v2.ang) + p.seed;
        bool gate = p.seed < 0.0f && (p.acc + p.lam) > 0.0f;
        float new_impulse = gate ? rp_max(p.lam - p.r * dvel, 0.0f) : p.lam;
        float dl = new_impulse - p.lam;
        p.lam = new_impulse;
        v1.lin = v1.lin + c.dim1 * dl;
        v1.ang = v1.ang + p.c * dl;
        v2.lin = v2.lin + c.dim2 * (-dl);
        v2.ang = v2.ang + p.d * dl;
    }
    isl_set_vel(L, c.id1, v1); isl_set_vel(L, c.id2, v2);
}

// writeback_impulses (:783-829)
RP_DEV void isl_writeback(const DevWorld &w, const IslCons &c, int s) {
    V3 wtw = c.t0 * c.t_imp0 + c.t1 * c.t_imp1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= c.n) break;
        int cid = (c.cids >> (8 * k)) & 0xff;
        PT(w.pt_imp, cid, s) = make_float4(c.P[k].acc + c.P[k].lam, c.P[k].lam, c.tw_imp, 0.0f);
        PT(w.pt_wst, cid, s) = f4(wtw, 0.0f);
    }
}

#define ISL_THREADS 192

// Sort an island's manifold list by sweep stage (rank of the pair's colour) and hand every manifold
// its local stage index; overflow-colour manifolds (serial in the reference, worker 0) each get a
// stage of their own after all colour stages.  Runs once per layout change (the result is cached in
// isl_cons / isl_cstage / isl_nstages), inside the solve kernel's own workgroup.
RP_DEV void island_sort(const DevWorld &w, int isl, int nc, int cb, int nst_global, int *T_slot, int *T_rank, int *K_slot, int *K_rank) {
    const int t = threadIdx.x;
    if (t < nc) {
        int s = w.isl_cons[cb + t];
        int color = w.p_color[s];
        T_slot[t] = s;
        T_rank[t] = color >= RP_COLOR_OVERFLOW ? nst_global : w.color_rank[color];
    }
    __syncthreads();
    if (t < nc) {
        int r = T_rank[t], s = T_slot[t], posn = 0;
        // ties broken by pair slot so the serial overflow order does not depend on atomics
        for (int j = 0; j < nc; ++j) { int rj = T_rank[j]; posn += (rj < r) || (rj == r && T_slot[j] < s); }
        K_slot[posn] = s; K_rank[posn] = r;
    }
    __syncthreads();
    if (t == 0) {
        int q = -1, prev = -1;
        for (int i = 0; i < nc; ++i) {
            int r = K_rank[i];
            if (r >= nst_global || r != prev) ++q;
            prev = r;
            w.isl_cons[cb + i] = K_slot[i];
            w.isl_cstage[cb + i] = q;
        }
        w.isl_nstages[isl] = q + 1;
        __threadfence();
        w.isl_sorted[isl] = 1;
    }
    __syncthreads();
}

// One workgroup = one island.
__global__ void __launch_bounds__(ISL_THREADS) k_island_solve(DevWorld w, int has_restitution, int fast) {
    if (fast && w.flags[FL_FAST_ABORT]) return; // fast graph gave up on this step (rp_api.hip)
    __shared__ float4 B_lin[RP_ISL_NB_MAX], B_ang[RP_ISL_NB_MAX], B_rot[RP_ISL_NB_MAX], B_trans[RP_ISL_NB_MAX];
    __shared__ float4 L_E[4 * RP_ISL_NC_MAX], L_F[4 * RP_ISL_NC_MAX], L_B0[RP_ISL_NC_MAX], L_B1[RP_ISL_NC_MAX];
    __shared__ int S_a[RP_ISL_NC_MAX], S_b[RP_ISL_NC_MAX], S_c[RP_ISL_NC_MAX], S_d[RP_ISL_NC_MAX];
    __shared__ int any_bouncy;

    const int t = threadIdx.x;
    const int n_islands = w.flags[FL_N_ISLANDS];
    const int nst_global = w.flags[FL_N_STAGES];
    const rp_integration_params &prm = w.prm.p;
    const bool fib = prm.friction_in_bias_pass || prm.num_internal_stabilization_iterations == 0;
    IslLds L;
    L.lin = B_lin; L.ang = B_ang; L.rot = B_rot; L.trans = B_trans; L.E = L_E; L.F = L_F; L.B0 = L_B0; L.B1 = L_B1;

    for (int isl = blockIdx.x; isl < n_islands; isl += gridDim.x) {
        const int nb = w.isl_nb[isl], nc = w.isl_nc[isl];
        const int bb = w.isl_body_begin[isl], cb = w.isl_cons_begin[isl];
        __syncthreads(); // previous island of this workgroup fully written back
        if (!w.isl_sorted[isl]) island_sort(w, isl, nc, cb, nst_global, S_a, S_b, S_c, S_d);
        // ---- bodies -> LDS (+ per-body constants in the owning thread's registers) (S0) ----
        int b_gid = -1, b_fl = 0;
        V3 b_incl = v3(0, 0, 0), b_inca = b_incl, b_invpi = b_incl; Q4 b_pframe = q4(0, 0, 0, 1);
        if (t < nb) {
            int g = w.isl_bodies[bb + t];
            V3 lin, ang, trans; Q4 rot;
            body_begin(w, g, lin, ang, rot, trans, b_incl, b_inca);
            b_gid = g; b_fl = w.b_flags[g];
            B_lin[t] = f4(lin, 0.0f); B_ang[t] = f4(ang, 0.0f); B_rot[t] = f4(rot); B_trans[t] = f4(trans, 0.0f);
            b_invpi = v3(w.b_invpi[g]); b_pframe = q4(w.b_pframe[g]);
        }
        if (t == 0) any_bouncy = 0;
        const int nls = w.isl_nstages[isl];
        int slot = -1, myq = -1;
        if (t < nc) { slot = w.isl_cons[cb + t]; myq = w.isl_cstage[cb + t]; }
        __syncthreads();
        IslCons c;
        c.n = 0; c.id1 = -1; c.id2 = -1; c.cids = 0;
        // ---- generate (S1) + pose stage for the initial poses ----
        if (t < nc) {
            int rb1 = w.c_parent[w.p_c1[slot]], rb2 = w.c_parent[w.p_c2[slot]];
            int rel_dom = w.p_reldom[slot];
            int g1 = (is_dyn(w, rb1) && rel_dom <= 0) ? rb1 : -1;
            int g2 = (is_dyn(w, rb2) && rel_dom >= 0) ? rb2 : -1;
            int l1 = g1 >= 0 ? w.b_local[g1] : -1, l2 = g2 >= 0 ? w.b_local[g2] : -1;
            if (isl_generate(w, c, L, t, slot, g1, g2, l1, l2)) any_bouncy = 1;
            isl_pose_stage(w, c, L, t, 0.0f);
        }

        for (int sub = 0; sub < w.prm.num_substeps; ++sub) {
            float solved_dt = (float)sub * w.prm.dt_sub;
            __syncthreads(); // pose stage read rot/trans; relax sweep of the previous substep done
            if (t < nb) { // S2
                V3 lin = v3(B_lin[t]), ang = v3(B_ang[t]);
                body_increment(w, b_fl, lin, ang, q4(B_rot[t]), b_incl, b_inca, b_invpi, b_pframe);
                B_lin[t] = f4(lin, 0.0f); B_ang[t] = f4(ang, 0.0f);
            }
            __syncthreads();
            for (int q = 0; q < nls; ++q) { if (myq == q) isl_warmstart(w, c, L); __syncthreads(); }
            for (int it = 0; it < prm.num_internal_pgs_iterations; ++it)
                for (int q = 0; q < nls; ++q) { if (myq == q) isl_solve(c, L, false, fib); __syncthreads(); }
            if (t < nb) { // S6
                V3 lin = v3(B_lin[t]), ang = v3(B_ang[t]), trans = v3(B_trans[t]); Q4 rot = q4(B_rot[t]);
                body_integrate(w, b_fl, lin, ang, rot, trans);
                B_lin[t] = f4(lin, 0.0f); B_ang[t] = f4(ang, 0.0f); B_rot[t] = f4(rot); B_trans[t] = f4(trans, 0.0f);
            }
            __syncthreads();
            if (t < nc) isl_pose_stage(w, c, L, t, solved_dt + w.prm.dt_sub);
            for (int it = 0; it < prm.num_internal_stabilization_iterations; ++it)
                for (int q = 0; q < nls; ++q) { if (myq == q) isl_solve(c, L, true, true); __syncthreads(); }
        }
        if (has_restitution && any_bouncy)
            for (int q = 0; q < nls; ++q) { if (myq == q) isl_restitution(c, L); __syncthreads(); }
        // ---- write-back (S9, S10, advance_to_final_positions) ----
        if (t < nc) isl_writeback(w, c, slot);
        if (t < nb) body_writeback(w, b_gid, v3(B_lin[t]), v3(B_ang[t]), q4(B_rot[t]), v3(B_trans[t]));
    }
}

void rp_launch_islands_build(const DevWorld &w, hipStream_t st) {
    hipLaunchKernelGGL(k_islands_build, dim3(1), dim3(1024), 0, st, w);
}
void rp_launch_island_solve(const DevWorld &w, hipStream_t st, int grid, int has_restitution, int fast) {
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_island_solve, dim3(grid), dim3(ISL_THREADS), 0, st, w, has_restitution, fast);
}
