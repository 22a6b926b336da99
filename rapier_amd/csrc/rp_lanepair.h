// rp_lanepair.h — two lanes per manifold: the solve of ContactWithTwistFriction (contact_with_twist_friction.rs:680-781) by a PAIR of
// adjacent lanes, shared by the per-island megakernel (rp_islands.hip: constraints resident in registers for the whole step) and the
// LDS tile sweeps of the global path (rp_tiles.hip: rows fetched per stage).  Solver bodies live in LDS (IslLds).
#pragma once
#include "rp_constraint.h"

struct IslLds {
    float4 *lin, *ang, *rot, *trans;   // [RP_ISL_NB_MAX] solver bodies
    float4 *E, *F;                     // [4][RP_ISL_NC_MAX] builder local_p1 / local_p2
    float4 *B0, *B1;                   // [RP_ISL_NC_MAX] builder local friction centres
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

// ---- two lanes per manifold -----------------------------------------------------------------------
// The sweeps are latency-bound (one wave walks a dependent instruction stream per colour stage), so
// each manifold is solved by a PAIR of adjacent lanes: the even lane owns body 1's half of every row
// (direction, torque arms, velocity), the odd lane body 2's half.  The two halves of each relative
// velocity meet through DPP quad permutes (no LDS, no extra wave), the even lane evaluates the impulse
// and broadcasts it back.  The operations and their order are exactly those of rp_constraint.h:
//   dvel = (((n.v1 + t1.w1) - n.v2) + t2.w2) + rhs           a * (-b) == (-a) * b,  x - y == x + (-y)
// so the results stay bit-identical to the single-lane form and to the oracle.
#define ISL_LANES (2 * RP_ISL_NC_MAX)   // lanes 2m, 2m+1 = manifold m
#define ISL_THREADS 512                  // the lanes beyond an island's 2 * nc only validate the step (fused fast path)
#define ISL_THREADS_DENSE ISL_LANES      // the dense form of k_island_solve (rp_islands.hip): the manifold lanes and nothing else
#define DPP_FROM_ODD 0xF5   // quad_perm [1,1,3,3]: both lanes of a pair read the odd lane
#define DPP_FROM_EVEN 0xA0  // quad_perm [0,0,2,2]: both lanes of a pair read the even lane
template <int CTRL> RP_DEV float dppf(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL> RP_DEV int dppi(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
template <int CTRL> RP_DEV V3 dppv(V3 v) { return v3(dppf<CTRL>(v.x), dppf<CTRL>(v.y), dppf<CTRL>(v.z)); }
RP_DEV float sel(bool odd, float o, float e) { return odd ? o : e; }
RP_DEV V3 sel(bool odd, V3 o, V3 e) { return v3(odd ? o.x : e.x, odd ? o.y : e.y, odd ? o.z : e.z); }

struct SidePoint {
    V3 pa, pc;                 // own torque_dir, ii_torque_dir (a,c on the even lane; b,d on the odd lane)
    float r, seed, d0;         // even lane
    float lam, acc, rhs, cfm;  // even lane
    float rhsR, rhsB, cfmB;    // even lane
};
struct IslSide {
    int id, n, cids;           // own body (LDS index or -1); point count (both lanes); contact ids (even)
    bool odd;
    V3 dir, t0, t1;            // both lanes
    V3 sdim, im, stw;          // even: dim1, im1, twa ; odd: -dim2, im2, -twb
    V3 td0, td1, itd0, itd1;   // own tangent torque dirs: T[0],T[1],T[4],T[5] | T[2],T[3],T[6],T[7]
    float mu, twist_r, k11, k22, k12, inv_det, rhs_wo0, rhs_wo1;   // even lane
    float td[4];
    float tw_imp, tw_acc, t_imp0, t_imp1, t_acc0, t_acc1, t_rhs0, t_rhs1, tb0, tb1;
    float cfm_factor, erp_inv_dt;
    SidePoint P[4];
};

// solve (:680-781); `relax` first switches to the bias-free right-hand sides of isl_pose_stage.
template <bool F4> RP_DEV void isl_solve_t(IslSide &h, const IslLds &L, bool relax, bool friction) {
    const int hn = F4 ? 4 : h.n;
    Vel v = isl_vel(L, h.id);

    float imp[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= hn) break;
        SidePoint &p = h.P[k];
        if (relax) { p.rhs = p.rhsR; p.cfm = 1.0f; }
        float X = dot(h.dir, v.lin), Y = dot(p.pa, v.ang);
        float S = X + Y;
        float dvel = S - dppf<DPP_FROM_ODD>(X) + dppf<DPP_FROM_ODD>(Y) + p.rhs;
        float new_impulse = p.cfm * rp_max(p.lam - p.r * dvel, 0.0f);
        float dl = dppf<DPP_FROM_EVEN>(new_impulse - p.lam);
        p.lam = new_impulse;
        imp[k] = new_impulse;
        v.lin = v.lin + h.sdim * dl;
        v.ang = v.ang + p.pc * dl;
    }
    if (friction) {
        if (relax) { h.t_rhs0 = h.rhs_wo0; h.t_rhs1 = h.rhs_wo1; }
        float tangent_limit = 0.0f, twist_limit = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (k >= hn) break; tangent_limit += imp[k]; twist_limit += imp[k] * h.td[k]; }
        tangent_limit *= h.mu; twist_limit *= h.mu;
        if (hn > 1) {
            V3 w2 = dppv<DPP_FROM_ODD>(v.ang);
            float dvel = dot(h.dir, v.ang - w2) + 0.0f;
            float new_impulse = rp_clamp(h.tw_imp - h.twist_r * dvel, -twist_limit, twist_limit);
            float dl = dppf<DPP_FROM_EVEN>(new_impulse - h.tw_imp);
            h.tw_imp = new_impulse;
            v.ang = v.ang + h.stw * dl;
        }
        {
            float X0 = dot(h.t0, v.lin), Y0 = dot(h.td0, v.ang), X1 = dot(h.t1, v.lin), Y1 = dot(h.td1, v.ang);
            float S0 = X0 + Y0, S1 = X1 + Y1;
            float dvel_0 = S0 - dppf<DPP_FROM_ODD>(X0) + dppf<DPP_FROM_ODD>(Y0) + h.t_rhs0;
            float dvel_1 = S1 - dppf<DPP_FROM_ODD>(X1) + dppf<DPP_FROM_ODD>(Y1) + h.t_rhs1;
            float d0 = (h.k22 * dvel_0 - h.k12 * dvel_1) * h.inv_det;
            float d1 = (h.k11 * dvel_1 - h.k12 * dvel_0) * h.inv_det;
            float n0 = h.t_imp0 - d0, n1 = h.t_imp1 - d1;
            float l = sqrtf(n0 * n0 + n1 * n1);
            if (l > tangent_limit) { float sc = tangent_limit / l; n0 *= sc; n1 *= sc; }
            float dl0 = dppf<DPP_FROM_EVEN>(n0 - h.t_imp0), dl1 = dppf<DPP_FROM_EVEN>(n1 - h.t_imp1);
            h.t_imp0 = n0; h.t_imp1 = n1;
            float s0 = h.odd ? -dl0 : dl0, s1 = h.odd ? -dl1 : dl1;
            v.lin = v.lin + cmul(h.t0 * s0 + h.t1 * s1, h.im);
            v.ang = v.ang + (h.itd0 * dl0 + h.itd1 * dl1);
        }
    }
    isl_set_vel(L, h.id, v);
}

// Wave-uniform dispatch: when every active manifold of this wave has 4 points (face/face contacts, the
// common case) the per-point exec-mask branches disappear.
RP_DEV void isl_solve(IslSide &h, const IslLds &L, bool relax, bool friction) {
    if (__all(h.n == 4)) isl_solve_t<true>(h, L, relax, friction); else isl_solve_t<false>(h, L, relax, friction);
}

