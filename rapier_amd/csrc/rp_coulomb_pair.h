// rp_coulomb_pair.h — FrictionModel::Coulomb by a PAIR of lanes: the register-resident constraint of the per-island megakernel
// (rp_islands.hip, island_solve_body<WIDE, COUL = true>) for ContactWithCoulombFriction
// (/root/reference/src/dynamics/solver/contact_constraint/contact_with_coulomb_friction.rs:52-760; element solves
// contact_constraint_element.rs:64-176 tangent part, :226-310 normal part).
//
// Same split as rp_lanepair.h: the even lane owns body 1's half of every row, the odd lane body 2's half, the halves of a relative
// velocity meet through DPP quad permutes, the even lane evaluates the impulse and broadcasts it.  What differs from the twist model:
// every contact POINT carries its own coupled 2x2 tangent constraint (capped at mu * lambda of ITS point), there is no friction centre
// and no twist row, and a warm start adds 4 normal + 4 tangent terms per manifold (16 rows of W instead of 11).
// Every f32 expression is the one of rp_coulomb.h (single lane, rows in HBM: k_island_generic and the global path) with
//   a * (-b) == (-a) * b,  x - y == x + (-y)
// so the results stay bit-identical to that form and to the oracle.  Pose-dependent halves (update :362-455, refresh_rhs_wo_bias
// :460-489) run once per substep for all manifolds (coul_pose_stage), as isl_pose_stage does for the twist model.
#pragma once
#include "rp_lanepair.h"

struct SidePointC {
    V3 pa, pc;                         // own torque_dir, ii_torque_dir of the normal part
    float r, seed, d0;                 // even lane: projected mass, restitution seed, dist - geometric dist at generate
    float lam, acc;                    // even lane: impulse, impulse_accumulator
    float rhsR, rhsB, cfmB;            // even lane: bias-free rhs; biased rhs and cfm (the sweeps pick by `relax`)
    V3 td0, td1, itd0, itd1;           // own tangent torque dirs and their inertia products
    float k11, k22, k12, inv_det;      // even lane: the 2x2 tangent system (r[0], r[1], r[2] / 2, 1 / det)
    float t_imp0, t_imp1, t_acc0, t_acc1, t_rhs0, t_rhs1; // even lane (t_rhs: the BIASED rhs; the relaxed sweeps use rhs_wo)
};
struct IslSideC {
    int id, n, cids;
    bool odd;
    V3 dir, t0, t1;                    // both lanes
    V3 sdim, im;                       // even: dim1, im1 ; odd: -dim2, im2
    float mu, rhs_wo0, rhs_wo1;        // even lane
    float cfm_factor, erp_inv_dt;
    SidePointC P[4];
};
#define WS_SLOTS_COUL 16 // per lane: 4 x (lin, ang) normal terms, then 4 x (lin, ang) tangent terms — the order coul_update_warmstart adds them in

// generate (:52-300) by the lane pair (see isl_generate for the conventions)
RP_DEV bool isl_generate(const DevWorld &w, IslSideC &h, const IslLds &L, int m, int s, int gid, int lid, bool odd, bool is_static) {
    h.odd = odd; h.id = lid;
    Vel vels = isl_vel(L, lid);
    Xf pose = isl_xf(L, lid);
    V3 im = gid >= 0 ? v3(w.b_eim[gid]) : v3(0, 0, 0);
    Sym3 ii = load_ii(w, gid);
    V3 world_com = pose.t;
    float4 nf = w.p_normal[s];
    V3 dir = -v3(nf);
    V3 sdir = odd ? -dir : dir;
    float restitution = w.p_misc[s].x;
    int count = w.p_nsc[s]; if (count > 4) count = 4;
    V3 t0 = orthonormal_vector(dir); // contact_constraint/mod.rs:27-46
    V3 t1 = cross(dir, t0);
    int cids = 0;
    bool bouncy_seed = false;
    V3 imsum = im + dppv<DPP_FROM_ODD>(im);   // even lane: im1 + im2
    h.n = count; h.dir = dir; h.t0 = t0; h.t1 = t1; h.im = im; h.mu = nf.w;
    const float4 *anchors = odd ? w.sc_a2 : w.sc_a1;
    const float4 *levers = odd ? w.pt_dp2 : w.pt_dp1;
    float4 *LP = odd ? L.F : L.E;
    // tangent_velocity is zero in this scope (no contact-modification hooks); the products keep the sign of the zero
    V3 tangent_velocity = v3(0, 0, 0);
    h.rhs_wo0 = dot(tangent_velocity, t0); h.rhs_wo1 = dot(tangent_velocity, t1);
    const float nn = dot(dir, cmul(imsum, dir)), tt0 = dot(t0, cmul(imsum, t0)), tt1 = dot(t1, cmul(imsum, t1));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= count) break;
        SidePointC &q = h.P[k];
        float4 an = PT(anchors, k, s);
        int cid = __float_as_int(PT(w.sc_a2, k, s).w);
        cids |= (cid & 0xff) << (8 * k);
        float4 pimp = PT(w.pt_imp, cid, s);
        V3 wt = v3(PT(w.pt_wst, cid, s));
        float warmstart_impulse = pimp.y;
        float wti0 = dot(wt, t0), wti1 = dot(wt, t1);
        bool is_new = pimp.x == 0.0f;
        float is_bouncy = is_new ? (restitution > 0.0f ? 1.0f : 0.0f) : (restitution >= 1.0f ? 1.0f : 0.0f);
        V3 pw = xf_tp(pose, v3(an));
        float dist = dot(pw - dppv<DPP_FROM_ODD>(pw), dir);
        V3 dp = v3(PT(levers, cid, s));
        V3 point = world_com + dp;
        V3 vel = vels.lin + cross(vels.ang, dp);
        V3 torque_dir = cross(dp, sdir);
        V3 ii_torque_dir = sym_mul(ii, torque_dir);
        float G = dot(ii_torque_dir, torque_dir);
        float projected_mass = rp_inv(nn + G + dppf<DPP_FROM_ODD>(G));
        float projected_velocity = dot(vel - dppv<DPP_FROM_ODD>(vel), dir);
        float restitution_seed = is_bouncy * restitution * projected_velocity;
        bouncy_seed |= restitution_seed < 0.0f;
        float info_dist = dist - dot(point - dppv<DPP_FROM_ODD>(point), dir);
        q.lam = warmstart_impulse; q.acc = -warmstart_impulse;
        q.pa = torque_dir; q.r = projected_mass; q.seed = restitution_seed;
        q.pc = ii_torque_dir; q.d0 = info_dist;
        q.rhsR = 0.0f; q.rhsB = 0.0f; q.cfmB = 1.0f;
        LP[k * RP_ISL_NC_MAX + m] = f4(xf_itp(pose, point), 0.0f);
        // the point's own tangent constraint (:200-260)
        V3 td0 = cross(dp, odd ? -t0 : t0), td1 = cross(dp, odd ? -t1 : t1);
        V3 itd0 = sym_mul(ii, td0), itd1 = sym_mul(ii, td1);
        float G0 = dot(itd0, td0), G1 = dot(itd1, td1), K = dot(itd0, td1);
        q.k11 = tt0 + G0 + dppf<DPP_FROM_ODD>(G0);
        q.k22 = tt1 + G1 + dppf<DPP_FROM_ODD>(G1);
        q.k12 = (2.0f * (K + dppf<DPP_FROM_ODD>(K))) * 0.5f;
        q.inv_det = rp_inv(q.k11 * q.k22 - q.k12 * q.k12);
        q.td0 = td0; q.td1 = td1; q.itd0 = itd0; q.itd1 = itd1;
        q.t_imp0 = wti0; q.t_imp1 = wti1; q.t_acc0 = -wti0; q.t_acc1 = -wti1;
        q.t_rhs0 = h.rhs_wo0; q.t_rhs1 = h.rhs_wo1;
    }
    h.cids = cids;
    V3 dim = cmul(dir, im);
    h.sdim = odd ? -dim : dim;
    float fstatic = is_static ? 1.0f : 0.0f;
    h.cfm_factor = w.prm.dyn_cfm + fstatic * (w.prm.static_cfm - w.prm.dyn_cfm);
    h.erp_inv_dt = w.prm.dyn_erp_inv_dt + fstatic * (w.prm.static_erp_inv_dt - w.prm.dyn_erp_inv_dt);
    return bouncy_seed;
}

// pose-dependent half of update (:362-455) / refresh_rhs_wo_bias (:460-489) for the poses currently in LDS
RP_DEV void isl_pose_stage(const DevWorld &w, IslSideC &h, const IslLds &L, int m, float /*solved_dt: scales the zero tangent velocity only*/) {
    Xf x = isl_xf(L, h.id);
    float inv_dt = w.prm.inv_dt_sub, maxcv = w.prm.max_corrective_velocity;
    const float4 *LP = h.odd ? L.F : L.E;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= h.n) break;
        SidePointC &p = h.P[k];
        V3 pw = xf_tp(x, v3(LP[k * RP_ISL_NC_MAX + m]));
        V3 dpw = pw - dppv<DPP_FROM_ODD>(pw);          // p1 - p2 (even lane)
        float dist = p.d0 + dot(dpw, h.dir);
        float rhs_wo_bias = rp_max(dist, 0.0f) * inv_dt;
        float rhs_bias = rp_clamp(dist * h.erp_inv_dt, -maxcv, 0.0f);
        p.rhsR = rhs_wo_bias;
        p.rhsB = rhs_wo_bias + rhs_bias;
        p.cfmB = dist <= 0.0f ? h.cfm_factor : 1.0f;
        p.t_rhs0 = h.rhs_wo0 + dot(dpw, h.t0) * inv_dt;
        p.t_rhs1 = h.rhs_wo1 + dot(dpw, h.t1) * inv_dt;
    }
}

// the warm-start terms of this lane (update + warmstart :362-455, :561-603), dense rows (see isl_ws_terms<true>)
RP_DEV void isl_ws_terms_coul(const DevWorld &w, IslSideC &h, float4 *W, int t) {
    float wc = w.prm.p.warmstart_coefficient;
    bool ws = wc != 0.0f;
    const float4 z = make_float4(-0.0f, -0.0f, -0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= h.n) {
            if (ws) for (int j = k; j < 4; ++j) { W[(2 * j) * WS_STRIDE + t] = z; W[(2 * j + 1) * WS_STRIDE + t] = z; W[(8 + 2 * j) * WS_STRIDE + t] = z; W[(9 + 2 * j) * WS_STRIDE + t] = z; }
            break;
        }
        SidePointC &p = h.P[k];
        p.acc += p.lam;
        p.lam *= wc;
        p.t_acc0 += p.t_imp0; p.t_acc1 += p.t_imp1;
        p.t_imp0 *= wc; p.t_imp1 *= wc;
        if (ws) {
            float lam = dppf<DPP_FROM_EVEN>(p.lam);
            W[(2 * k) * WS_STRIDE + t] = f4(h.sdim * lam, 0.0f); W[(2 * k + 1) * WS_STRIDE + t] = f4(p.pc * lam, 0.0f);
            float i0 = dppf<DPP_FROM_EVEN>(p.t_imp0), i1 = dppf<DPP_FROM_EVEN>(p.t_imp1);
            float s0 = h.odd ? -i0 : i0, s1 = h.odd ? -i1 : i1;
            W[(8 + 2 * k) * WS_STRIDE + t] = f4(cmul(h.t0 * s0 + h.t1 * s1, h.im), 0.0f);
            W[(9 + 2 * k) * WS_STRIDE + t] = f4(p.itd0 * i0 + p.itd1 * i1, 0.0f);
        }
    }
}
RP_DEV void isl_ws_accumulate_lin_coul(const float4 *W, int begin, int count, V3 &lin) {
#pragma unroll 2
    for (int e = 0; e < count; ++e) {
        const int row = begin + e;
        const float4 l0 = W[0 * WS_STRIDE + row], l1 = W[2 * WS_STRIDE + row], l2 = W[4 * WS_STRIDE + row], l3 = W[6 * WS_STRIDE + row];
        const float4 t0 = W[8 * WS_STRIDE + row], t1 = W[10 * WS_STRIDE + row], t2 = W[12 * WS_STRIDE + row], t3 = W[14 * WS_STRIDE + row];
        lin = lin + v3(l0); lin = lin + v3(l1); lin = lin + v3(l2); lin = lin + v3(l3);
        lin = lin + v3(t0); lin = lin + v3(t1); lin = lin + v3(t2); lin = lin + v3(t3);
    }
}
RP_DEV void isl_ws_accumulate_ang_coul(const float4 *W, int begin, int count, V3 &ang) {
#pragma unroll 2
    for (int e = 0; e < count; ++e) {
        const int row = begin + e;
        const float4 a0 = W[1 * WS_STRIDE + row], a1 = W[3 * WS_STRIDE + row], a2 = W[5 * WS_STRIDE + row], a3 = W[7 * WS_STRIDE + row];
        const float4 t0 = W[9 * WS_STRIDE + row], t1 = W[11 * WS_STRIDE + row], t2 = W[13 * WS_STRIDE + row], t3 = W[15 * WS_STRIDE + row];
        ang = ang + v3(a0); ang = ang + v3(a1); ang = ang + v3(a2); ang = ang + v3(a3);
        ang = ang + v3(t0); ang = ang + v3(t1); ang = ang + v3(t2); ang = ang + v3(t3);
    }
}

// solve (:605-690): the four normal parts, then the four tangent parts (contact_constraint_element.rs:100-176)
template <bool F4> RP_DEV void isl_solve_coul_t(IslSideC &h, const IslLds &L, bool relax, bool friction) {
    const int hn = F4 ? 4 : h.n;
    Vel v = isl_vel(L, h.id);
    float imp[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= hn) break;
        SidePointC &p = h.P[k];
        const float rhs = relax ? p.rhsR : p.rhsB, cfm = relax ? 1.0f : p.cfmB;
        float X = dot(h.dir, v.lin), Y = dot(p.pa, v.ang);
        float S = X + Y;
        float dvel = S - dppf<DPP_FROM_ODD>(X) + dppf<DPP_FROM_ODD>(Y) + rhs;
        float new_impulse = cfm * rp_max(p.lam - p.r * dvel, 0.0f);
        float dl = dppf<DPP_FROM_EVEN>(new_impulse - p.lam);
        p.lam = new_impulse;
        imp[k] = new_impulse;
        v.lin = v.lin + h.sdim * dl;
        v.ang = v.ang + p.pc * dl;
    }
    if (friction) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= hn) break;
            SidePointC &p = h.P[k];
            const float limit = h.mu * imp[k];
            const float rhs0 = relax ? h.rhs_wo0 : p.t_rhs0, rhs1 = relax ? h.rhs_wo1 : p.t_rhs1;
            float X0 = dot(h.t0, v.lin), Y0 = dot(p.td0, v.ang), X1 = dot(h.t1, v.lin), Y1 = dot(p.td1, v.ang);
            float S0 = X0 + Y0, S1 = X1 + Y1;
            float dvel_0 = S0 - dppf<DPP_FROM_ODD>(X0) + dppf<DPP_FROM_ODD>(Y0) + rhs0;
            float dvel_1 = S1 - dppf<DPP_FROM_ODD>(X1) + dppf<DPP_FROM_ODD>(Y1) + rhs1;
            float d0 = (p.k22 * dvel_0 - p.k12 * dvel_1) * p.inv_det;
            float d1 = (p.k11 * dvel_1 - p.k12 * dvel_0) * p.inv_det;
            float n0 = p.t_imp0 - d0, n1 = p.t_imp1 - d1;
            float l = sqrtf(n0 * n0 + n1 * n1); // nalgebra simd_cap_magnitude(limit)
            if (l > limit) { float sc = limit / l; n0 *= sc; n1 *= sc; }
            float dl0 = dppf<DPP_FROM_EVEN>(n0 - p.t_imp0), dl1 = dppf<DPP_FROM_EVEN>(n1 - p.t_imp1);
            p.t_imp0 = n0; p.t_imp1 = n1;
            float s0 = h.odd ? -dl0 : dl0, s1 = h.odd ? -dl1 : dl1;
            v.lin = v.lin + cmul(h.t0 * s0 + h.t1 * s1, h.im);
            v.ang = v.ang + (p.itd0 * dl0 + p.itd1 * dl1);
        }
    }
    isl_set_vel(L, h.id, v);
}
RP_DEV void isl_solve(IslSideC &h, const IslLds &L, bool relax, bool friction) {
    if (__all(h.n == 4)) isl_solve_coul_t<true>(h, L, relax, friction); else isl_solve_coul_t<false>(h, L, relax, friction);
}

// writeback_impulses (:692-760) — even lane: per-point world-space friction impulse; the twist warm start stays what it was
RP_DEV void isl_writeback(const DevWorld &w, const IslSideC &h, int s) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= h.n) break;
        const SidePointC &p = h.P[k];
        int cid = (h.cids >> (8 * k)) & 0xff;
        float4 old = PT(w.pt_imp, cid, s);
        V3 wtw = h.t0 * rp_canon0(p.t_imp0) + h.t1 * rp_canon0(p.t_imp1);
        PT(w.pt_imp, cid, s) = make_float4(rp_canon0(p.acc + p.lam), rp_canon0(p.lam), old.z, 0.0f);
        PT(w.pt_wst, cid, s) = f4(v3(rp_canon0(wtw.x), rp_canon0(wtw.y), rp_canon0(wtw.z)), 0.0f);
    }
}
