// rp_coulomb.h — FrictionModel::Coulomb: one Coulomb friction constraint per contact point.
//
// Restates ContactWithCoulombFrictionBuilder::{generate, update, refresh_rhs_wo_bias} and
// ContactWithCoulombFriction::{warmstart, solve, writeback_impulses}
// (/root/reference/src/dynamics/solver/contact_constraint/contact_with_coulomb_friction.rs:52-760) with the
// element solves of contact_constraint_element.rs:64-176 (ContactConstraintTangentPart: exact coupled 2x2
// tangent solve, capped at mu * lambda_k of ITS point) and :226-310 (ContactConstraintNormalPart).
// Selected by IntegrationParameters::friction_model (staged_island_solver/init.rs:419); the default
// Simplified model is rp_constraint.h.  Its 87 planes per manifold do not fit the registers of k_island_solve (which holds the twist
// constraint only): Coulomb islands run on k_island_generic (rp_islands.hip: one workgroup per island, these functions over thread-private
// HBM rows, bodies in LDS), everything else on the global path of rp_solver.hip / rp_flow.hip.
// The normal parts share the twist model's planes (NP_*); each point adds 9 tangent planes behind CP_COUNT.
#pragma once
#include "rp_constraint.h"

enum {
    CQ_TD10 = 0, // tangent torque_dir1[0].xyz, rhs[0]
    CQ_TD11,     // tangent torque_dir1[1].xyz, rhs[1]
    CQ_TD20,     // tangent torque_dir2[0].xyz, rhs_wo_bias[0]
    CQ_TD21,     // tangent torque_dir2[1].xyz, rhs_wo_bias[1]
    CQ_I10,      // ii_torque_dir1[0].xyz, r[0]
    CQ_I11,      // ii_torque_dir1[1].xyz, r[1]
    CQ_I20,      // ii_torque_dir2[0].xyz, r[2]
    CQ_I21,      // ii_torque_dir2[1].xyz, -
    CQ_M         // mutable: impulse[0], impulse[1], impulse_accumulator[0], [1]
};
static_assert(CQ_M + 1 == CQ_PER_POINT, "tangent planes per point (rp_world.h)");
#define CQL(k, sub) (CP_COUNT + CQ_PER_POINT * (k) + (sub))

// generate — contact_with_coulomb_friction.rs:52-300
template <class Acc>
RP_DEV bool coul_generate(const DevWorld &w, const Acc &A, int s, int gid1, int gid2, int id1, int id2) {
    constexpr int RP_UNR = Acc::PRELOAD ? 4 : 1; // the point loops are unrolled only where the preloaded rows need static indices
    Vel vels1 = A.vel(id1), vels2 = A.vel(id2);
    Xf poses1 = A.xf(id1), poses2 = A.xf(id2);
    V3 im1 = gid1 >= 0 ? v3(w.b_eim[gid1]) : v3(0, 0, 0), im2 = gid2 >= 0 ? v3(w.b_eim[gid2]) : v3(0, 0, 0);
    Sym3 ii1 = load_ii(w, gid1), ii2 = load_ii(w, gid2);
    V3 world_com1 = poses1.t, world_com2 = poses2.t;
    float4 nf = w.p_normal[s];
    V3 force_dir1 = -v3(nf);
    float friction = nf.w;
    float restitution = w.p_misc[s].x;
    int count = w.p_nsc[s]; if (count > 4) count = 4;
    V3 t0 = orthonormal_vector(force_dir1); // compute_tangent_contact_directions, contact_constraint/mod.rs:27-46
    V3 t1 = cross(force_dir1, t0);
    int cids = 0;
    bool bouncy_seed = false;
    V3 imsum = im1 + im2;
    // inputs of every point first, rows after (see cons_generate)
    float4 in_a1[4], in_a2[4], in_imp[4], in_wst[4], in_dp1[4], in_dp2[4];
    if (Acc::PRELOAD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < count) { in_a1[k] = PT(w.sc_a1, k, s); in_a2[k] = PT(w.sc_a2, k, s); }
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < count) {
            const int cid = __float_as_int(in_a2[k].w);
            in_imp[k] = PT(w.pt_imp, cid, s); in_wst[k] = PT(w.pt_wst, cid, s); in_dp1[k] = PT(w.pt_dp1, cid, s); in_dp2[k] = PT(w.pt_dp2, cid, s);
        }
    }
#pragma unroll RP_UNR
    for (int k = 0; k < 4; ++k) {
        if (k >= count) break;
        float4 a1 = Acc::PRELOAD ? in_a1[k] : PT(w.sc_a1, k, s), a2 = Acc::PRELOAD ? in_a2[k] : PT(w.sc_a2, k, s);
        int cid = __float_as_int(a2.w);
        cids |= (cid & 0xff) << (8 * k);
        float4 pimp = Acc::PRELOAD ? in_imp[k] : PT(w.pt_imp, cid, s);
        V3 wt = v3(Acc::PRELOAD ? in_wst[k] : PT(w.pt_wst, cid, s));
        float warmstart_impulse = pimp.y;
        float wti0 = dot(wt, t0), wti1 = dot(wt, t1);
        bool is_new = pimp.x == 0.0f;
        float is_bouncy = is_new ? (restitution > 0.0f ? 1.0f : 0.0f) : (restitution >= 1.0f ? 1.0f : 0.0f);
        V3 p1 = xf_tp(poses1, v3(a1));
        V3 p2 = xf_tp(poses2, v3(a2));
        float dist = dot(p1 - p2, force_dir1);
        V3 dp1 = v3(Acc::PRELOAD ? in_dp1[k] : PT(w.pt_dp1, cid, s)), dp2 = v3(Acc::PRELOAD ? in_dp2[k] : PT(w.pt_dp2, cid, s));
        V3 vel1 = vels1.lin + cross(vels1.ang, dp1);
        V3 vel2 = vels2.lin + cross(vels2.ang, dp2);
        {
            V3 torque_dir1 = cross(dp1, force_dir1);
            V3 torque_dir2 = cross(dp2, -force_dir1);
            V3 ii_torque_dir1 = sym_mul(ii1, torque_dir1);
            V3 ii_torque_dir2 = sym_mul(ii2, torque_dir2);
            float projected_mass = rp_inv(dot(force_dir1, cmul(imsum, force_dir1)) + dot(ii_torque_dir1, torque_dir1) + dot(ii_torque_dir2, torque_dir2));
            float projected_velocity = dot(vel1 - vel2, force_dir1);
            float restitution_seed = is_bouncy * restitution * projected_velocity;
            bouncy_seed |= restitution_seed < 0.0f;
            V3 point = world_com1 + dp1;
            float info_dist = dist - dot(point - (world_com2 + dp2), force_dir1);
            A.st(NPL(k, NP_M), make_float4(0.0f, 1.0f, warmstart_impulse, -warmstart_impulse));
            A.st(NPL(k, NP_A), f4(torque_dir1, projected_mass));
            A.st(NPL(k, NP_B), f4(torque_dir2, restitution_seed));
            A.st(NPL(k, NP_C), f4(ii_torque_dir1, info_dist));
            A.st(NPL(k, NP_D), f4(ii_torque_dir2, 0.0f));
            A.st(NPL(k, NP_E), f4(xf_itp(poses1, point), 0.0f));
            A.st(NPL(k, NP_F), f4(xf_itp(poses2, world_com2 + dp2), 0.0f));
        }
        V3 td1[2], td2[2], itd1[2], itd2[2]; float r[3];
        for (int j = 0; j < 2; ++j) {
            V3 tj = j == 0 ? t0 : t1;
            td1[j] = cross(dp1, tj);
            td2[j] = cross(dp2, -tj);
            itd1[j] = sym_mul(ii1, td1[j]);
            itd2[j] = sym_mul(ii2, td2[j]);
            r[j] = dot(tj, cmul(imsum, tj)) + dot(itd1[j], td1[j]) + dot(itd2[j], td2[j]);
        }
        r[2] = 2.0f * (dot(itd1[0], td1[1]) + dot(itd2[0], td2[1]));
        // tangent_velocity is zero in this scope (no contact-modification hooks); the products are kept so that the
        // sign of the zero matches the reference expression
        V3 tangent_velocity = v3(0, 0, 0);
        float rhs_wo0 = dot(tangent_velocity, t0), rhs_wo1 = dot(tangent_velocity, t1);
        A.st(CQL(k, CQ_TD10), f4(td1[0], rhs_wo0)); A.st(CQL(k, CQ_TD11), f4(td1[1], rhs_wo1));
        A.st(CQL(k, CQ_TD20), f4(td2[0], rhs_wo0)); A.st(CQL(k, CQ_TD21), f4(td2[1], rhs_wo1));
        A.st(CQL(k, CQ_I10), f4(itd1[0], r[0])); A.st(CQL(k, CQ_I11), f4(itd1[1], r[1]));
        A.st(CQL(k, CQ_I20), f4(itd2[0], r[2])); A.st(CQL(k, CQ_I21), f4(itd2[1], 0.0f));
        A.st(CQL(k, CQ_M), make_float4(wti0, wti1, -wti0, -wti1));
    }
    A.st(CP_H0, f4(force_dir1, friction));
    A.st(CP_H1, f4(im1, 0.0f));
    A.st(CP_H2, f4(im2, 0.0f));
    A.st(CP_H6, f4(t0, 0.0f));
    A.set_meta(id1, id2, count, cids);
    return bouncy_seed;
}

// update (:362-455) + warmstart (:561-603)
template <class Acc>
RP_DEV void coul_update_warmstart(const DevWorld &w, const Acc &A) {
    constexpr int RP_UNR = Acc::PRELOAD ? 4 : 1; // the point loops are unrolled only where the preloaded rows need static indices
    int id1 = A.id1(), id2 = A.id2(), n = A.n();
    bool is_static = id1 < 0 || id2 < 0;
    float fstatic = is_static ? 1.0f : 0.0f;
    float cfm_factor = w.prm.dyn_cfm + fstatic * (w.prm.static_cfm - w.prm.dyn_cfm);
    float erp_inv_dt = w.prm.dyn_erp_inv_dt + fstatic * (w.prm.static_erp_inv_dt - w.prm.dyn_erp_inv_dt);
    float inv_dt = w.prm.inv_dt_sub;
    float maxcv = w.prm.max_corrective_velocity;
    float wc = w.prm.p.warmstart_coefficient;
    Xf x1 = A.xf(id1), x2 = A.xf(id2);
    V3 dir1 = v3(A.ld(CP_H0)), t0 = v3(A.ld(CP_H6)), t1 = cross(dir1, t0);
    V3 im1 = v3(A.ld(CP_H1)), im2 = v3(A.ld(CP_H2));
    Vel v1 = A.vel(id1), v2 = A.vel(id2);
    bool ws = wc != 0.0f;
    float ti0[4] = {0, 0, 0, 0}, ti1[4] = {0, 0, 0, 0};
    // (Acc::PRELOAD: every row in before the first one goes out, see GlobalAccT in rp_constraint.h)
    float4 pm[4], pc[4], pd[4], pe[4], pf[4], qm[4], qa[4], qb[4], qc[4], qd[4], qi10[4], qi11[4], qi20[4], qi21[4];
    if (Acc::PRELOAD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < n) {
            pm[k] = A.ld(NPL(k, NP_M)); pc[k] = A.ld(NPL(k, NP_C)); pd[k] = A.ld(NPL(k, NP_D)); pe[k] = A.ld(NPL(k, NP_E)); pf[k] = A.ld(NPL(k, NP_F));
            qm[k] = A.ld(CQL(k, CQ_M)); qa[k] = A.ld(CQL(k, CQ_TD10)); qb[k] = A.ld(CQL(k, CQ_TD11)); qc[k] = A.ld(CQL(k, CQ_TD20)); qd[k] = A.ld(CQL(k, CQ_TD21));
            if (ws) { qi10[k] = A.ld(CQL(k, CQ_I10)); qi11[k] = A.ld(CQL(k, CQ_I11)); qi20[k] = A.ld(CQL(k, CQ_I20)); qi21[k] = A.ld(CQL(k, CQ_I21)); }
        }
    }
#pragma unroll RP_UNR
    for (int k = 0; k < 4; ++k) {
        if (k >= n) break;
        float4 m = ROWK(pm, k, NPL(k, NP_M));
        float4 c = ROWK(pc, k, NPL(k, NP_C)), d = ROWK(pd, k, NPL(k, NP_D));
        V3 p1 = xf_tp(x1, v3(ROWK(pe, k, NPL(k, NP_E))));
        V3 p2 = xf_tp(x2, v3(ROWK(pf, k, NPL(k, NP_F))));
        float dist = c.w + dot(p1 - p2, dir1);
        float rhs_wo_bias = rp_max(dist, 0.0f) * inv_dt;
        float rhs_bias = rp_clamp(dist * erp_inv_dt, -maxcv, 0.0f);
        m.x = rhs_wo_bias + rhs_bias;
        m.y = dist <= 0.0f ? cfm_factor : 1.0f;
        m.w += m.z;
        m.z *= wc;
        A.st(NPL(k, NP_M), m);
        float4 tm = ROWK(qm, k, CQL(k, CQ_M));
        tm.z += tm.x; tm.w += tm.y;
        tm.x *= wc; tm.y *= wc;
        A.st(CQL(k, CQ_M), tm);
        ti0[k] = tm.x; ti1[k] = tm.y;
        float4 a = ROWK(qa, k, CQL(k, CQ_TD10)), b = ROWK(qb, k, CQL(k, CQ_TD11));
        a.w = ROWK(qc, k, CQL(k, CQ_TD20)).w + dot(p1 - p2, t0) * inv_dt;
        b.w = ROWK(qd, k, CQL(k, CQ_TD21)).w + dot(p1 - p2, t1) * inv_dt;
        A.st(CQL(k, CQ_TD10), a); A.st(CQL(k, CQ_TD11), b);
        if (ws) { // ContactConstraintNormalPart::warmstart, contact_constraint_element.rs:226-240
            v1.lin = v1.lin + cmul(dir1, im1) * m.z;
            v1.ang = v1.ang + v3(c) * m.z;
            v2.lin = v2.lin + cmul(dir1, im2) * (-m.z);
            v2.ang = v2.ang + v3(d) * m.z;
        }
    }
    if (ws) {
#pragma unroll RP_UNR
        for (int k = 0; k < 4; ++k) { // ContactConstraintTangentPart::warmstart, :64-97
            if (k >= n) break;
            float i0 = ti0[k], i1 = ti1[k];
            v1.lin = v1.lin + cmul(t0 * i0 + t1 * i1, im1);
            v1.ang = v1.ang + (v3(ROWK(qi10, k, CQL(k, CQ_I10))) * i0 + v3(ROWK(qi11, k, CQL(k, CQ_I11))) * i1);
            v2.lin = v2.lin + cmul(t0 * (-i0) + t1 * (-i1), im2);
            v2.ang = v2.ang + (v3(ROWK(qi20, k, CQL(k, CQ_I20))) * i0 + v3(ROWK(qi21, k, CQL(k, CQ_I21))) * i1);
        }
        A.set_vel(id1, v1); A.set_vel(id2, v2);
    }
}

// solve (:605-690) (+ refresh_rhs_wo_bias :460-489 when `refresh`)
template <class Acc>
RP_DEV void coul_solve(const DevWorld &w, const Acc &A, bool refresh, bool friction) {
    constexpr int RP_UNR = Acc::PRELOAD ? 4 : 1; // the point loops are unrolled only where the preloaded rows need static indices
    int id1 = A.id1(), id2 = A.id2(), n = A.n();
    float4 h0 = A.ld(CP_H0);
    V3 dir1 = v3(h0);
    V3 im1 = v3(A.ld(CP_H1)), im2 = v3(A.ld(CP_H2));
    Vel v1 = A.vel(id1), v2 = A.vel(id2);
    Xf x1, x2;
    x1.r = q4(0, 0, 0, 1); x1.t = v3(0, 0, 0); x2 = x1;
    if (refresh) { x1 = A.xf(id1); x2 = A.xf(id2); }
    float imp[4] = {0, 0, 0, 0};
    // (Acc::PRELOAD: every row in before the first one goes out, see GlobalAccT in rp_constraint.h)
    float4 pm[4], pa[4], pb[4], pc[4], pd[4], pe[4], pf[4];
    float4 r10[4], r11[4], r20[4], r21[4], s10[4], s11[4], s20[4], s21[4], rm[4];
    float4 ph6 = make_float4(0, 0, 0, 0);
    if (Acc::PRELOAD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < n) {
            pm[k] = A.ld(NPL(k, NP_M)); pa[k] = A.ld(NPL(k, NP_A)); pb[k] = A.ld(NPL(k, NP_B)); pc[k] = A.ld(NPL(k, NP_C)); pd[k] = A.ld(NPL(k, NP_D));
            if (refresh) { pe[k] = A.ld(NPL(k, NP_E)); pf[k] = A.ld(NPL(k, NP_F)); }
            if (friction) {
                r10[k] = A.ld(CQL(k, CQ_TD10)); r11[k] = A.ld(CQL(k, CQ_TD11)); r20[k] = A.ld(CQL(k, CQ_TD20)); r21[k] = A.ld(CQL(k, CQ_TD21));
                s10[k] = A.ld(CQL(k, CQ_I10)); s11[k] = A.ld(CQL(k, CQ_I11)); s20[k] = A.ld(CQL(k, CQ_I20)); s21[k] = A.ld(CQL(k, CQ_I21));
                rm[k] = A.ld(CQL(k, CQ_M));
            }
        }
        if (friction) ph6 = A.ld(CP_H6);
    }
#pragma unroll RP_UNR
    for (int k = 0; k < 4; ++k) {
        if (k >= n) break;
        float4 m = ROWK(pm, k, NPL(k, NP_M));
        float4 a = ROWK(pa, k, NPL(k, NP_A)), b = ROWK(pb, k, NPL(k, NP_B)), c = ROWK(pc, k, NPL(k, NP_C)), d = ROWK(pd, k, NPL(k, NP_D));
        if (refresh) {
            V3 p1 = xf_tp(x1, v3(ROWK(pe, k, NPL(k, NP_E))));
            V3 p2 = xf_tp(x2, v3(ROWK(pf, k, NPL(k, NP_F))));
            float dist = c.w + dot(p1 - p2, dir1);
            m.x = rp_max(dist, 0.0f) * w.prm.inv_dt_sub;
            m.y = 1.0f;
        }
        float dvel = dot(dir1, v1.lin) + dot(v3(a), v1.ang) - dot(dir1, v2.lin) + dot(v3(b), v2.ang) + m.x;
        float new_impulse = m.y * rp_max(m.z - a.w * dvel, 0.0f);
        float dl = new_impulse - m.z;
        m.z = new_impulse;
        imp[k] = new_impulse;
        A.st(NPL(k, NP_M), m);
        v1.lin = v1.lin + cmul(dir1, im1) * dl;
        v1.ang = v1.ang + v3(c) * dl;
        v2.lin = v2.lin + cmul(dir1, im2) * (-dl);
        v2.ang = v2.ang + v3(d) * dl;
    }
    if (friction) {
        V3 t0 = v3(ROW1(ph6, CP_H6)), t1 = cross(dir1, t0);
#pragma unroll RP_UNR
        for (int k = 0; k < 4; ++k) { // ContactConstraintTangentPart::solve, contact_constraint_element.rs:100-176
            if (k >= n) break;
            float limit = h0.w * imp[k];
            float4 td10 = ROWK(r10, k, CQL(k, CQ_TD10)), td11 = ROWK(r11, k, CQL(k, CQ_TD11)), td20 = ROWK(r20, k, CQL(k, CQ_TD20)), td21 = ROWK(r21, k, CQL(k, CQ_TD21));
            float4 i10 = ROWK(s10, k, CQL(k, CQ_I10)), i11 = ROWK(s11, k, CQL(k, CQ_I11)), i20 = ROWK(s20, k, CQL(k, CQ_I20)), i21 = ROWK(s21, k, CQL(k, CQ_I21));
            float4 tm = ROWK(rm, k, CQL(k, CQ_M));
            if (refresh) { td10.w = td20.w; td11.w = td21.w; A.st(CQL(k, CQ_TD10), td10); A.st(CQL(k, CQ_TD11), td11); }
            float dvel_0 = dot(t0, v1.lin) + dot(v3(td10), v1.ang) - dot(t0, v2.lin) + dot(v3(td20), v2.ang) + td10.w;
            float dvel_1 = dot(t1, v1.lin) + dot(v3(td11), v1.ang) - dot(t1, v2.lin) + dot(v3(td21), v2.ang) + td11.w;
            float k11 = i10.w, k22 = i11.w, k12 = i20.w * 0.5f;
            float inv_det = rp_inv(k11 * k22 - k12 * k12);
            float d0 = (k22 * dvel_0 - k12 * dvel_1) * inv_det;
            float d1 = (k11 * dvel_1 - k12 * dvel_0) * inv_det;
            float n0 = tm.x - d0, n1 = tm.y - d1;
            float l = sqrtf(n0 * n0 + n1 * n1); // nalgebra simd_cap_magnitude(limit)
            if (l > limit) { float sc = limit / l; n0 *= sc; n1 *= sc; }
            float dl0 = n0 - tm.x, dl1 = n1 - tm.y;
            tm.x = n0; tm.y = n1;
            A.st(CQL(k, CQ_M), tm);
            v1.lin = v1.lin + cmul(t0 * dl0 + t1 * dl1, im1);
            v1.ang = v1.ang + (v3(i10) * dl0 + v3(i11) * dl1);
            v2.lin = v2.lin + cmul(t0 * (-dl0) + t1 * (-dl1), im2);
            v2.ang = v2.ang + (v3(i20) * dl0 + v3(i21) * dl1);
        }
    }
    A.set_vel(id1, v1); A.set_vel(id2, v2);
}

// writeback_impulses (:692-760): per-point world-space friction impulse; the twist warm start is left untouched
template <class Acc>
RP_DEV void coul_writeback(const DevWorld &w, const Acc &A, int s) {
    int n = A.n(), cids = A.cids();
    V3 dir1 = v3(A.ld(CP_H0)), t0 = v3(A.ld(CP_H6)), t1 = cross(dir1, t0);
    for (int k = 0; k < 4; ++k) {
        if (k >= n) break;
        int cid = (cids >> (8 * k)) & 0xff;
        float4 m = A.ld(NPL(k, NP_M)), tm = A.ld(CQL(k, CQ_M));
        float4 old = PT(w.pt_imp, cid, s);
        V3 wtw = t0 * rp_canon0(tm.x) + t1 * rp_canon0(tm.y); // canonicalised like every stored impulse (:690-711)
        PT(w.pt_imp, cid, s) = make_float4(rp_canon0(m.w + m.z), rp_canon0(m.z), old.z, 0.0f);
        PT(w.pt_wst, cid, s) = f4(v3(rp_canon0(wtw.x), rp_canon0(wtw.y), rp_canon0(wtw.z)), 0.0f);
    }
}

// ---- model dispatch used by the global path: a compile-time switch, so the default (twist) kernels carry no Coulomb
// code (their register / scratch budget is unchanged); the host launches the variant of IntegrationParameters::friction_model
RP_DEV bool coulomb_model(const DevWorld &w) { return w.prm.p.friction_model == RP_FRICTION_COULOMB; }
template <bool COUL, class Acc>
RP_DEV void cons_apply_model(const DevWorld &w, const Acc &A, int mode, bool friction_in_bias, float solved_dt) {
    if (!COUL) { cons_apply(w, A, mode, friction_in_bias, solved_dt); return; }
    if (mode == MODE_WARMSTART) coul_update_warmstart(w, A);
    else if (mode == MODE_BIAS) coul_solve(w, A, false, friction_in_bias);
    else if (mode == MODE_RELAX) coul_solve(w, A, true, true);
    else cons_restitution(w, A); // normal parts only: same planes as the twist model
}
