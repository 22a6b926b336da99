// rp_constraint.h — the per-manifold contact constraint kernels, written once and instantiated over
// two storage accessors:
//   * GlobalAcc — constraint planes and solver bodies in HBM (per-colour launch path, rp_solver.hip);
//   * LdsAcc    — constraint planes and solver bodies staged in the workgroup's LDS (per-island
//                 megakernel, rp_islands.hip).
// Restates ContactWithTwistFrictionBuilder::{generate, update, refresh_rhs_wo_bias, apply_restitution}
// and ContactWithTwistFriction::{warmstart, solve, writeback_impulses}
// (/root/reference/src/dynamics/solver/contact_constraint/contact_with_twist_friction.rs:58-829) with
// the Slim element solves of contact_constraint_element.rs:465-755, one SIMD lane per thread.
// The expression order is identical in both instantiations (and in the CPU oracle), so every path
// produces bit-identical f32 results.
#pragma once
#include "rp_world.h"

#define PT(plane, k, s) plane[(size_t)(k) * w.pool_cap + (s)]

struct Vel { V3 lin, ang; };
struct Xf { Q4 r; V3 t; };
RP_DEV V3 xf_tp(const Xf &x, V3 p) { return qrot(x.r, p) + x.t; }
RP_DEV V3 xf_itp(const Xf &x, V3 p) { return qrot_inv(x.r, p - x.t); }
RP_DEV Sym3 load_ii(const DevWorld &w, int gid) {
    Sym3 m = {0, 0, 0, 0, 0, 0};
    if (gid >= 0) { float4 a = w.b_eii0[gid], b = w.b_eii1[gid]; m.m11 = a.x; m.m12 = a.y; m.m13 = a.z; m.m22 = a.w; m.m23 = b.x; m.m33 = b.y; }
    return m;
}

// the plane that currently holds `plane`: the six mutable planes live in place (par = 0) or in the shadow planes behind CP_COUNT (par = 1)
RP_DEV int cplane_mut_index(int plane) { return plane == CP_HM0 ? 4 : (plane == CP_HM1 ? 5 : ((plane >= CP_N0 && (plane - CP_N0) % 7 == NP_M) ? (plane - CP_N0) / 7 : -1)); }
RP_DEV int cplane(int plane, int par) { const int m = cplane_mut_index(plane); return (m >= 0 && par) ? CP_COUNT + m : plane; }

// ---- accessor over HBM ------------------------------------------------------------------------
// PRELOAD (a trait of every accessor): the constraint functions fetch every input they read before storing their first row, instead
// of point by point.  Measured on b3d_large_pyramid: it pays in k_generate (whose inputs — solver contacts, tracked impulses, lever
// arms — come from other arrays than the rows it stores, so the compiler must keep every load behind the preceding stores: with all
// of them issued up front, and the whole register file instead of the 128 VGPRs + scratch a 1024-thread bound left it, 59 -> 36 us)
// and for the twist contacts of the dataflow launch; it does NOT pay in the colour-stage kernels (the rows of one constraint are
// distinct offsets from one base: the compiler already hoists those loads; k_stage stayed at 5.9 / 8.6 us) and costs the fused
// kernels their registers (k_island_generic 0.57 -> 1.67 ms in scratch spills) — those keep the plain form.
// Same operands, same order either way: only the issue order of the loads changes.
template <bool PRE>
struct GlobalAccT {
    static constexpr bool PRELOAD = PRE;
    const DevWorld &w; int pos;
    RP_DEV GlobalAccT(const DevWorld &w_, int pos_) : w(w_), pos(pos_) {}
    RP_DEV float4 ld(int plane) const { return w.C[(size_t)cplane(plane, w.c_par) * w.cons_cap + pos]; }
    RP_DEV void st(int plane, float4 v) const { w.C[(size_t)cplane(plane, w.c_par) * w.cons_cap + pos] = v; }
    RP_DEV int id1() const { return w.k_b1[pos]; }
    RP_DEV int id2() const { return w.k_b2[pos]; }
    RP_DEV int n() const { return w.k_n[pos]; }
    RP_DEV int cids() const { return w.k_cid[pos]; }
    RP_DEV void set_meta(int a, int b, int cnt, int cid) const { w.k_b1[pos] = a; w.k_b2[pos] = b; w.k_n[pos] = cnt; w.k_cid[pos] = cid; }
    RP_DEV Vel vel(int id) const {
        Vel v;
        if (id < 0) { v.lin = v3(0, 0, 0); v.ang = v3(0, 0, 0); } else { v.lin = v3(w.s_lin[id]); v.ang = v3(w.s_ang[id]); }
        return v;
    }
    RP_DEV void set_vel(int id, const Vel &v) const { if (id >= 0) { w.s_lin[id] = f4(v.lin, 0.0f); w.s_ang[id] = f4(v.ang, 0.0f); } }
    RP_DEV Xf xf(int id) const {
        Xf x;
        if (id < 0) { x.r = q4(0, 0, 0, 1); x.t = v3(0, 0, 0); } else { x.r = q4(w.s_rot[id]); x.t = v3(w.s_trans[id]); }
        return x;
    }
};
typedef GlobalAccT<false> GlobalAcc;
typedef GlobalAccT<true> GlobalAccP;
// row `plane` of point k: from the preloaded copy, or straight from the accessor
#define ROWK(arr, k, plane) (Acc::PRELOAD ? arr[k] : A.ld(plane))
#define ROW1(var, plane) (Acc::PRELOAD ? var : A.ld(plane))

#define NPL(k, sub) (CP_N0 + 7 * (k) + (sub))

// S1: generate.  `gid1/gid2` = arena indices of the two (dynamic, solver-attached) bodies or -1;
// `id1/id2` = the same bodies in the accessor's index space.
template <class Acc>
RP_DEV bool cons_generate(const DevWorld &w, const Acc &A, int s, int gid1, int gid2, int id1, int id2) {
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
    V3 t0 = orthonormal_vector(force_dir1); // contact_constraint/mod.rs:27-46
    V3 t1 = cross(force_dir1, t0);
    float inv_num_points = 1.0f / (float)count;

    V3 friction_center = v3(0, 0, 0), friction_center2 = v3(0, 0, 0), tangent_vel = v3(0, 0, 0);
    float twist_warmstart = 0.0f, tw0 = 0.0f, tw1 = 0.0f;
    V3 points0 = v3(0, 0, 0), points1 = points0, points2 = points0, points3 = points0;
    int cids = 0;
    bool bouncy_seed = false;
    V3 imsum = im1 + im2;
    // every input of every point is fetched before the first constraint row is stored: the rows may alias the inputs as far as the
    // compiler can tell, so loads issued after a store wait for it — four points x two dependent round trips, ~45 us of a 59 us
    // k_generate on b3d_large_pyramid before this
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
    float d0s0 = 0.0f, d0s1 = 0.0f, d0s2 = 0.0f, d0s3 = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= count) break;
        float weight = inv_num_points;
        float4 a1 = Acc::PRELOAD ? in_a1[k] : PT(w.sc_a1, k, s), a2 = Acc::PRELOAD ? in_a2[k] : PT(w.sc_a2, k, s);
        int cid = __float_as_int(a2.w);
        cids |= (cid & 0xff) << (8 * k);
        float4 pimp = Acc::PRELOAD ? in_imp[k] : PT(w.pt_imp, cid, s);
        V3 wt = v3(Acc::PRELOAD ? in_wst[k] : PT(w.pt_wst, cid, s));
        float warmstart_impulse = pimp.y;
        float wti0 = dot(wt, t0), wti1 = dot(wt, t1);
        float warmstart_twist_impulse = pimp.z;
        bool is_new = pimp.x == 0.0f;
        float is_bouncy = is_new ? (restitution > 0.0f ? 1.0f : 0.0f) : (restitution >= 1.0f ? 1.0f : 0.0f);
        V3 p1 = xf_tp(poses1, v3(a1));
        V3 p2 = xf_tp(poses2, v3(a2));
        float dist = dot(p1 - p2, force_dir1);
        V3 dp1 = v3(Acc::PRELOAD ? in_dp1[k] : PT(w.pt_dp1, cid, s)), dp2 = v3(Acc::PRELOAD ? in_dp2[k] : PT(w.pt_dp2, cid, s));
        V3 point = world_com1 + dp1;
        if (k == 0) points0 = point; else if (k == 1) points1 = point; else if (k == 2) points2 = point; else points3 = point;
        friction_center = friction_center + point * weight;
        friction_center2 = friction_center2 + (world_com2 + dp2) * weight;
        V3 vel1 = vels1.lin + cross(vels1.ang, dp1);
        V3 vel2 = vels2.lin + cross(vels2.ang, dp2);
        twist_warmstart += warmstart_twist_impulse * weight;
        tw0 += wti0 * weight; tw1 += wti1 * weight;
        // tangent_velocity is always zero in this scope (no contact-modification hooks)
        V3 torque_dir1 = cross(dp1, force_dir1);
        V3 torque_dir2 = cross(dp2, -force_dir1);
        V3 ii_torque_dir1 = sym_mul(ii1, torque_dir1);
        V3 ii_torque_dir2 = sym_mul(ii2, torque_dir2);
        float projected_mass = rp_inv(dot(force_dir1, cmul(imsum, force_dir1)) + dot(ii_torque_dir1, torque_dir1) + dot(ii_torque_dir2, torque_dir2));
        float projected_velocity = dot(vel1 - vel2, force_dir1);
        float restitution_seed = is_bouncy * restitution * projected_velocity;
        bouncy_seed |= restitution_seed < 0.0f;
        float info_dist = dist - dot(point - (world_com2 + dp2), force_dir1);
        if (k == 0) d0s0 = info_dist; else if (k == 1) d0s1 = info_dist; else if (k == 2) d0s2 = info_dist; else d0s3 = info_dist;
        A.st(NPL(k, NP_M), make_float4(0.0f, 1.0f, warmstart_impulse, -warmstart_impulse));
        A.st(NPL(k, NP_A), f4(torque_dir1, projected_mass));
        A.st(NPL(k, NP_B), f4(torque_dir2, restitution_seed));
        A.st(NPL(k, NP_C), f4(ii_torque_dir1, info_dist));
        A.st(NPL(k, NP_D), f4(ii_torque_dir2, 0.0f));
        A.st(NPL(k, NP_E), f4(xf_itp(poses1, point), 0.0f));
        A.st(NPL(k, NP_F), f4(xf_itp(poses2, world_com2 + dp2), 0.0f));
    }
    float twist_imp = count > 1 ? twist_warmstart : 0.0f;
    V3 dp1 = friction_center - world_com1, dp2 = friction_center2 - world_com2;
    float twist_r = 0.0f;
    float4 tdists = make_float4(0, 0, 0, 0);
    if (count > 1) {
        tdists.x = len(friction_center - points0);
        tdists.y = len(friction_center - points1);
        if (count > 2) tdists.z = len(friction_center - points2);
        if (count > 3) tdists.w = len(friction_center - points3);
        V3 ii_twist_dir1 = sym_mul(ii1, force_dir1);
        V3 ii_twist_dir2 = sym_mul(ii2, -force_dir1);
        twist_r = rp_inv(dot(ii_twist_dir1, force_dir1) + dot(ii_twist_dir2, -force_dir1));
    }
    V3 td1[2], td2[2], itd1[2], itd2[2]; float r[3], rhs_wo[2];
    for (int j = 0; j < 2; ++j) {
        V3 tj = j == 0 ? t0 : t1;
        td1[j] = cross(dp1, tj);
        td2[j] = cross(dp2, -tj);
        itd1[j] = sym_mul(ii1, td1[j]);
        itd2[j] = sym_mul(ii2, td2[j]);
        r[j] = dot(tj, cmul(imsum, tj)) + dot(itd1[j], td1[j]) + dot(itd2[j], td2[j]);
        rhs_wo[j] = dot(tangent_vel, tj);
    }
    r[2] = 2.0f * (dot(itd1[0], td1[1]) + dot(itd2[0], td2[1]));
    A.st(CP_H0, f4(force_dir1, friction));
    A.st(CP_H1, f4(im1, twist_r));
    A.st(CP_H2, f4(im2, r[2]));
    A.st(CP_H3, make_float4(ii1.m11, ii1.m12, ii1.m13, ii1.m22));
    A.st(CP_H4, make_float4(ii1.m23, ii1.m33, ii2.m11, ii2.m12));
    A.st(CP_H5, make_float4(ii2.m13, ii2.m22, ii2.m23, ii2.m33));
    A.st(CP_H6, f4(t0, rhs_wo[0]));
    // (the spare words of T0, T1, B2 and H7 carry a second copy of the points' builder distances: the tile sweeps recompute the
    // ii_torque_dir rows from the inertia instead of fetching them — rp_tiles.hip, tile_apply2 — and NP_C, which holds the distance, is one of them)
    A.st(CP_H7, make_float4(rhs_wo[1], r[0], r[1], d0s3));
    A.st(CP_H8, tdists);
    A.st(CP_HM0, make_float4(twist_imp, -twist_imp, tw0, tw1));
    A.st(CP_HM1, make_float4(-tw0, -tw1, rhs_wo[0], rhs_wo[1]));
    A.st(CP_T0, f4(td1[0], d0s0)); A.st(CP_T1, f4(td1[1], d0s1));
    A.st(CP_T2, f4(td2[0], 0.0f)); A.st(CP_T3, f4(td2[1], 0.0f));
    A.st(CP_T4, f4(itd1[0], 0.0f)); A.st(CP_T5, f4(itd1[1], 0.0f));
    A.st(CP_T6, f4(itd2[0], 0.0f)); A.st(CP_T7, f4(itd2[1], 0.0f));
    A.st(CP_B0, f4(xf_itp(poses1, friction_center), 0.0f));
    A.st(CP_B1, f4(xf_itp(poses2, friction_center2), 0.0f));
    A.st(CP_B2, f4(tangent_vel, d0s2));
    A.set_meta(id1, id2, count, cids);
    return bouncy_seed;
}

// update (+ warmstart): contact_with_twist_friction.rs:426-522 and :633-678
template <class Acc>
RP_DEV void cons_update_warmstart(const DevWorld &w, const Acc &A, float solved_dt) {
    int id1 = A.id1(), id2 = A.id2(), n = A.n();
    bool is_static = id1 < 0 || id2 < 0;
    float fstatic = is_static ? 1.0f : 0.0f;
    float cfm_factor = w.prm.dyn_cfm + fstatic * (w.prm.static_cfm - w.prm.dyn_cfm);
    float erp_inv_dt = w.prm.dyn_erp_inv_dt + fstatic * (w.prm.static_erp_inv_dt - w.prm.dyn_erp_inv_dt);
    float inv_dt = w.prm.inv_dt_sub;
    float maxcv = w.prm.max_corrective_velocity;
    float wc = w.prm.p.warmstart_coefficient;
    Xf x1 = A.xf(id1), x2 = A.xf(id2);
    float4 h0 = A.ld(CP_H0), h6 = A.ld(CP_H6);
    V3 dir1 = v3(h0), t0 = v3(h6), t1 = cross(dir1, t0);
    V3 tangent_delta = v3(A.ld(CP_B2)) * solved_dt;
    V3 im1 = v3(A.ld(CP_H1)), im2 = v3(A.ld(CP_H2));
    Vel v1 = A.vel(id1), v2 = A.vel(id2);
    bool ws = wc != 0.0f;
    // (Acc::PRELOAD: every row in before the first one goes out, see GlobalAccT)
    float4 pm[4], pc[4], pd[4], pe[4], pf[4];
    float4 phm0 = make_float4(0, 0, 0, 0), phm1 = phm0, ph7 = phm0, pb0 = phm0, pb1 = phm0, t4 = phm0, t5 = phm0, t6 = phm0, t7 = phm0, ph3 = phm0, ph4 = phm0, ph5 = phm0;
    if (Acc::PRELOAD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < n) { pm[k] = A.ld(NPL(k, NP_M)); pc[k] = A.ld(NPL(k, NP_C)); pd[k] = A.ld(NPL(k, NP_D)); pe[k] = A.ld(NPL(k, NP_E)); pf[k] = A.ld(NPL(k, NP_F)); }
        phm0 = A.ld(CP_HM0); phm1 = A.ld(CP_HM1); ph7 = A.ld(CP_H7); pb0 = A.ld(CP_B0); pb1 = A.ld(CP_B1);
        if (ws) { t4 = A.ld(CP_T4); t5 = A.ld(CP_T5); t6 = A.ld(CP_T6); t7 = A.ld(CP_T7); if (n > 1) { ph3 = A.ld(CP_H3); ph4 = A.ld(CP_H4); ph5 = A.ld(CP_H5); } }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= n) break;
        float4 m = ROWK(pm, k, NPL(k, NP_M));
        float4 c = ROWK(pc, k, NPL(k, NP_C)), d = ROWK(pd, k, NPL(k, NP_D));
        V3 p1 = xf_tp(x1, v3(ROWK(pe, k, NPL(k, NP_E)))) + tangent_delta;
        V3 p2 = xf_tp(x2, v3(ROWK(pf, k, NPL(k, NP_F))));
        float dist = c.w + dot(p1 - p2, dir1);
        float rhs_wo_bias = rp_max(dist, 0.0f) * inv_dt;
        float rhs_bias = rp_clamp(dist * erp_inv_dt, -maxcv, 0.0f);
        m.x = rhs_wo_bias + rhs_bias;
        m.y = dist <= 0.0f ? cfm_factor : 1.0f;
        m.w += m.z;
        m.z *= wc;
        A.st(NPL(k, NP_M), m);
        if (ws) { // ContactConstraintNormalPartSlim::warmstart, contact_constraint_element.rs:465-478
            v1.lin = v1.lin + cmul(dir1, im1) * m.z;
            v1.ang = v1.ang + v3(c) * m.z;
            v2.lin = v2.lin + cmul(dir1, im2) * (-m.z);
            v2.ang = v2.ang + v3(d) * m.z;
        }
    }
    float4 hm0 = ROW1(phm0, CP_HM0), hm1 = ROW1(phm1, CP_HM1), h7 = ROW1(ph7, CP_H7);
    {
        V3 p1 = xf_tp(x1, v3(ROW1(pb0, CP_B0))) + tangent_delta;
        V3 p2 = xf_tp(x2, v3(ROW1(pb1, CP_B1)));
        float bias0 = dot(p1 - p2, t0) * inv_dt, bias1 = dot(p1 - p2, t1) * inv_dt;
        hm1.z = h6.w + bias0; hm1.w = h7.x + bias1;
        hm1.x += hm0.z; hm1.y += hm0.w;
        hm0.z *= wc; hm0.w *= wc;
        hm0.y += hm0.x;
        hm0.x *= wc;
    }
    A.st(CP_HM0, hm0); A.st(CP_HM1, hm1);
    if (ws) {
        float i0 = hm0.z, i1 = hm0.w;
        v1.lin = v1.lin + cmul(t0 * i0 + t1 * i1, im1);
        v1.ang = v1.ang + (v3(ROW1(t4, CP_T4)) * i0 + v3(ROW1(t5, CP_T5)) * i1);
        v2.lin = v2.lin + cmul(t0 * (-i0) + t1 * (-i1), im2);
        v2.ang = v2.ang + (v3(ROW1(t6, CP_T6)) * i0 + v3(ROW1(t7, CP_T7)) * i1);
        if (n > 1) {
            float4 h3 = ROW1(ph3, CP_H3), h4 = ROW1(ph4, CP_H4), h5 = ROW1(ph5, CP_H5);
            Sym3 ii1 = {h3.x, h3.y, h3.z, h3.w, h4.x, h4.y}, ii2 = {h4.z, h4.w, h5.x, h5.y, h5.z, h5.w};
            v1.ang = v1.ang + sym_mul(ii1, dir1) * hm0.x;
            v2.ang = v2.ang - sym_mul(ii2, dir1) * hm0.x;
        }
        A.set_vel(id1, v1); A.set_vel(id2, v2);
    }
}

// solve: contact_with_twist_friction.rs:680-781 (+ refresh_rhs_wo_bias :529-554 when `refresh`)
template <class Acc>
RP_DEV void cons_solve(const DevWorld &w, const Acc &A, bool refresh, bool friction, float solved_dt) {
    int id1 = A.id1(), id2 = A.id2(), n = A.n();
    float4 h0 = A.ld(CP_H0);
    V3 dir1 = v3(h0);
    float4 h1 = A.ld(CP_H1), h2 = A.ld(CP_H2);
    V3 im1 = v3(h1), im2 = v3(h2);
    Vel v1 = A.vel(id1), v2 = A.vel(id2);
    Xf x1, x2; V3 tangent_delta = v3(0, 0, 0);
    x1.r = q4(0, 0, 0, 1); x1.t = v3(0, 0, 0); x2 = x1;
    if (refresh) { x1 = A.xf(id1); x2 = A.xf(id2); tangent_delta = v3(A.ld(CP_B2)) * solved_dt; }
    float imp[4] = {0, 0, 0, 0};
    // (Acc::PRELOAD: all rows in before the first one goes out — the four point solves and the friction solve then run back to back
    // instead of each waiting for its own loads)
    float4 pm[4], pa[4], pb[4], pc[4], pd[4], pe[4], pf[4];
    float4 ph6 = make_float4(0, 0, 0, 0), ph7 = ph6, ph8 = ph6, phm0 = ph6, phm1 = ph6, ph3 = ph6, ph4 = ph6, ph5 = ph6, q0 = ph6, q1 = ph6, q2 = ph6, q3 = ph6, q4_ = ph6, q5 = ph6, q6 = ph6, q7 = ph6;
    if (Acc::PRELOAD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < n) {
            pm[k] = A.ld(NPL(k, NP_M)); pa[k] = A.ld(NPL(k, NP_A)); pb[k] = A.ld(NPL(k, NP_B)); pc[k] = A.ld(NPL(k, NP_C)); pd[k] = A.ld(NPL(k, NP_D));
            if (refresh) { pe[k] = A.ld(NPL(k, NP_E)); pf[k] = A.ld(NPL(k, NP_F)); }
        }
        if (friction) {
            ph6 = A.ld(CP_H6); ph7 = A.ld(CP_H7); ph8 = A.ld(CP_H8); phm0 = A.ld(CP_HM0); phm1 = A.ld(CP_HM1);
            if (n > 1) { ph3 = A.ld(CP_H3); ph4 = A.ld(CP_H4); ph5 = A.ld(CP_H5); }
            q0 = A.ld(CP_T0); q1 = A.ld(CP_T1); q2 = A.ld(CP_T2); q3 = A.ld(CP_T3); q4_ = A.ld(CP_T4); q5 = A.ld(CP_T5); q6 = A.ld(CP_T6); q7 = A.ld(CP_T7);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= n) break;
        float4 m = ROWK(pm, k, NPL(k, NP_M));
        float4 a = ROWK(pa, k, NPL(k, NP_A)), b = ROWK(pb, k, NPL(k, NP_B)), c = ROWK(pc, k, NPL(k, NP_C)), d = ROWK(pd, k, NPL(k, NP_D));
        if (refresh) {
            V3 p1 = xf_tp(x1, v3(ROWK(pe, k, NPL(k, NP_E)))) + tangent_delta;
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
        float4 h6 = ROW1(ph6, CP_H6), h7 = ROW1(ph7, CP_H7), h8 = ROW1(ph8, CP_H8);
        float4 hm0 = ROW1(phm0, CP_HM0), hm1 = ROW1(phm1, CP_HM1);
        if (refresh) { hm1.z = h6.w; hm1.w = h7.x; }
        V3 t0 = v3(h6), t1 = cross(dir1, t0);
        float tdist[4] = {h8.x, h8.y, h8.z, h8.w};
        float tangent_limit = 0.0f, twist_limit = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (k >= n) break; tangent_limit += imp[k]; twist_limit += imp[k] * tdist[k]; }
        tangent_limit *= h0.w; twist_limit *= h0.w;
        if (n > 1) { // ContactConstraintTwistPartSlim::solve, contact_constraint_element.rs:735-755
            float4 h3 = ROW1(ph3, CP_H3), h4 = ROW1(ph4, CP_H4), h5 = ROW1(ph5, CP_H5);
            Sym3 ii1 = {h3.x, h3.y, h3.z, h3.w, h4.x, h4.y}, ii2 = {h4.z, h4.w, h5.x, h5.y, h5.z, h5.w};
            V3 a = sym_mul(ii1, dir1), b = sym_mul(ii2, dir1);
            float dvel = dot(dir1, v1.ang - v2.ang) + 0.0f; // twist rhs is always zero
            float new_impulse = rp_clamp(hm0.x - h1.w * dvel, -twist_limit, twist_limit);
            float dl = new_impulse - hm0.x;
            hm0.x = new_impulse;
            v1.ang = v1.ang + a * dl;
            v2.ang = v2.ang - b * dl;
        }
        { // ContactConstraintTangentPartSlim::solve, contact_constraint_element.rs:650-705
            V3 td10 = v3(ROW1(q0, CP_T0)), td11 = v3(ROW1(q1, CP_T1)), td20 = v3(ROW1(q2, CP_T2)), td21 = v3(ROW1(q3, CP_T3));
            float dvel_0 = dot(t0, v1.lin) + dot(td10, v1.ang) - dot(t0, v2.lin) + dot(td20, v2.ang) + hm1.z;
            float dvel_1 = dot(t1, v1.lin) + dot(td11, v1.ang) - dot(t1, v2.lin) + dot(td21, v2.ang) + hm1.w;
            float k11 = h7.y, k22 = h7.z, k12 = h2.w * 0.5f;
            float inv_det = rp_inv(k11 * k22 - k12 * k12);
            float d0 = (k22 * dvel_0 - k12 * dvel_1) * inv_det;
            float d1 = (k11 * dvel_1 - k12 * dvel_0) * inv_det;
            float n0 = hm0.z - d0, n1 = hm0.w - d1;
            float l = sqrtf(n0 * n0 + n1 * n1);
            if (l > tangent_limit) { float sc = tangent_limit / l; n0 *= sc; n1 *= sc; }
            float dl0 = n0 - hm0.z, dl1 = n1 - hm0.w;
            hm0.z = n0; hm0.w = n1;
            v1.lin = v1.lin + cmul(t0 * dl0 + t1 * dl1, im1);
            v1.ang = v1.ang + (v3(ROW1(q4_, CP_T4)) * dl0 + v3(ROW1(q5, CP_T5)) * dl1);
            v2.lin = v2.lin + cmul(t0 * (-dl0) + t1 * (-dl1), im2);
            v2.ang = v2.ang + (v3(ROW1(q6, CP_T6)) * dl0 + v3(ROW1(q7, CP_T7)) * dl1);
        }
        A.st(CP_HM0, hm0);
        if (refresh) A.st(CP_HM1, hm1);
    }
    A.set_vel(id1, v1); A.set_vel(id2, v2);
}

// apply_restitution — contact_with_twist_friction.rs:568-597, contact_constraint_element.rs:508-534
template <class Acc>
RP_DEV void cons_restitution(const DevWorld &w, const Acc &A) {
    int id1 = A.id1(), id2 = A.id2(), n = A.n();
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (k >= n) break; any |= A.ld(NPL(k, NP_B)).w < 0.0f; }
    if (!any) return;
    V3 dir1 = v3(A.ld(CP_H0)), im1 = v3(A.ld(CP_H1)), im2 = v3(A.ld(CP_H2));
    Vel v1 = A.vel(id1), v2 = A.vel(id2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= n) break;
        float4 m = A.ld(NPL(k, NP_M));
        float4 a = A.ld(NPL(k, NP_A)), b = A.ld(NPL(k, NP_B)), c = A.ld(NPL(k, NP_C)), d = A.ld(NPL(k, NP_D));
        float seed = b.w;
        float dvel = dot(dir1, v1.lin) + dot(v3(a), v1.ang) - dot(dir1, v2.lin) + dot(v3(b), v2.ang) + seed;
        bool gate = seed < 0.0f && (m.w + m.z) > 0.0f;
        float new_impulse = gate ? rp_max(m.z - a.w * dvel, 0.0f) : m.z;
        float dl = new_impulse - m.z;
        m.z = new_impulse;
        A.st(NPL(k, NP_M), m);
        v1.lin = v1.lin + cmul(dir1, im1) * dl;
        v1.ang = v1.ang + v3(c) * dl;
        v2.lin = v2.lin + cmul(dir1, im2) * (-dl);
        v2.ang = v2.ang + v3(d) * dl;
    }
    A.set_vel(id1, v1); A.set_vel(id2, v2);
}

// S9: writeback_impulses — contact_with_twist_friction.rs:783-829
template <class Acc>
RP_DEV void cons_writeback(const DevWorld &w, const Acc &A, int s) {
    int n = A.n(), cids = A.cids();
    float4 h0 = A.ld(CP_H0), h6 = A.ld(CP_H6), hm0 = A.ld(CP_HM0);
    V3 dir1 = v3(h0), t0 = v3(h6), t1 = cross(dir1, t0);
    // stored impulses are canonicalised (signed zeros -> +0.0, writeback_impulses :783-805)
    V3 wtw = t0 * rp_canon0(hm0.z) + t1 * rp_canon0(hm0.w);
    wtw = v3(rp_canon0(wtw.x), rp_canon0(wtw.y), rp_canon0(wtw.z));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= n) break;
        int cid = (cids >> (8 * k)) & 0xff;
        float4 m = A.ld(NPL(k, NP_M));
        PT(w.pt_imp, cid, s) = make_float4(rp_canon0(m.w + m.z), rp_canon0(m.z), rp_canon0(hm0.x), 0.0f);
        PT(w.pt_wst, cid, s) = f4(wtw, 0.0f);
    }
}

enum { MODE_WARMSTART = 0, MODE_BIAS = 1, MODE_RELAX = 2, MODE_RESTITUTION = 3 };

template <class Acc>
RP_DEV void cons_apply(const DevWorld &w, const Acc &A, int mode, bool friction_in_bias, float solved_dt) {
    if (mode == MODE_WARMSTART) cons_update_warmstart(w, A, solved_dt);
    else if (mode == MODE_BIAS) cons_solve(w, A, false, friction_in_bias, solved_dt);
    else if (mode == MODE_RELAX) cons_solve(w, A, true, true, solved_dt);
    else cons_restitution(w, A);
}

// ---- per-body stages (shared by both paths) -----------------------------------------------------
// gyroscopic_corrected_angvel — dynamics/rigid_body.rs:2023-2046
RP_DEV V3 gyro_corrected(V3 angvel, Q4 axes, V3 pi, V3 inv_pi, float dt) {
    V3 wl = qrot_inv(axes, angvel);
    V3 curr = cmul(pi, wl);
    V3 eg = (-cross(wl, curr)) * dt;
    V3 total = curr + eg;
    float sq = len2(total);
    if (sq != 0.0f) { V3 capped = total * sqrtf(len2(curr) / sq); return qrot(axes, cmul(inv_pi, capped)); }
    return angvel;
}
// S0 for one dynamic body: forces + increments (solve.rs:234-291, worker.rs:46-104)
RP_DEV void body_begin(const DevWorld &w, int i, V3 &lin, V3 &ang, Q4 &rot, V3 &trans, V3 &incl, V3 &inca) {
    V3 im = v3(w.b_eim[i]);
    V3 mass = v3(rp_inv(im.x), rp_inv(im.y), rp_inv(im.z));
    float4 damp = w.b_damp[i];
    V3 g = v3(w.prm.gravity[0], w.prm.gravity[1], w.prm.gravity[2]);
    V3 force = v3(w.b_uforce[i]) + cmul(g, mass) * damp.z;
    V3 torque = v3(w.b_utorque[i]);
    Sym3 ii = load_ii(w, i);
    float dts = w.prm.dt_sub;
    inca = sym_mul(ii, torque) * dts;
    incl = cmul(force, im) * dts;
    lin = v3(w.b_linvel[i]); ang = v3(w.b_angvel[i]);
    rot = q4(w.b_rot[i]);
    trans = qrot(rot, v3(w.b_lcom_invm[i])) + v3(w.b_pos[i]);
}
// S2 — worker.rs:235-284
RP_DEV void body_increment(const DevWorld &w, int fl, V3 &lin, V3 &ang, Q4 rot, V3 incl, V3 inca, V3 inv_pi, Q4 pframe) {
    lin = lin + incl;
    ang = ang + inca;
    if (fl & RP_BF_GYRO) {
        V3 pi = v3(rp_inv(inv_pi.x), rp_inv(inv_pi.y), rp_inv(inv_pi.z));
        ang = gyro_corrected(ang, qmul(rot, pframe), pi, inv_pi, w.prm.dt_sub);
    }
}
// S6 — worker.rs:568-631, rigid_body_components.rs:884-898
RP_DEV void body_integrate(const DevWorld &w, int fl, V3 &lin, V3 &ang, Q4 &rot, V3 &trans) {
    if (w.prm.max_lin != 3.402823466e+38f) { float n = len(lin); if (n > w.prm.max_lin) lin = lin * (w.prm.max_lin / n); }
    if (!(fl & RP_BF_FASTROT)) { float n = len(ang); if (n > w.prm.max_ang) ang = ang * (w.prm.max_ang / n); }
    float dts = w.prm.dt_sub;
    V3 hang = ang * (dts * 0.5f);
    rot = qnormalize(qmul(q4(hang.x, hang.y, hang.z, 1.0f), rot));
    trans = trans + lin * dts;
}
// S10 + advance_to_final_positions — worker.rs:809-897, substep.rs:84-224, quarantine.rs:131
RP_DEV void body_writeback(const DevWorld &w, int i, V3 slin, V3 sang, Q4 rot, V3 com) {
    float4 damp = w.b_damp[i];
    float dt = w.prm.p.dt;
    V3 lin = slin * (1.0f / (1.0f + dt * damp.x));
    V3 ang = sang * (1.0f / (1.0f + dt * damp.y));
    V3 lcom = v3(w.b_lcom_invm[i]);
    V3 t = com + qrot(rot, -lcom);
    bool finite = isfinite(t.x) && isfinite(t.y) && isfinite(t.z) && isfinite(rot.x) && isfinite(rot.y) && isfinite(rot.z) && isfinite(rot.w) &&
                  isfinite(lin.x) && isfinite(lin.y) && isfinite(lin.z) && isfinite(ang.x) && isfinite(ang.y) && isfinite(ang.z);
    if (!finite) { // roll back to the last valid pose, stop the body
        atomicAdd(&w.flags[FL_QUARANTINE], 1);
        w.b_quar[i] = 1;
        w.b_linvel[i] = make_float4(0, 0, 0, 0); w.b_angvel[i] = make_float4(0, 0, 0, 0);
        w.b_uforce[i] = make_float4(0, 0, 0, 0); w.b_utorque[i] = make_float4(0, 0, 0, 0); // sanitize_body_dynamics (quarantine.rs:56-63)
        return;
    }
    const float4 invpi_ext = w.b_invpi[i]; // xyz: inverse principal inertia, w: max_extent of the attached shapes
    if (w.prm.p.max_ccd_substeps != 0 && damp.w < 3.0e38f) {
        // CCD activation (worker.rs:845-865, RigidBodyCcd::is_moving_fast_with_next_position, rigid_body_components.rs:1131-1157): the
        // farthest point of the body moved more than half its thinnest extent this step.  damp.w = ccd_thickness (min over the
        // attached shapes).
        const float max_extent = invpi_ext.w;
        V3 dcom = com - v3(w.b_wcom[i]);
        Q4 dq = qmul(rot, qconj(q4(w.b_rot[i])));
        V3 dv = v3(dq.x, dq.y, dq.z);
        // cheap bound first (no square root, no atan): motion <= |dcom| + pi |dq.v| max_extent; a body at rest stops here
        const float quarter = 0.25f * damp.w, q2 = quarter * quarter;
        if (dot(dcom, dcom) > q2 || dot(dv, dv) * (9.8696044f * max_extent * max_extent) > q2) {
            float inv_dt = dt == 0.0f ? 0.0f : 1.0f / dt;
            float max_delta = len(dcom) + 2.0f * len(dv) * max_extent;
            float max_vel = len(dcom * inv_dt) + len(quat_to_scaled_axis(dq) * inv_dt) * max_extent; // ccd_vels = interpolate_velocity(inv_dt)
            float max_motion = rp_max(max_delta, max_vel * dt);
            if (max_motion > 0.5f * damp.w) { // the continuous-collision pass (k_ccd, full steps) sweeps this body from its start-of-step pose
                atomicAdd(&w.flags[FL_CCD_ACTIVE], 1);
                w.b_ccd0_pos[i] = w.b_pos[i]; w.b_ccd0_rot[i] = w.b_rot[i];
                const int k = atomicAdd(&w.flags[FL_CCD_N], 1);
                if (k < w.n_bodies) w.ccd_list[k] = i;
            }
        }
    }
    w.b_linvel[i] = f4(lin, 0.0f); w.b_angvel[i] = f4(ang, 0.0f);
    w.b_pos[i] = f4(t, 0.0f); w.b_rot[i] = f4(rot);
    w.b_wcom[i] = f4(qrot(rot, lcom) + t, 0.0f);
    Sym3 ii = world_inv_inertia(v3(invpi_ext), q4(w.b_pframe[i]), rot);
    apply_locked_rotations((w.b_flags[i] >> RP_BF_LOCK_SHIFT) & 0x3f, ii);
    w.b_eii0[i] = make_float4(ii.m11, ii.m12, ii.m13, ii.m22);
    w.b_eii1[i] = make_float4(ii.m23, ii.m33, 0.0f, 0.0f);
}
