// rp_solver.hip — the colour-ordered TGS-soft contact velocity solver + integrator on device.
//
// Restates StagedIslandSolver::init_and_solve / run_worker
// (/root/reference/src/dynamics/solver/staged_island_solver/{init.rs:30-545, worker.rs:32-898,
// solve.rs:12-209}) with ContactWithTwistFriction kernels
// (/root/reference/src/dynamics/solver/contact_constraint/contact_with_twist_friction.rs and
// contact_constraint_element.rs).  The reference's 4-lane AoSoA chunks + worker stage machine become:
//   * one thread per solver manifold, constraint data in float4 planes C[plane][position] so a
//     wavefront's 64 consecutive positions load 1 KiB per plane (coalesced, HBM/L2 streaming);
//   * one launch per (sweep, colour stage) for colours with >= 32 chunks ("parallel" colours,
//     init.rs:169): same-colour manifolds touch disjoint dynamic bodies, so the body
//     gather/scatter needs no atomics;
//   * one single-workgroup "tail" launch per sweep for the small colours (ascending) and the
//     overflow colour (serial, lane 0) — the reference runs exactly those on worker 0
//     (init.rs:192-254).  The tail also absorbs any parallel stage the host did not launch, so the
//     Gauss-Seidel order never depends on what the host knows about the layout.
// No FMA contraction (-ffp-contract=off), IEEE divide/sqrt: same arithmetic as the reference's
// scalar lanes.
#include "rp_world.h"

#define CP(plane, pos) w.C[(size_t)(plane) * w.cons_cap + (pos)]
#define CN(k, sub, pos) w.C[(size_t)(CP_N0 + 7 * (k) + (sub)) * w.cons_cap + (pos)]
#define PT(plane, k, s) plane[(size_t)(k) * w.pool_cap + (s)]

struct Vel { V3 lin, ang; };
RP_DEV Vel load_vel(const DevWorld &w, int id) {
    Vel v;
    if (id < 0) { v.lin = v3(0, 0, 0); v.ang = v3(0, 0, 0); }
    else { v.lin = v3(w.s_lin[id]); v.ang = v3(w.s_ang[id]); }
    return v;
}
RP_DEV void store_vel(const DevWorld &w, int id, const Vel &v) {
    if (id >= 0) { w.s_lin[id] = f4(v.lin, 0.0f); w.s_ang[id] = f4(v.ang, 0.0f); }
}
struct Xf { Q4 r; V3 t; };
RP_DEV Xf load_xf(const DevWorld &w, int id) {
    Xf x;
    if (id < 0) { x.r = q4(0, 0, 0, 1); x.t = v3(0, 0, 0); }
    else { x.r = q4(w.s_rot[id]); x.t = v3(w.s_trans[id]); }
    return x;
}
RP_DEV V3 xf_tp(const Xf &x, V3 p) { return qrot(x.r, p) + x.t; }
RP_DEV V3 xf_itp(const Xf &x, V3 p) { return qrot_inv(x.r, p - x.t); }
RP_DEV Sym3 load_ii(const DevWorld &w, int id) {
    Sym3 m = {0, 0, 0, 0, 0, 0};
    if (id >= 0) { float4 a = w.b_eii0[id], b = w.b_eii1[id]; m.m11 = a.x; m.m12 = a.y; m.m13 = a.z; m.m22 = a.w; m.m23 = b.x; m.m33 = b.y; }
    return m;
}

// S0: RigidBody -> solver body + per-substep increments (worker.rs:46-104, solver_body.rs:82-121)
// fused with the force pass (solve.rs:234-291, rigid_body_components.rs:1030-1033).
__global__ void k_solver_begin(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) w.flags[FL_ANY_BOUNCY] = 0;
    if (i >= w.n_bodies) return;
    int fl = w.b_flags[i];
    if ((fl & RP_BF_TYPE_MASK) != RP_BODY_DYNAMIC) return;
    V3 im = v3(w.b_eim[i]);
    V3 mass = v3(rp_inv(im.x), rp_inv(im.y), rp_inv(im.z));
    float4 damp = w.b_damp[i];
    V3 g = v3(w.prm.gravity[0], w.prm.gravity[1], w.prm.gravity[2]);
    V3 force = v3(w.b_uforce[i]) + cmul(g, mass) * damp.z;
    V3 torque = v3(w.b_utorque[i]);
    Sym3 ii = load_ii(w, i);
    float dts = w.prm.dt_sub;
    w.s_inca[i] = f4(sym_mul(ii, torque) * dts, 0.0f);
    w.s_incl[i] = f4(cmul(force, im) * dts, 0.0f);
    w.s_lin[i] = w.b_linvel[i];
    w.s_ang[i] = w.b_angvel[i];
    Q4 rot = q4(w.b_rot[i]);
    w.s_rot[i] = f4(rot);
    w.s_trans[i] = f4(qrot(rot, v3(w.b_lcom_invm[i])) + v3(w.b_pos[i]), 0.0f);
}

// S1: ContactWithTwistFrictionBuilder::generate — contact_with_twist_friction.rs:58-424 (one lane)
__global__ void k_generate(DevWorld w) {
    int M = w.flags[FL_N_CONS];
    if (M > w.cons_cap) M = w.cons_cap;
    int stride = gridDim.x * blockDim.x;
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < M; pos += stride) {
        int s = w.cons_pair[pos];
        int c1 = w.p_c1[s], c2 = w.p_c2[s];
        int rb1 = w.c_parent[c1], rb2 = w.c_parent[c2];
        int rel_dom = w.p_reldom[s];
        bool dyn1 = rb1 >= 0 && (w.b_flags[rb1] & RP_BF_TYPE_MASK) == RP_BODY_DYNAMIC;
        bool dyn2 = rb2 >= 0 && (w.b_flags[rb2] & RP_BF_TYPE_MASK) == RP_BODY_DYNAMIC;
        int id1 = (dyn1 && rel_dom <= 0) ? rb1 : -1;
        int id2 = (dyn2 && rel_dom >= 0) ? rb2 : -1;
        Vel vels1 = load_vel(w, id1), vels2 = load_vel(w, id2);
        Xf poses1 = load_xf(w, id1), poses2 = load_xf(w, id2);
        V3 im1 = id1 >= 0 ? v3(w.b_eim[id1]) : v3(0, 0, 0), im2 = id2 >= 0 ? v3(w.b_eim[id2]) : v3(0, 0, 0);
        Sym3 ii1 = load_ii(w, id1), ii2 = load_ii(w, id2);
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
        V3 points[4];
        int cids = 0;
        bool bouncy_seed = false;
        V3 imsum = im1 + im2;
        for (int k = 0; k < count; ++k) {
            float weight = inv_num_points;
            float4 a1 = PT(w.sc_a1, k, s), a2 = PT(w.sc_a2, k, s);
            int cid = __float_as_int(a2.w);
            cids |= (cid & 0xff) << (8 * k);
            float4 pimp = PT(w.pt_imp, cid, s);
            V3 wt = v3(PT(w.pt_wst, cid, s));
            float warmstart_impulse = pimp.y;
            float wti0 = dot(wt, t0), wti1 = dot(wt, t1);
            float warmstart_twist_impulse = pimp.z;
            bool is_new = pimp.x == 0.0f;
            float is_bouncy = is_new ? (restitution > 0.0f ? 1.0f : 0.0f) : (restitution >= 1.0f ? 1.0f : 0.0f);
            V3 p1 = xf_tp(poses1, v3(a1));
            V3 p2 = xf_tp(poses2, v3(a2));
            float dist = dot(p1 - p2, force_dir1);
            V3 dp1 = v3(PT(w.pt_dp1, cid, s)), dp2 = v3(PT(w.pt_dp2, cid, s));
            V3 point = world_com1 + dp1;
            points[k] = point;
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
            CN(k, NP_M, pos) = make_float4(0.0f, 1.0f, warmstart_impulse, -warmstart_impulse);
            CN(k, NP_A, pos) = f4(torque_dir1, projected_mass);
            CN(k, NP_B, pos) = f4(torque_dir2, restitution_seed);
            CN(k, NP_C, pos) = f4(ii_torque_dir1, info_dist);
            CN(k, NP_D, pos) = f4(ii_torque_dir2, 0.0f);
            CN(k, NP_E, pos) = f4(xf_itp(poses1, point), 0.0f);
            CN(k, NP_F, pos) = f4(xf_itp(poses2, world_com2 + dp2), 0.0f);
        }
        if (bouncy_seed) w.flags[FL_ANY_BOUNCY] = 1;
        float twist_imp = count > 1 ? twist_warmstart : 0.0f;
        V3 dp1 = friction_center - world_com1, dp2 = friction_center2 - world_com2;
        float twist_r = 0.0f;
        float4 tdists = make_float4(0, 0, 0, 0);
        if (count > 1) {
            float td[4] = {0, 0, 0, 0};
            for (int k = 0; k < count; ++k) td[k] = len(friction_center - points[k]);
            tdists = make_float4(td[0], td[1], td[2], td[3]);
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
        CP(CP_H0, pos) = f4(force_dir1, friction);
        CP(CP_H1, pos) = f4(im1, twist_r);
        CP(CP_H2, pos) = f4(im2, r[2]);
        CP(CP_H3, pos) = make_float4(ii1.m11, ii1.m12, ii1.m13, ii1.m22);
        CP(CP_H4, pos) = make_float4(ii1.m23, ii1.m33, ii2.m11, ii2.m12);
        CP(CP_H5, pos) = make_float4(ii2.m13, ii2.m22, ii2.m23, ii2.m33);
        CP(CP_H6, pos) = f4(t0, rhs_wo[0]);
        CP(CP_H7, pos) = make_float4(rhs_wo[1], r[0], r[1], 0.0f);
        CP(CP_H8, pos) = tdists;
        CP(CP_HM0, pos) = make_float4(twist_imp, -twist_imp, tw0, tw1);
        CP(CP_HM1, pos) = make_float4(-tw0, -tw1, rhs_wo[0], rhs_wo[1]);
        CP(CP_T0, pos) = f4(td1[0], 0.0f); CP(CP_T1, pos) = f4(td1[1], 0.0f);
        CP(CP_T2, pos) = f4(td2[0], 0.0f); CP(CP_T3, pos) = f4(td2[1], 0.0f);
        CP(CP_T4, pos) = f4(itd1[0], 0.0f); CP(CP_T5, pos) = f4(itd1[1], 0.0f);
        CP(CP_T6, pos) = f4(itd2[0], 0.0f); CP(CP_T7, pos) = f4(itd2[1], 0.0f);
        CP(CP_B0, pos) = f4(xf_itp(poses1, friction_center), 0.0f);
        CP(CP_B1, pos) = f4(xf_itp(poses2, friction_center2), 0.0f);
        CP(CP_B2, pos) = f4(tangent_vel, 0.0f);
        w.k_b1[pos] = id1; w.k_b2[pos] = id2; w.k_n[pos] = count; w.k_cid[pos] = cids;
    }
}

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
// S2: increments + gyroscopic correction — worker.rs:235-284
__global__ void k_increment(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w.n_bodies) return;
    int fl = w.b_flags[i];
    if ((fl & RP_BF_TYPE_MASK) != RP_BODY_DYNAMIC) return;
    V3 lin = v3(w.s_lin[i]) + v3(w.s_incl[i]);
    V3 ang = v3(w.s_ang[i]) + v3(w.s_inca[i]);
    if (fl & RP_BF_GYRO) {
        V3 inv_pi = v3(w.b_invpi[i]);
        V3 pi = v3(rp_inv(inv_pi.x), rp_inv(inv_pi.y), rp_inv(inv_pi.z));
        Q4 axes = qmul(q4(w.s_rot[i]), q4(w.b_pframe[i]));
        ang = gyro_corrected(ang, axes, pi, inv_pi, w.prm.dt_sub);
    }
    w.s_lin[i] = f4(lin, 0.0f);
    w.s_ang[i] = f4(ang, 0.0f);
}
// S6: speed caps + integrate_linearized — worker.rs:568-631, rigid_body_components.rs:884-898
__global__ void k_integrate(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w.n_bodies) return;
    int fl = w.b_flags[i];
    if ((fl & RP_BF_TYPE_MASK) != RP_BODY_DYNAMIC) return;
    V3 lin = v3(w.s_lin[i]), ang = v3(w.s_ang[i]);
    bool changed = false;
    if (w.prm.max_lin != 3.402823466e+38f) { float n = len(lin); if (n > w.prm.max_lin) { lin = lin * (w.prm.max_lin / n); changed = true; } }
    if (!(fl & RP_BF_FASTROT)) { float n = len(ang); if (n > w.prm.max_ang) { ang = ang * (w.prm.max_ang / n); changed = true; } }
    if (changed) { w.s_lin[i] = f4(lin, 0.0f); w.s_ang[i] = f4(ang, 0.0f); }
    float dts = w.prm.dt_sub;
    V3 hang = ang * (dts * 0.5f);
    Q4 q = qmul(q4(hang.x, hang.y, hang.z, 1.0f), q4(w.s_rot[i]));
    w.s_rot[i] = f4(qnormalize(q));
    w.s_trans[i] = f4(v3(w.s_trans[i]) + lin * dts, 0.0f);
}

// ---- per-manifold sweeps -------------------------------------------------------------------
// update (+ warmstart): contact_with_twist_friction.rs:426-522 and :633-678
RP_DEV void cons_update_warmstart(const DevWorld &w, int pos, float solved_dt) {
    int id1 = w.k_b1[pos], id2 = w.k_b2[pos], n = w.k_n[pos];
    bool is_static = id1 < 0 || id2 < 0;
    float fstatic = is_static ? 1.0f : 0.0f;
    float cfm_factor = w.prm.dyn_cfm + fstatic * (w.prm.static_cfm - w.prm.dyn_cfm);
    float erp_inv_dt = w.prm.dyn_erp_inv_dt + fstatic * (w.prm.static_erp_inv_dt - w.prm.dyn_erp_inv_dt);
    float inv_dt = w.prm.inv_dt_sub;
    float maxcv = w.prm.max_corrective_velocity;
    float wc = w.prm.p.warmstart_coefficient;
    Xf x1 = load_xf(w, id1), x2 = load_xf(w, id2);
    float4 h0 = CP(CP_H0, pos), h6 = CP(CP_H6, pos);
    V3 dir1 = v3(h0), t0 = v3(h6), t1 = cross(dir1, t0);
    V3 tangent_delta = v3(CP(CP_B2, pos)) * solved_dt;
    V3 im1 = v3(CP(CP_H1, pos)), im2 = v3(CP(CP_H2, pos));
    Vel v1 = load_vel(w, id1), v2 = load_vel(w, id2);
    bool ws = wc != 0.0f;
    for (int k = 0; k < n; ++k) {
        float4 m = CN(k, NP_M, pos);
        float4 c = CN(k, NP_C, pos), d = CN(k, NP_D, pos);
        V3 p1 = xf_tp(x1, v3(CN(k, NP_E, pos))) + tangent_delta;
        V3 p2 = xf_tp(x2, v3(CN(k, NP_F, pos)));
        float dist = c.w + dot(p1 - p2, dir1);
        float rhs_wo_bias = rp_max(dist, 0.0f) * inv_dt;
        float rhs_bias = rp_clamp(dist * erp_inv_dt, -maxcv, 0.0f);
        m.x = rhs_wo_bias + rhs_bias;
        m.y = dist <= 0.0f ? cfm_factor : 1.0f;
        m.w += m.z;
        m.z *= wc;
        CN(k, NP_M, pos) = m;
        if (ws) { // ContactConstraintNormalPartSlim::warmstart, contact_constraint_element.rs:465-478
            v1.lin = v1.lin + cmul(dir1, im1) * m.z;
            v1.ang = v1.ang + v3(c) * m.z;
            v2.lin = v2.lin + cmul(dir1, im2) * (-m.z);
            v2.ang = v2.ang + v3(d) * m.z;
        }
    }
    float4 hm0 = CP(CP_HM0, pos), hm1 = CP(CP_HM1, pos), h7 = CP(CP_H7, pos);
    {
        V3 p1 = xf_tp(x1, v3(CP(CP_B0, pos))) + tangent_delta;
        V3 p2 = xf_tp(x2, v3(CP(CP_B1, pos)));
        float bias0 = dot(p1 - p2, t0) * inv_dt, bias1 = dot(p1 - p2, t1) * inv_dt;
        hm1.z = h6.w + bias0; hm1.w = h7.x + bias1;
        hm1.x += hm0.z; hm1.y += hm0.w;
        hm0.z *= wc; hm0.w *= wc;
        hm0.y += hm0.x;
        hm0.x *= wc;
    }
    CP(CP_HM0, pos) = hm0; CP(CP_HM1, pos) = hm1;
    if (ws) {
        float i0 = hm0.z, i1 = hm0.w;
        v1.lin = v1.lin + cmul(t0 * i0 + t1 * i1, im1);
        v1.ang = v1.ang + (v3(CP(CP_T4, pos)) * i0 + v3(CP(CP_T5, pos)) * i1);
        v2.lin = v2.lin + cmul(t0 * (-i0) + t1 * (-i1), im2);
        v2.ang = v2.ang + (v3(CP(CP_T6, pos)) * i0 + v3(CP(CP_T7, pos)) * i1);
        if (n > 1) {
            float4 h3 = CP(CP_H3, pos), h4 = CP(CP_H4, pos), h5 = CP(CP_H5, pos);
            Sym3 ii1 = {h3.x, h3.y, h3.z, h3.w, h4.x, h4.y}, ii2 = {h4.z, h4.w, h5.x, h5.y, h5.z, h5.w};
            v1.ang = v1.ang + sym_mul(ii1, dir1) * hm0.x;
            v2.ang = v2.ang - sym_mul(ii2, dir1) * hm0.x;
        }
        store_vel(w, id1, v1); store_vel(w, id2, v2);
    }
}

// solve: contact_with_twist_friction.rs:680-781 (+ refresh_rhs_wo_bias :529-554 when `refresh`)
RP_DEV void cons_solve(const DevWorld &w, int pos, bool refresh, bool friction, float solved_dt) {
    int id1 = w.k_b1[pos], id2 = w.k_b2[pos], n = w.k_n[pos];
    float4 h0 = CP(CP_H0, pos);
    V3 dir1 = v3(h0);
    float4 h1 = CP(CP_H1, pos), h2 = CP(CP_H2, pos);
    V3 im1 = v3(h1), im2 = v3(h2);
    Vel v1 = load_vel(w, id1), v2 = load_vel(w, id2);
    Xf x1, x2; V3 tangent_delta = v3(0, 0, 0);
    if (refresh) { x1 = load_xf(w, id1); x2 = load_xf(w, id2); tangent_delta = v3(CP(CP_B2, pos)) * solved_dt; }
    float imp[4] = {0, 0, 0, 0};
    for (int k = 0; k < n; ++k) {
        float4 m = CN(k, NP_M, pos);
        float4 a = CN(k, NP_A, pos), b = CN(k, NP_B, pos), c = CN(k, NP_C, pos), d = CN(k, NP_D, pos);
        if (refresh) {
            V3 p1 = xf_tp(x1, v3(CN(k, NP_E, pos))) + tangent_delta;
            V3 p2 = xf_tp(x2, v3(CN(k, NP_F, pos)));
            float dist = c.w + dot(p1 - p2, dir1);
            m.x = rp_max(dist, 0.0f) * w.prm.inv_dt_sub;
            m.y = 1.0f;
        }
        float dvel = dot(dir1, v1.lin) + dot(v3(a), v1.ang) - dot(dir1, v2.lin) + dot(v3(b), v2.ang) + m.x;
        float new_impulse = m.y * rp_max(m.z - a.w * dvel, 0.0f);
        float dl = new_impulse - m.z;
        m.z = new_impulse;
        imp[k] = new_impulse;
        CN(k, NP_M, pos) = m;
        v1.lin = v1.lin + cmul(dir1, im1) * dl;
        v1.ang = v1.ang + v3(c) * dl;
        v2.lin = v2.lin + cmul(dir1, im2) * (-dl);
        v2.ang = v2.ang + v3(d) * dl;
    }
    if (friction) {
        float4 h6 = CP(CP_H6, pos), h7 = CP(CP_H7, pos), h8 = CP(CP_H8, pos);
        float4 hm0 = CP(CP_HM0, pos), hm1 = CP(CP_HM1, pos);
        if (refresh) { hm1.z = h6.w; hm1.w = h7.x; }
        V3 t0 = v3(h6), t1 = cross(dir1, t0);
        float tdist[4] = {h8.x, h8.y, h8.z, h8.w};
        float tangent_limit = 0.0f, twist_limit = 0.0f;
        for (int k = 0; k < n; ++k) { tangent_limit += imp[k]; twist_limit += imp[k] * tdist[k]; }
        tangent_limit *= h0.w; twist_limit *= h0.w;
        if (n > 1) { // ContactConstraintTwistPartSlim::solve, contact_constraint_element.rs:735-755
            float4 h3 = CP(CP_H3, pos), h4 = CP(CP_H4, pos), h5 = CP(CP_H5, pos);
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
            V3 td10 = v3(CP(CP_T0, pos)), td11 = v3(CP(CP_T1, pos)), td20 = v3(CP(CP_T2, pos)), td21 = v3(CP(CP_T3, pos));
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
            v1.ang = v1.ang + (v3(CP(CP_T4, pos)) * dl0 + v3(CP(CP_T5, pos)) * dl1);
            v2.lin = v2.lin + cmul(t0 * (-dl0) + t1 * (-dl1), im2);
            v2.ang = v2.ang + (v3(CP(CP_T6, pos)) * dl0 + v3(CP(CP_T7, pos)) * dl1);
        }
        CP(CP_HM0, pos) = hm0;
        if (refresh) CP(CP_HM1, pos) = hm1;
    }
    store_vel(w, id1, v1); store_vel(w, id2, v2);
}

// apply_restitution — contact_with_twist_friction.rs:568-597, contact_constraint_element.rs:508-534
RP_DEV void cons_restitution(const DevWorld &w, int pos) {
    int id1 = w.k_b1[pos], id2 = w.k_b2[pos], n = w.k_n[pos];
    bool any = false;
    for (int k = 0; k < n; ++k) any |= CN(k, NP_B, pos).w < 0.0f;
    if (!any) return;
    V3 dir1 = v3(CP(CP_H0, pos)), im1 = v3(CP(CP_H1, pos)), im2 = v3(CP(CP_H2, pos));
    Vel v1 = load_vel(w, id1), v2 = load_vel(w, id2);
    for (int k = 0; k < n; ++k) {
        float4 m = CN(k, NP_M, pos);
        float4 a = CN(k, NP_A, pos), b = CN(k, NP_B, pos), c = CN(k, NP_C, pos), d = CN(k, NP_D, pos);
        float seed = b.w;
        float dvel = dot(dir1, v1.lin) + dot(v3(a), v1.ang) - dot(dir1, v2.lin) + dot(v3(b), v2.ang) + seed;
        bool gate = seed < 0.0f && (m.w + m.z) > 0.0f;
        float new_impulse = gate ? rp_max(m.z - a.w * dvel, 0.0f) : m.z;
        float dl = new_impulse - m.z;
        m.z = new_impulse;
        CN(k, NP_M, pos) = m;
        v1.lin = v1.lin + cmul(dir1, im1) * dl;
        v1.ang = v1.ang + v3(c) * dl;
        v2.lin = v2.lin + cmul(dir1, im2) * (-dl);
        v2.ang = v2.ang + v3(d) * dl;
    }
    store_vel(w, id1, v1); store_vel(w, id2, v2);
}

enum { MODE_WARMSTART = 0, MODE_BIAS = 1, MODE_RELAX = 2, MODE_RESTITUTION = 3 };

RP_DEV void cons_apply(const DevWorld &w, int pos, int mode, bool friction_in_bias, float solved_dt) {
    if (mode == MODE_WARMSTART) cons_update_warmstart(w, pos, solved_dt);
    else if (mode == MODE_BIAS) cons_solve(w, pos, false, friction_in_bias, solved_dt);
    else if (mode == MODE_RELAX) cons_solve(w, pos, true, true, solved_dt);
    else cons_restitution(w, pos);
}

// One parallel colour stage: the reference's claim/steal chunk loop becomes a grid-stride loop.
template <int MODE>
__global__ void __launch_bounds__(256) k_stage(DevWorld w, int stage, int friction_in_bias, float solved_dt) {
    if (stage >= w.flags[FL_N_PARALLEL]) return;
    if (MODE == MODE_RESTITUTION && !w.flags[FL_ANY_BOUNCY]) return;
    int beg = w.stage_begin[stage], cnt = w.stage_count[stage];
    int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) cons_apply(w, beg + i, MODE, friction_in_bias != 0, solved_dt);
}

// Serial tail (worker 0 of the reference): stages [first, n_stages) one after the other inside one
// workgroup, then the overflow colour on lane 0.
template <int MODE>
__global__ void __launch_bounds__(1024) k_tail(DevWorld w, int first, int friction_in_bias, float solved_dt) {
    if (MODE == MODE_RESTITUTION && !w.flags[FL_ANY_BOUNCY]) return;
    int nst = w.flags[FL_N_STAGES], npar = w.flags[FL_N_PARALLEL];
    int start = first < npar ? first : npar;
    for (int st = start; st < nst; ++st) {
        int beg = w.stage_begin[st], cnt = w.stage_count[st];
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) cons_apply(w, beg + i, MODE, friction_in_bias != 0, solved_dt);
        __threadfence();
        __syncthreads();
    }
    if (w.flags[FL_HAS_OVERFLOW_COLOR] && threadIdx.x == 0) {
        int beg = w.stage_begin[nst], cnt = w.stage_count[nst];
        for (int i = 0; i < cnt; ++i) { cons_apply(w, beg + i, MODE, friction_in_bias != 0, solved_dt); __threadfence(); }
    }
}

// S9: writeback_impulses — contact_with_twist_friction.rs:783-829
__global__ void k_writeback_impulses(DevWorld w) {
    int M = w.flags[FL_N_CONS];
    if (M > w.cons_cap) M = w.cons_cap;
    int stride = gridDim.x * blockDim.x;
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < M; pos += stride) {
        int s = w.cons_pair[pos];
        int n = w.k_n[pos], cids = w.k_cid[pos];
        float4 h0 = CP(CP_H0, pos), h6 = CP(CP_H6, pos), hm0 = CP(CP_HM0, pos);
        V3 dir1 = v3(h0), t0 = v3(h6), t1 = cross(dir1, t0);
        V3 wtw = t0 * hm0.z + t1 * hm0.w;
        for (int k = 0; k < n; ++k) {
            int cid = (cids >> (8 * k)) & 0xff;
            float4 m = CN(k, NP_M, pos);
            PT(w.pt_imp, cid, s) = make_float4(m.w + m.z, m.z, hm0.x, 0.0f);
            PT(w.pt_wst, cid, s) = f4(wtw, 0.0f);
        }
    }
}

// S10 + advance_to_final_positions: velocities (+damping) and poses back to the bodies, world mass
// properties refreshed from the new rotation — worker.rs:809-897, substep.rs:84-224,
// rigid_body_components.rs:528-578,835-841.  Non-finite poses are quarantined (quarantine.rs:131).
__global__ void k_writeback_bodies(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) w.flags[FL_STEP] += 1;
    if (i >= w.n_bodies) return;
    int fl = w.b_flags[i];
    if ((fl & RP_BF_TYPE_MASK) != RP_BODY_DYNAMIC) return;
    float4 damp = w.b_damp[i];
    float dt = w.prm.p.dt;
    V3 lin = v3(w.s_lin[i]) * (1.0f / (1.0f + dt * damp.x));
    V3 ang = v3(w.s_ang[i]) * (1.0f / (1.0f + dt * damp.y));
    Q4 rot = q4(w.s_rot[i]);
    V3 com = v3(w.s_trans[i]);
    V3 lcom = v3(w.b_lcom_invm[i]);
    V3 t = com + qrot(rot, -lcom);
    bool finite = isfinite(t.x) && isfinite(t.y) && isfinite(t.z) && isfinite(rot.x) && isfinite(rot.y) && isfinite(rot.z) && isfinite(rot.w) &&
                  isfinite(lin.x) && isfinite(lin.y) && isfinite(lin.z) && isfinite(ang.x) && isfinite(ang.y) && isfinite(ang.z);
    if (!finite) { // roll back to the last valid pose, stop the body
        atomicAdd(&w.flags[FL_QUARANTINE], 1);
        w.b_linvel[i] = make_float4(0, 0, 0, 0); w.b_angvel[i] = make_float4(0, 0, 0, 0);
        return;
    }
    w.b_linvel[i] = f4(lin, 0.0f); w.b_angvel[i] = f4(ang, 0.0f);
    w.b_pos[i] = f4(t, 0.0f); w.b_rot[i] = f4(rot);
    w.b_wcom[i] = f4(qrot(rot, lcom) + t, 0.0f);
    Sym3 ii = world_inv_inertia(v3(w.b_invpi[i]), q4(w.b_pframe[i]), rot);
    w.b_eii0[i] = make_float4(ii.m11, ii.m12, ii.m13, ii.m22);
    w.b_eii1[i] = make_float4(ii.m23, ii.m33, 0.0f, 0.0f);
}

// World mass properties at insertion time (RigidBodyMassProps::update_world_mass_properties).
__global__ void k_init_bodies(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w.n_bodies) return;
    int fl = w.b_flags[i];
    Q4 rot = q4(w.b_rot[i]);
    V3 t = v3(w.b_pos[i]);
    float4 li = w.b_lcom_invm[i];
    w.b_wcom[i] = f4(qrot(rot, v3(li)) + t, 0.0f);
    if ((fl & RP_BF_TYPE_MASK) == RP_BODY_DYNAMIC) {
        w.b_eim[i] = make_float4(li.w, li.w, li.w, 0.0f);
        Sym3 ii = world_inv_inertia(v3(w.b_invpi[i]), q4(w.b_pframe[i]), rot);
        w.b_eii0[i] = make_float4(ii.m11, ii.m12, ii.m13, ii.m22);
        w.b_eii1[i] = make_float4(ii.m23, ii.m33, 0.0f, 0.0f);
    } else {
        w.b_eim[i] = make_float4(0, 0, 0, 0); w.b_eii0[i] = make_float4(0, 0, 0, 0); w.b_eii1[i] = make_float4(0, 0, 0, 0);
    }
}

// ---- host-side launch sequence ---------------------------------------------------------------
struct SolverLaunchPlan { int parallel_stages; int stage_blocks; int has_restitution; };

template <int MODE>
static void launch_sweep(const DevWorld &w, hipStream_t st, const SolverLaunchPlan &plan, int fib, float solved_dt) {
    for (int s = 0; s < plan.parallel_stages; ++s)
        hipLaunchKernelGGL(k_stage<MODE>, dim3(plan.stage_blocks), dim3(256), 0, st, w, s, fib, solved_dt);
    hipLaunchKernelGGL(k_tail<MODE>, dim3(1), dim3(1024), 0, st, w, plan.parallel_stages, fib, solved_dt);
}

void rp_launch_init_bodies(const DevWorld &w, hipStream_t st) {
    if (w.n_bodies == 0) return;
    hipLaunchKernelGGL(k_init_bodies, dim3((w.n_bodies + 255) / 256), dim3(256), 0, st, w);
}

void rp_launch_solver_assembly(const DevWorld &w, hipStream_t st) {
    int nb = (w.n_bodies + 255) / 256; if (nb < 1) nb = 1;
    int cb = (w.cons_cap + 255) / 256; if (cb > 2048) cb = 2048; if (cb < 1) cb = 1;
    hipLaunchKernelGGL(k_solver_begin, dim3(nb), dim3(256), 0, st, w);
    hipLaunchKernelGGL(k_generate, dim3(cb), dim3(256), 0, st, w);
}

// The TGS loop proper: S2..S7 for every substep (+ S8 restitution) — worker.rs:207-734.
void rp_launch_solver_loop(const DevWorld &w, hipStream_t st, int parallel_stages, int stage_blocks, int has_restitution) {
    SolverLaunchPlan plan = {parallel_stages, stage_blocks < 1 ? 1 : stage_blocks, has_restitution};
    int nb = (w.n_bodies + 255) / 256; if (nb < 1) nb = 1;
    const rp_integration_params &p = w.prm.p;
    int fib = (p.friction_in_bias_pass || p.num_internal_stabilization_iterations == 0) ? 1 : 0;
    for (int s = 0; s < w.prm.num_substeps; ++s) {
        float solved_dt = (float)s * w.prm.dt_sub;
        hipLaunchKernelGGL(k_increment, dim3(nb), dim3(256), 0, st, w);
        launch_sweep<MODE_WARMSTART>(w, st, plan, fib, solved_dt);
        for (int it = 0; it < p.num_internal_pgs_iterations; ++it) launch_sweep<MODE_BIAS>(w, st, plan, fib, solved_dt);
        hipLaunchKernelGGL(k_integrate, dim3(nb), dim3(256), 0, st, w);
        for (int it = 0; it < p.num_internal_stabilization_iterations; ++it) launch_sweep<MODE_RELAX>(w, st, plan, fib, solved_dt + w.prm.dt_sub);
    }
    if (has_restitution) launch_sweep<MODE_RESTITUTION>(w, st, plan, fib, 0.0f);
}

void rp_launch_solver_writeback(const DevWorld &w, hipStream_t st) {
    int nb = (w.n_bodies + 255) / 256; if (nb < 1) nb = 1;
    int cb = (w.cons_cap + 255) / 256; if (cb > 2048) cb = 2048; if (cb < 1) cb = 1;
    hipLaunchKernelGGL(k_writeback_impulses, dim3(cb), dim3(256), 0, st, w);
    hipLaunchKernelGGL(k_writeback_bodies, dim3(nb), dim3(256), 0, st, w);
}
