// rp_islands_lean.h — the register-lean form of the per-island megakernel: TWO islands per CU.
//
// k_island_solve (rp_islands.hip) keeps a manifold's whole constraint in the VGPRs of a lane pair and needs 256 of them, so one
// 512-thread workgroup fills a CU's register file and a world with more islands than CUs runs in passes.  This form fits the budget
// of three wavefronts per SIMD (168 VGPRs, no scratch) with 320-thread workgroups (the 2 x 160 manifold lanes and nothing else) and
// < 80 KB of LDS, so two islands share a CU.  Where the registers went:
//   * the scalars of a constraint (effective masses, impulses, right-hand sides: contact_with_twist_friction.rs:600-630) are used by
//     the EVEN lane of a pair only; the odd lane's copy of those registers was dead.  Every such register now holds two values: the
//     hot one (read by the sweeps) on the even lane, a cold one (read once per substep or once per step: accumulated impulse,
//     restitution seed, the biased right-hand side of the next warm start, distance offset, contact ids ...) parked on the odd lane.
//     A parked value crosses over with one DPP quad permute when it is needed;
//   * the bias-free right-hand side is written straight into the active slot by the pose stage (the relaxed sweep follows it), so the
//     third copy (`rhsR`) and the `relax` switch of the sweep are gone;
//   * the per-body constants of the owner lanes (increments, principal frame) live in LDS;
//   * generate parks its world points in the (then idle) warm-start rows instead of twelve registers.
// Arithmetic, operand order and stage order are those of k_island_solve: both forms, the global path and the oracle agree bit for bit.
#pragma once

struct PkPoint {
    V3 pa, pc;          // own torque_dir, ii_torque_dir
    float r_seed;       // even: projected mass r            | odd: restitution seed
    float lam_acc;      // even: impulse                     | odd: accumulated impulse of the earlier substeps
    float rhs_rhsB;     // even: active right-hand side      | odd: biased right-hand side for the next warm start
    float cfm_cfmB;     // even: active cfm factor           | odd: cfm factor for the next biased sweep
};
struct IslPk {
    int id, n;          // own body (LDS index or -1); point count
    bool odd;
    V3 dir, t0, t1;
    V3 sdim, im, stw;   // even: dim1, im1, twa ; odd: -dim2, im2, -twb
    V3 td0, td1, itd0, itd1;
    float mu_cfmf;      // even: friction coefficient        | odd: cfm_factor of this pair
    float twr_erp;      // even: twist r                     | odd: erp_inv_dt of this pair
    float k11_rw0;      // even: k11                         | odd: rhs_wo0
    float k22_rw1;      // even: k22                         | odd: rhs_wo1
    float k12_tb0;      // even: k12                         | odd: tangent bias 0 (pose stage)
    float idet_tb1;     // even: inv_det                     | odd: tangent bias 1
    float tw_ia;        // even: twist impulse               | odd: accumulated twist impulse
    float t0_ia, t1_ia; // even: tangent impulses            | odd: accumulated tangent impulses
    float trhs0_cids;   // even: tangent rhs 0               | odd: contact ids (bits)
    float trhs1;        // even: tangent rhs 1
    float td_d0[4];     // even: twist lever of point k      | odd: distance offset d0 of point k
    PkPoint P[4];
};
// an opaque copy of a lane index: addresses derived from it cannot be hoisted out of the loop the copy is made in (LICM would otherwise
// keep one VGPR per LDS array and lane alive across the whole step — and spill it — instead of folding the array's offset into the access)
RP_DEV int pk_opaque(int x) { asm volatile("" : "+v"(x)); return x; }
RP_DEV float pk_ev(float x) { return dppf<DPP_FROM_EVEN>(x); }
RP_DEV float pk_od(float x) { return dppf<DPP_FROM_ODD>(x); }

// generate (ContactWithTwistFrictionBuilder::generate :58-424), see isl_generate.  Wscr: this lane's row of the warm-start terms.
RP_DEV bool lean_generate(const DevWorld &w, IslPk &h, const IslLds &L, float4 *Wscr, int m, int s, int gid, int lid, bool odd, bool is_static) {
    h.odd = odd; h.id = lid;
    Vel vels = isl_vel(L, lid);
    Xf pose = isl_xf(L, lid);
    V3 im = gid >= 0 ? v3(w.b_eim[gid]) : v3(0, 0, 0);
    Sym3 ii = load_ii(w, gid);
    V3 world_com = pose.t;
    float4 nf = w.p_normal[s];
    V3 dir = -v3(nf);
    V3 sdir = odd ? -dir : dir;
    float friction = nf.w;
    float restitution = w.p_misc[s].x;
    int count = w.p_nsc[s]; if (count > 4) count = 4;
    V3 t0 = orthonormal_vector(dir);
    V3 t1 = cross(dir, t0);
    float inv_num_points = 1.0f / (float)count;
    V3 friction_center = v3(0, 0, 0), tangent_vel = v3(0, 0, 0);
    float twist_warmstart = 0.0f, tw0 = 0.0f, tw1 = 0.0f;
    float info[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int cids = 0;
    bool bouncy_seed = false;
    V3 im2 = dppv<DPP_FROM_ODD>(im);
    V3 imsum = im + im2;
    h.n = count; h.dir = dir; h.t0 = t0; h.t1 = t1; h.im = im;
    const float4 *anchors = odd ? w.sc_a2 : w.sc_a1;
    const float4 *levers = odd ? w.pt_dp2 : w.pt_dp1;
    float4 *LP = odd ? L.F : L.E;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= count) break;
        PkPoint &q = h.P[k];
        float weight = inv_num_points;
        float4 an = PT(anchors, k, s);
        int cid = __float_as_int(PT(w.sc_a2, k, s).w);
        cids |= (cid & 0xff) << (8 * k);
        float4 pimp = PT(w.pt_imp, cid, s);
        V3 wt = v3(PT(w.pt_wst, cid, s));
        float warmstart_impulse = pimp.y;
        float wti0 = dot(wt, t0), wti1 = dot(wt, t1);
        float warmstart_twist_impulse = pimp.z;
        bool is_new = pimp.x == 0.0f;
        float is_bouncy = is_new ? (restitution > 0.0f ? 1.0f : 0.0f) : (restitution >= 1.0f ? 1.0f : 0.0f);
        V3 pw = xf_tp(pose, v3(an));
        float dist = dot(pw - dppv<DPP_FROM_ODD>(pw), dir);
        V3 dp = v3(PT(levers, cid, s));
        V3 point = world_com + dp;
        Wscr[k * WS_STRIDE] = f4(point, 0.0f);
        friction_center = friction_center + point * weight;
        V3 vel = vels.lin + cross(vels.ang, dp);
        twist_warmstart += warmstart_twist_impulse * weight;
        tw0 += wti0 * weight; tw1 += wti1 * weight;
        V3 torque_dir = cross(dp, sdir);
        V3 ii_torque_dir = sym_mul(ii, torque_dir);
        float G = dot(ii_torque_dir, torque_dir);
        float projected_mass = rp_inv(dot(dir, cmul(imsum, dir)) + G + pk_od(G));
        float projected_velocity = dot(vel - dppv<DPP_FROM_ODD>(vel), dir);
        float restitution_seed = is_bouncy * restitution * projected_velocity;
        bouncy_seed |= restitution_seed < 0.0f;
        info[k] = dist - dot(point - dppv<DPP_FROM_ODD>(point), dir);
        q.pa = torque_dir; q.pc = ii_torque_dir;
        const float seed_e = pk_ev(restitution_seed);
        q.r_seed = odd ? seed_e : projected_mass;
        q.lam_acc = odd ? -warmstart_impulse : warmstart_impulse;
        q.rhs_rhsB = 0.0f; q.cfm_cfmB = 1.0f;
        LP[k * RP_ISL_NC_MAX + m] = f4(xf_itp(pose, point), 0.0f);
        if (k <= 2) __builtin_amdgcn_sched_barrier(0);
    }
    float twist_imp = count > 1 ? twist_warmstart : 0.0f;
    V3 dpf = friction_center - world_com;
    float twist_r = 0.0f;
    float tdl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    V3 tw = sym_mul(ii, dir);
    h.stw = odd ? -tw : tw;
    if (count > 1) {
        tdl[0] = len(friction_center - v3(Wscr[0 * WS_STRIDE]));
        tdl[1] = len(friction_center - v3(Wscr[1 * WS_STRIDE]));
        if (count > 2) tdl[2] = len(friction_center - v3(Wscr[2 * WS_STRIDE]));
        if (count > 3) tdl[3] = len(friction_center - v3(Wscr[3 * WS_STRIDE]));
        V3 ii_twist_dir = sym_mul(ii, sdir);
        float Hh = dot(ii_twist_dir, sdir);
        twist_r = rp_inv(Hh + pk_od(Hh));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d0e = pk_ev(info[k]); h.td_d0[k] = odd ? d0e : tdl[k]; }
    float r[3], rhs_wo[2];
    V3 td[2], itd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        V3 tj = j == 0 ? t0 : t1;
        td[j] = cross(dpf, odd ? -tj : tj);
        itd[j] = sym_mul(ii, td[j]);
        float G = dot(itd[j], td[j]);
        r[j] = dot(tj, cmul(imsum, tj)) + G + pk_od(G);
        rhs_wo[j] = dot(tangent_vel, tj);
    }
    {
        float K = dot(itd[0], td[1]);
        r[2] = 2.0f * (K + pk_od(K));
    }
    h.td0 = td[0]; h.td1 = td[1]; h.itd0 = itd[0]; h.itd1 = itd[1];
    (odd ? L.B1 : L.B0)[m] = f4(xf_itp(pose, friction_center), 0.0f);
    V3 dim = cmul(dir, im);
    h.sdim = odd ? -dim : dim;
    const float k12 = r[2] * 0.5f;
    const float inv_det = rp_inv(r[0] * r[1] - k12 * k12);
    float fstatic = is_static ? 1.0f : 0.0f;
    const float cfm_factor = w.prm.dyn_cfm + fstatic * (w.prm.static_cfm - w.prm.dyn_cfm);
    const float erp_inv_dt = w.prm.dyn_erp_inv_dt + fstatic * (w.prm.static_erp_inv_dt - w.prm.dyn_erp_inv_dt);
    h.mu_cfmf = odd ? cfm_factor : friction;
    h.twr_erp = odd ? erp_inv_dt : twist_r;
    h.k11_rw0 = odd ? rhs_wo[0] : r[0];
    h.k22_rw1 = odd ? rhs_wo[1] : r[1];
    h.k12_tb0 = odd ? 0.0f : k12;
    h.idet_tb1 = odd ? 0.0f : inv_det;
    h.tw_ia = odd ? -twist_imp : twist_imp;
    h.t0_ia = odd ? -tw0 : tw0;
    h.t1_ia = odd ? -tw1 : tw1;
    h.trhs0_cids = odd ? __int_as_float(cids) : rhs_wo[0];
    h.trhs1 = rhs_wo[1];
    return bouncy_seed;
}

// Pose-dependent half of update / refresh_rhs_wo_bias (:426-554), see isl_pose_stage.  The bias-free right-hand sides go straight
// into the active slots (the relaxed sweep is what follows a pose stage; a warm start overwrites them from the parked values).
RP_DEV void lean_pose_stage(const DevWorld &w, IslPk &h, const IslLds &L, int m, float solved_dt) {
    Xf x = isl_xf(L, h.id);
    V3 tangent_delta = v3(0.0f, 0.0f, 0.0f) * solved_dt;
    const float inv_dt = w.prm.inv_dt_sub, maxcv = w.prm.max_corrective_velocity;
    const float erp_inv_dt = pk_od(h.twr_erp), cfm_factor = pk_od(h.mu_cfmf);
    const float4 *LP = h.odd ? L.F : L.E;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= h.n) break;
        PkPoint &p = h.P[k];
        V3 pw = xf_tp(x, v3(LP[k * RP_ISL_NC_MAX + m]));
        pw = sel(h.odd, pw, pw + tangent_delta);
        V3 p2 = dppv<DPP_FROM_ODD>(pw);
        float dist = pk_od(h.td_d0[k]) + dot(pw - p2, h.dir);
        float rhs_wo_bias = rp_max(dist, 0.0f) * inv_dt;
        float rhs_bias = rp_clamp(dist * erp_inv_dt, -maxcv, 0.0f);
        const float rhsB = pk_ev(rhs_wo_bias + rhs_bias);
        const float cfmB = pk_ev(dist <= 0.0f ? cfm_factor : 1.0f);
        p.rhs_rhsB = h.odd ? rhsB : rhs_wo_bias;
        p.cfm_cfmB = h.odd ? cfmB : 1.0f;
    }
    V3 pf = xf_tp(x, v3((h.odd ? L.B1 : L.B0)[m]));
    pf = sel(h.odd, pf, pf + tangent_delta);
    V3 pf2 = dppv<DPP_FROM_ODD>(pf);
    const float tb0 = pk_ev(dot(pf - pf2, h.t0) * inv_dt), tb1 = pk_ev(dot(pf - pf2, h.t1) * inv_dt);
    h.k12_tb0 = h.odd ? tb0 : h.k12_tb0;
    h.idet_tb1 = h.odd ? tb1 : h.idet_tb1;
    const float rw0 = pk_od(h.k11_rw0), rw1 = pk_od(h.k22_rw1);
    h.trhs0_cids = h.odd ? h.trhs0_cids : rw0;
    h.trhs1 = rw1;
}

// update + warm-start terms (:426-522, :633-678), see isl_ws_terms.  PHASE 1: the update and the linear terms; PHASE 2: the angular
// terms into the same rows (after the owners of the linear halves have read theirs).
template <int PHASE>
RP_DEV void lean_ws_terms(const DevWorld &w, IslPk &h, float4 *W, int t) {
    const float wc = w.prm.p.warmstart_coefficient;
    const bool ws = wc != 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= h.n) break;
        PkPoint &p = h.P[k];
        if (PHASE == 1) {
            p.rhs_rhsB = pk_od(p.rhs_rhsB); p.cfm_cfmB = pk_od(p.cfm_cfmB); // both lanes take the parked value (the odd lane keeps it)
            const float lam0 = pk_ev(p.lam_acc);
            p.lam_acc = h.odd ? p.lam_acc + lam0 : p.lam_acc * wc;          // acc += lam ; lam *= wc
        }
        if (ws) {
            const float lam = pk_ev(p.lam_acc);
            if (PHASE == 1) W[k * WS_STRIDE + t] = f4(h.sdim * lam, 0.0f);
            else W[k * WS_STRIDE + t] = f4(p.pc * lam, 0.0f);
        }
    }
    if (PHASE == 1) {
        const float r0 = pk_od(h.k11_rw0) + pk_od(h.k12_tb0), r1 = pk_od(h.k22_rw1) + pk_od(h.idet_tb1);
        h.trhs0_cids = h.odd ? h.trhs0_cids : r0; h.trhs1 = r1;
        const float a0 = pk_ev(h.t0_ia), a1 = pk_ev(h.t1_ia), aw = pk_ev(h.tw_ia);
        h.t0_ia = h.odd ? h.t0_ia + a0 : h.t0_ia * wc;
        h.t1_ia = h.odd ? h.t1_ia + a1 : h.t1_ia * wc;
        h.tw_ia = h.odd ? h.tw_ia + aw : h.tw_ia * wc;
    }
    if (ws) {
        const float i0 = pk_ev(h.t0_ia), i1 = pk_ev(h.t1_ia);
        const float s0 = h.odd ? -i0 : i0, s1 = h.odd ? -i1 : i1;
        const float tw = pk_ev(h.tw_ia);
        if (PHASE == 1) W[4 * WS_STRIDE + t] = f4(cmul(h.t0 * s0 + h.t1 * s1, h.im), __int_as_float(h.n));
        else {
            W[4 * WS_STRIDE + t] = f4(h.itd0 * i0 + h.itd1 * i1, __int_as_float(h.n));
            if (h.n > 1) W[5 * WS_STRIDE + t] = f4(h.stw * tw, 0.0f);
        }
    }
}

// solve (:680-781), see isl_solve_t.  The active right-hand sides were chosen by the stage before (warm start: biased; pose stage:
// bias-free), so the biased and the relaxed sweep are the same code.
template <bool F4> RP_DEV void lean_solve_t(IslPk &h, const IslLds &L, bool friction) {
    const int hn = F4 ? 4 : h.n;
    Vel v = isl_vel(L, h.id);
    float imp[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= hn) break;
        PkPoint &p = h.P[k];
        float X = dot(h.dir, v.lin), Y = dot(p.pa, v.ang);
        float S = X + Y;
        float dvel = S - pk_od(X) + pk_od(Y) + p.rhs_rhsB;
        float new_impulse = p.cfm_cfmB * rp_max(p.lam_acc - p.r_seed * dvel, 0.0f);
        float dl = pk_ev(new_impulse - p.lam_acc);
        p.lam_acc = h.odd ? p.lam_acc : new_impulse;
        imp[k] = new_impulse;
        v.lin = v.lin + h.sdim * dl;
        v.ang = v.ang + p.pc * dl;
    }
    if (friction) {
        float tangent_limit = 0.0f, twist_limit = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (k >= hn) break; tangent_limit += imp[k]; twist_limit += imp[k] * h.td_d0[k]; }
        tangent_limit *= h.mu_cfmf; twist_limit *= h.mu_cfmf;
        if (hn > 1) {
            V3 w2 = dppv<DPP_FROM_ODD>(v.ang);
            float dvel = dot(h.dir, v.ang - w2) + 0.0f;
            float new_impulse = rp_clamp(h.tw_ia - h.twr_erp * dvel, -twist_limit, twist_limit);
            float dl = pk_ev(new_impulse - h.tw_ia);
            h.tw_ia = h.odd ? h.tw_ia : new_impulse;
            v.ang = v.ang + h.stw * dl;
        }
        {
            float X0 = dot(h.t0, v.lin), Y0 = dot(h.td0, v.ang), X1 = dot(h.t1, v.lin), Y1 = dot(h.td1, v.ang);
            float S0 = X0 + Y0, S1 = X1 + Y1;
            float dvel_0 = S0 - pk_od(X0) + pk_od(Y0) + h.trhs0_cids;
            float dvel_1 = S1 - pk_od(X1) + pk_od(Y1) + h.trhs1;
            float d0 = (h.k22_rw1 * dvel_0 - h.k12_tb0 * dvel_1) * h.idet_tb1;
            float d1 = (h.k11_rw0 * dvel_1 - h.k12_tb0 * dvel_0) * h.idet_tb1;
            float n0 = h.t0_ia - d0, n1 = h.t1_ia - d1;
            float l = sqrtf(n0 * n0 + n1 * n1);
            if (l > tangent_limit) { float sc = tangent_limit / l; n0 *= sc; n1 *= sc; }
            float dl0 = pk_ev(n0 - h.t0_ia), dl1 = pk_ev(n1 - h.t1_ia);
            h.t0_ia = h.odd ? h.t0_ia : n0; h.t1_ia = h.odd ? h.t1_ia : n1;
            float s0 = h.odd ? -dl0 : dl0, s1 = h.odd ? -dl1 : dl1;
            v.lin = v.lin + cmul(h.t0 * s0 + h.t1 * s1, h.im);
            v.ang = v.ang + (h.itd0 * dl0 + h.itd1 * dl1);
        }
    }
    isl_set_vel(L, h.id, v);
}
RP_DEV void lean_solve(IslPk &h, const IslLds &L, bool friction) {
    if (__all(h.n == 4)) lean_solve_t<true>(h, L, friction); else lean_solve_t<false>(h, L, friction);
}

// apply_restitution (:568-597), see isl_restitution
RP_DEV void lean_restitution(IslPk &h, const IslLds &L) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (k >= h.n) break; any |= pk_od(h.P[k].r_seed) < 0.0f; }
    if (!any) return;
    Vel v = isl_vel(L, h.id);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= h.n) break;
        PkPoint &p = h.P[k];
        const float seed = pk_od(p.r_seed), acc = pk_od(p.lam_acc);
        float X = dot(h.dir, v.lin), Y = dot(p.pa, v.ang);
        float S = X + Y;
        float dvel = S - pk_od(X) + pk_od(Y) + seed;
        bool gate = seed < 0.0f && (acc + p.lam_acc) > 0.0f;
        float new_impulse = gate ? rp_max(p.lam_acc - p.r_seed * dvel, 0.0f) : p.lam_acc;
        float dl = pk_ev(new_impulse - p.lam_acc);
        p.lam_acc = h.odd ? p.lam_acc : new_impulse;
        v.lin = v.lin + h.sdim * dl;
        v.ang = v.ang + p.pc * dl;
    }
    isl_set_vel(L, h.id, v);
}

// writeback_impulses (:783-829), see isl_writeback; called by BOTH lanes of a pair (the parked values cross over), the even lane stores
RP_DEV void lean_writeback(const DevWorld &w, const IslPk &h, int s) {
    V3 wtw = h.t0 * rp_canon0(h.t0_ia) + h.t1 * rp_canon0(h.t1_ia);
    wtw = v3(rp_canon0(wtw.x), rp_canon0(wtw.y), rp_canon0(wtw.z));
    const int cids = __float_as_int(pk_od(h.trhs0_cids));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= h.n) break;
        const float acc = pk_od(h.P[k].lam_acc);
        if (!h.odd) {
            int cid = (cids >> (8 * k)) & 0xff;
            PT(w.pt_imp, cid, s) = make_float4(rp_canon0(acc + h.P[k].lam_acc), rp_canon0(h.P[k].lam_acc), rp_canon0(h.tw_ia), 0.0f);
            PT(w.pt_wst, cid, s) = f4(wtw, 0.0f);
        }
    }
}

// One workgroup = TWO islands, 640 manifold lanes (+ 128 validating lanes: twelve wavefronts, three per SIMD at 168 VGPRs — the layout the CU actually
// accepts: the waves of a workgroup are spread so that a SIMD holds ceil(waves / 4) of them and two workgroups never interleave, so
// two separate 5-wave workgroups would need 4 x 168 registers on one SIMD and are NOT co-resident, whatever the occupancy API answers;
// measured with tools/ubench/coresident.hip, profiles/r05_ubench_coresident_*.txt).  Threads [0, 320) hold the island of the
// workgroup's even slot, [320, 640) the one of its odd slot; inside a half: lanes 2m, 2m+1 = manifold m, threads [0, nb) also own
// the linear half of body t, threads [64, 64 + nb) the angular half of body t - 64.  The two islands share every barrier (a stage of
// the sweep is a stage of both).  Stage order and fused-step protocol: island_solve_body (rp_islands.hip).
#define LEAN_HALF ISL_THREADS_DENSE        // threads per island
#define LEAN_VALIDATORS 128                // two more wavefronts (twelve = three per SIMD, the same register budget): they hold no manifold and
                                           // prove, under cover of generate, that the fused step needs neither broad nor narrow phase
#define LEAN_THREADS (2 * LEAN_HALF + LEAN_VALIDATORS)
struct LeanLds { // per island
    float4 B_lin[RP_ISL_NB_MAX], B_ang[RP_ISL_NB_MAX], B_rot[RP_ISL_NB_MAX], B_trans[RP_ISL_NB_MAX];
    float4 O_incl[RP_ISL_NB_MAX], O_inca[RP_ISL_NB_MAX], O_invpi[RP_ISL_NB_MAX], O_pframe[RP_ISL_NB_MAX]; // owner constants: .w of incl = first row, of inca = row count, of invpi = body flags
    float4 L_E[4 * RP_ISL_NC_MAX], L_F[4 * RP_ISL_NC_MAX], L_B0[RP_ISL_NC_MAX], L_B1[RP_ISL_NC_MAX];
    float4 W[WS_SLOTS_2PHASE * WS_STRIDE];
    int any_bouncy, nls;
};
// `nsteps`: several fused steps in one launch, exactly as in island_solve_body (rp_islands.hip: per-step validate / arrive / commit on
// cumulative counts, abort fields by step parity, one island PAIR per workgroup at most, no pair into another island).
template <bool WIDE> __device__ __forceinline__ void island_solve_lean(const DevWorld &w, int has_restitution, int fast, int retire, int fused, int nsteps = 1) {
    const bool early_poll = ((fused & 2) != 0) || ((nsteps & (1 << 16)) != 0); // the verdict on a step asked for in substep 0 already (island_solve_body, rp_islands.hip)
    fused &= 1; nsteps &= 0xffff;
    const bool aborted = (fast && w.flags[FL_FAST_ABORT]) || lean_dead(w);
    if (retire && blockIdx.x == 0) {
        if (threadIdx.x == 0) { w.flags[FL_SEQ] += (fused && nsteps > 1) ? nsteps : 1; if (!aborted && !fused) w.flags[FL_STEP] += 1; if (fused) w.flags[FL_FULL_UPDATES] = 0; }
        __threadfence(); __syncthreads();
        publish_flags(w);
    }
    if (aborted) return;
    __shared__ int s_abort, s_go, s_slp[2], s_cross, s_early;
    if (threadIdx.x == 0) { s_cross = 0; s_early = 0; }
    __shared__ LeanLds LD[2];
    __shared__ int S_a[RP_ISL_NC_MAX], S_b[RP_ISL_NC_MAX], S_c[RP_ISL_NC_MAX], S_d[RP_ISL_NC_MAX];
    const int n_islands = w.flags[FL_N_ISLANDS];
#ifdef RP_ISL_PROFILE
    long long t_fused0 = (long long)__builtin_readcyclecounter();
#endif
    if (fused) {
        // the islands of this workgroup BEYOND its first two are validated up front; the first two by the validator wavefronts under
        // cover of generate (below).  FL_ARRIVE counts arrivals in its low 16 bits and aborting workgroups above.
        const int t = threadIdx.x;
        if (t == 0) {
            s_abort = 0; s_slp[0] = 0; s_slp[1] = 0;
            if (blockIdx.x == 0 && fused_world_abort<WIDE>(w)) s_abort = 1;
            if constexpr (WIDE) if (w.sleep_enabled) sleep_begin_scan(w);
        }
        __syncthreads();
        bool bad = false;
        const int stamp_before = (WIDE && w.sleep_enabled) ? pi_stamp_before(w) : 0;
        for (int base = 2 * (blockIdx.x + gridDim.x); base < n_islands; base += 2 * gridDim.x) {
            for (int isl = base; isl < base + 2 && isl < n_islands; ++isl) {
                int slp = 0;
                if (fused_validate_island<WIDE>(w, isl, t, blockDim.x, stamp_before, slp)) bad = true;
                if constexpr (WIDE) if (w.sleep_enabled && fused_sleep_abort(__syncthreads_or(slp))) bad = true;
            }
        }
        if (bad) s_abort = 1;
        if (2 * (int)blockIdx.x >= n_islands && !(nsteps > 1 && n_islands <= 2 * (int)gridDim.x)) { // no island at all: arrive now (a launch of several steps: once per step, below)
            __syncthreads();
            if (t == 0) atomicAdd(&w.flags[FL_ARRIVE], 1 + (s_abort ? (1 << 16) : 0));
        }
#ifdef RP_ISL_PROFILE
        if (blockIdx.x == 0 && threadIdx.x == 0) w.dbg[10] += (long long)__builtin_readcyclecounter() - t_fused0;
#endif
    }
    bool decided = !fused, go = true;
    const bool validator = threadIdx.x >= 2 * LEAN_HALF;  // wave-uniform, like `half`: LEAN_HALF is five wavefronts
    const int half = (!validator && threadIdx.x >= LEAN_HALF) ? 1 : 0;
    const int t0 = validator ? LEAN_HALF : (int)threadIdx.x - half * LEAN_HALF; // (a validator's lane index lies beyond every manifold and body)
    const int nst_global = w.flags[FL_N_STAGES];
    const rp_integration_params &prm = w.prm.p;
    const bool fib = prm.friction_in_bias_pass || prm.num_internal_stabilization_iterations == 0;
    const bool wsc = prm.warmstart_coefficient != 0.0f;

    const int ns = (fused && nsteps > 1 && n_islands <= 2 * (int)gridDim.x) ? nsteps : 1; // (more island pairs than workgroups: one step, pairs in rounds)
    for (int step = 0; step < ns; ++step) {
    const int arrive_target = (step + 1) * (int)gridDim.x, ab_shift = 16 + 8 * (step & 1); // (see island_solve_body)
    const unsigned ab_one = 1u << ab_shift;
    if (step > 0) {
        decided = false;
        if (threadIdx.x == 0) { s_abort = s_cross; s_slp[0] = 0; s_slp[1] = 0; s_early = 0; }
        __syncthreads();
    }
    if (ns > 1 && 2 * (int)blockIdx.x >= n_islands) { // a workgroup without an island follows the protocol of every step
        if (threadIdx.x == 0) {
            atomicAdd((unsigned *)&w.flags[FL_ARRIVE], 1u + (s_abort ? ab_one : 0u));
            unsigned v; int spins = 0;
            while (((v = __hip_atomic_load((unsigned *)&w.flags[FL_ARRIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffffu) < (unsigned)arrive_target && ((v >> ab_shift) & 0xffu) == 0 && ++spins < (1 << 23)) __builtin_amdgcn_s_sleep(8);
            s_go = (v & 0xffffu) >= (unsigned)arrive_target && ((v >> ab_shift) & 0xffu) == 0;
        }
        __syncthreads();
        go = s_go != 0;
    }
    for (int base = 2 * blockIdx.x; base < n_islands; base += 2 * gridDim.x) {
        __syncthreads(); // the previous islands of this workgroup are fully written back
#ifdef RP_ISL_PROFILE
        long long t_prev = (long long)__builtin_readcyclecounter();
#endif
        // (the stage sort of an island runs once per layout change, on the whole workgroup, one island after the other)
        for (int k = 0; k < 2; ++k) {
            const int i2 = base + k;
            if (i2 < n_islands && !w.isl_sorted[i2]) island_sort(w, i2, w.isl_nc[i2], w.isl_cons_begin[i2], nst_global, S_a, S_b, S_c, S_d);
        }
        const int isl = base + half;
        const bool have = isl < n_islands;              // an odd island count leaves the last workgroup's second half idle (it still meets every barrier)
        const int nb = have ? w.isl_nb[isl] : 0, nc = have ? w.isl_nc[isl] : 0;
        const int bb = have ? w.isl_body_begin[isl] : 0, cb = have ? w.isl_cons_begin[isl] : 0;
        const int t = pk_opaque(t0), m = t >> 1;
        const bool odd = (t & 1) != 0;
        LeanLds &D = LD[half];
        IslLds L;
        L.lin = D.B_lin; L.ang = D.B_ang; L.rot = D.B_rot; L.trans = D.B_trans; L.E = D.L_E; L.F = D.L_F; L.B0 = D.L_B0; L.B1 = D.L_B1;
        const int bt = t & (RP_ISL_NB_MAX - 1);
        const bool role_lin = t < nb, role_ang = t >= RP_ISL_NB_MAX && t < RP_ISL_NB_MAX + nb;
        if (role_lin || role_ang) {
            const int g = w.isl_bodies[bb + bt];
            V3 lin, ang, trans, incl, inca; Q4 rot;
            body_begin(w, g, lin, ang, rot, trans, incl, inca);
            if (role_lin) {
                D.B_lin[bt] = f4(lin, 0.0f); D.B_rot[bt] = f4(rot); D.B_trans[bt] = f4(trans, 0.0f);
                D.O_incl[bt] = f4(incl, __int_as_float(w.isl_inc_begin[bb + bt]));
            } else {
                D.B_ang[bt] = f4(ang, 0.0f);
                D.O_inca[bt] = f4(inca, __int_as_float(w.isl_inc_cnt[bb + bt]));
                D.O_invpi[bt] = f4(v3(w.b_invpi[g]), __int_as_float(w.b_flags[g]));
                D.O_pframe[bt] = w.b_pframe[g];
            }
        }
        if (t == 0) { D.any_bouncy = 0; D.nls = have ? w.isl_nstages[isl] : 0; }
        const bool live = m < nc;
        int myq = -1, own_g = -1, own_l = -1, ws_row0 = 0, slot = -1;
        bool pair_static = false;
        if (live) {
            slot = w.isl_cons[cb + m]; myq = w.isl_cstage[cb + m];
            const int l1 = w.isl_cl1[cb + m], l2 = w.isl_cl2[cb + m];
            own_g = (odd ? w.isl_cg2 : w.isl_cg1)[cb + m]; own_l = odd ? l2 : l1;
            pair_static = l1 < 0 || l2 < 0;
            ws_row0 = w.isl_inc_pos[2 * cb + t];
        }
        __syncthreads();
        const int nls = LD[0].nls > LD[1].nls ? LD[0].nls : LD[1].nls; // the stages of the longer island: the shorter one idles through the rest
        ISL_STAMP(0);
        IslPk h;
        h.n = 0; h.id = -1; h.odd = odd;
        if (live) {
            if (lean_generate(w, h, L, D.W + t, m, slot, own_g, own_l, odd, pair_static) && !odd) D.any_bouncy = 1;
            lean_pose_stage(w, h, L, m, 0.0f);
        }
        if (validator && fused && base == 2 * (int)blockIdx.x) {
            // one item (a body's collider, an active pair, a pair without solver contacts) per lane and round, over both islands
            const int vt = threadIdx.x - 2 * LEAN_HALF, stamp_before = (WIDE && w.sleep_enabled) ? pi_stamp_before(w) : 0;
            for (int i2 = base; i2 < base + 2 && i2 < n_islands; ++i2) {
                int slp = 0;
                bool cross = false;
                if (fused_validate_island<WIDE>(w, i2, vt, LEAN_VALIDATORS, stamp_before, slp, 0, 0, ns > 1 ? &cross : nullptr)) s_abort = 1;
                if (cross) { s_cross = 1; if (w.flags[FL_GRID_TIMEOUT] == 0) w.flags[FL_GRID_TIMEOUT] = 2; }
                if constexpr (WIDE) if (slp) atomicOr(&s_slp[i2 - base], slp);
            }
        }
        ISL_STAMP(1);

        for (int sub = 0; sub < w.prm.num_substeps; ++sub) {
            const float solved_dt = (float)sub * w.prm.dt_sub;
            const int t = pk_opaque(t0), bt = t & (RP_ISL_NB_MAX - 1), m = t >> 1, ws_row = pk_opaque(ws_row0);
            if (sub == 0) __syncthreads(); // generate's rows of W have been read back by their lanes
            if (live) lean_ws_terms<1>(w, h, D.W, ws_row);
            __syncthreads();
            if (early_poll && sub == 1 && !decided && s_early == 2) { go = false; decided = true; break; } // (an abort seen in substep 0's pose stage: uniform, barriers ago)
            if (fused && sub == 0 && base == 2 * (int)blockIdx.x && threadIdx.x == 0) atomicAdd((unsigned *)&w.flags[FL_ARRIVE], 1u + ((s_abort || (WIDE && (fused_sleep_abort(s_slp[0]) || fused_sleep_abort(s_slp[1])))) ? ab_one : 0u)); // this workgroup validated all of its islands
            ISL_STAMP(2);
            if (role_lin) { // S2 increment (worker.rs:235-284), then the warm start of this body in sweep order
                const float4 oi = D.O_incl[bt];
                V3 lin = v3(D.B_lin[bt]) + v3(oi);
                if (wsc) isl_ws_accumulate_lin<true>(D.W, __float_as_int(oi.w), __float_as_int(D.O_inca[bt].w), lin);
                D.B_lin[bt] = f4(lin, 0.0f);
            }
            __syncthreads();
            if (live && wsc) lean_ws_terms<2>(w, h, D.W, ws_row);
            __syncthreads();
            if (role_ang) {
                const float4 oa = D.O_inca[bt], op = D.O_invpi[bt];
                V3 lin_unused = v3(0, 0, 0), ang = v3(D.B_ang[bt]);
                body_increment(w, __float_as_int(op.w), lin_unused, ang, q4(D.B_rot[bt]), v3(0, 0, 0), v3(oa), v3(op), q4(D.O_pframe[bt]));
                if (wsc) isl_ws_accumulate_ang<true>(D.W, __float_as_int(D.O_incl[bt].w), __float_as_int(oa.w), ang);
                D.B_ang[bt] = f4(ang, 0.0f);
            }
            __syncthreads();
            ISL_STAMP(3);
            for (int it = 0; it < prm.num_internal_pgs_iterations; ++it)
                for (int q = 0; q < nls; ++q) { if (myq == q) lean_solve(h, L, fib); __syncthreads(); }
            ISL_STAMP(4);
            if (t < nb) { // S6
                V3 lin = v3(D.B_lin[t]), ang = v3(D.B_ang[t]), trans = v3(D.B_trans[t]); Q4 rot = q4(D.B_rot[t]);
                body_integrate(w, __float_as_int(D.O_invpi[t].w), lin, ang, rot, trans);
                D.B_lin[t] = f4(lin, 0.0f); D.B_ang[t] = f4(ang, 0.0f); D.B_rot[t] = f4(rot); D.B_trans[t] = f4(trans, 0.0f);
            }
            __syncthreads();
            ISL_STAMP(5);
            if (live) lean_pose_stage(w, h, L, m, solved_dt + w.prm.dt_sub);
            else if (early_poll && fused && !decided && sub == 0 && threadIdx.x == LEAN_THREADS - 1 && base == 2 * (int)blockIdx.x) {
                // (a validating lane, idle by now: is an abort of this step standing?  Then the step ends at the top of substep 1)
                const unsigned v = __hip_atomic_load((unsigned *)&w.flags[FL_ARRIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (((v >> ab_shift) & 0xffu) != 0) s_early = 2;
            }
            ISL_STAMP(6);
            for (int it = 0; it < prm.num_internal_stabilization_iterations; ++it)
                for (int q = 0; q < nls; ++q) { if (myq == q) lean_solve(h, L, true); __syncthreads(); }
            ISL_STAMP(7);
        }
        if (go && has_restitution && (LD[0].any_bouncy | LD[1].any_bouncy))
            for (int q = 0; q < nls; ++q) { if (myq == q && D.any_bouncy) lean_restitution(h, L); __syncthreads(); }
        if (!decided) { // fused: nothing leaves the workgroup before every workgroup validated its islands
#ifdef RP_ISL_PROFILE
            long long t_w0 = (long long)__builtin_readcyclecounter();
#endif
            if (threadIdx.x == 0) {
                int spins = 0; unsigned v;
                while (((v = __hip_atomic_load((unsigned *)&w.flags[FL_ARRIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffffu) < (unsigned)arrive_target) {
                    if (ns > 1 && ((v >> ab_shift) & 0xffu) != 0) break; // an abort of THIS step is final
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1 << 22)) { // ~1 s: a workgroup never became resident — turn the wait into an abort of the whole launch (see island_solve_body)
                        unsigned seen = v;
                        while ((seen & 0xffffu) < (unsigned)arrive_target &&
                               !__hip_atomic_compare_exchange_strong((unsigned *)&w.flags[FL_ARRIVE], &seen, seen + ab_one, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { }
                        if ((seen & 0xffffu) < (unsigned)arrive_target) { v = seen + ab_one; __hip_atomic_store(&w.flags[FL_GRID_TIMEOUT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        v = seen; break;
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
        if (live) lean_writeback(w, h, slot);
        if (t < nb) {
            const int g = w.isl_bodies[bb + t];
            body_writeback(w, g, v3(D.B_lin[t]), v3(D.B_ang[t]), q4(D.B_rot[t]), v3(D.B_trans[t]));
            if (ns > 1 && w.b_quar[g]) s_cross = 1; // a body went non-finite: no further step in this launch
        }
        ISL_STAMP(8);
#ifdef RP_ISL_PROFILE
        if (blockIdx.x == 0 && threadIdx.x == 0) w.dbg[63] += (base + 1 < n_islands) ? 2 : 1;
#endif
    }
    if (!go) break;
    __syncthreads(); // this step's write-back is visible to the whole workgroup before the next step reads it
    } // steps of this launch
    if (fused) { // the last workgroup to leave retires the step (or not, when aborted) and re-arms the counters
        __syncthreads();
        if (threadIdx.x == 0) {
            if (atomicAdd(&w.flags[FL_DEPART], 1) == (int)gridDim.x - 1) {
                const unsigned v = __hip_atomic_load((unsigned *)&w.flags[FL_ARRIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w.flags[FL_STEP] += (v >> 16) == 0 ? ns : (int)(((v & 0xffffu) - 1u) / gridDim.x); // (the steps every workgroup committed: island_solve_body)
                if ((v >> 16) != 0) w.flags[FL_FAST_ABORT] = 1;
                __hip_atomic_store(&w.flags[FL_ARRIVE], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&w.flags[FL_DEPART], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
