// rp_island_stages.h — the stages of the per-island megakernel shared by its two forms: k_island_solve (rp_islands.hip, one island per CU)
// and k_island_solve_dense (rp_islands_lean.hip, two islands per CU): generate / pose stage / warm-start terms / restitution /
// write-back of the lane-pair constraint, the body-centric warm-start accumulation and the per-island stage sort.
#pragma once
#include "rp_global.h"
#include "rp_lanepair.h"
#include "rp_pairs.h"
#include "rp_sleep_observe.h"

RP_DEV bool is_dyn(const DevWorld &w, int b) { return body_active(w, b); } // awake dynamic bodies: the active set

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
// generate (S1, ContactWithTwistFrictionBuilder::generate :58-424) by the lane pair: every lane builds
// its own body's half (world points, torque arms, inertia products), the halves of each effective mass
// meet through DPP, the even lane keeps the scalars.  `gid` / `lid` = the lane's own body as arena
// index / island-local index (-1 = world-attached side).  Returns true on the even lane when a
// restitution seed is armed.
RP_DEV bool isl_generate(const DevWorld &w, IslSide &h, const IslLds &L, int m, int s, int gid, int lid, bool odd, bool is_static) {
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
    V3 t0 = orthonormal_vector(dir); // contact_constraint/mod.rs:27-46
    V3 t1 = cross(dir, t0);
    float inv_num_points = 1.0f / (float)count;
    V3 friction_center = v3(0, 0, 0), tangent_vel = v3(0, 0, 0);
    float twist_warmstart = 0.0f, tw0 = 0.0f, tw1 = 0.0f;
    V3 points0 = v3(0, 0, 0), points1 = points0, points2 = points0, points3 = points0;
    int cids = 0;
    bool bouncy_seed = false;
    V3 im2 = dppv<DPP_FROM_ODD>(im);
    V3 imsum = im + im2;                   // even lane: im1 + im2
    h.n = count; h.dir = dir; h.t0 = t0; h.t1 = t1; h.im = im;
    const float4 *anchors = odd ? w.sc_a2 : w.sc_a1;
    const float4 *levers = odd ? w.pt_dp2 : w.pt_dp1;
    float4 *LP = odd ? L.F : L.E;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= count) break;
        SidePoint &q = h.P[k];
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
        if (k == 0) points0 = point; else if (k == 1) points1 = point; else if (k == 2) points2 = point; else points3 = point;
        friction_center = friction_center + point * weight;
        V3 vel = vels.lin + cross(vels.ang, dp);
        twist_warmstart += warmstart_twist_impulse * weight;
        tw0 += wti0 * weight; tw1 += wti1 * weight;
        // tangent_velocity is always zero in this scope (no contact-modification hooks)
        V3 torque_dir = cross(dp, sdir);
        V3 ii_torque_dir = sym_mul(ii, torque_dir);
        float G = dot(ii_torque_dir, torque_dir);
        float projected_mass = rp_inv(dot(dir, cmul(imsum, dir)) + G + dppf<DPP_FROM_ODD>(G));
        float projected_velocity = dot(vel - dppv<DPP_FROM_ODD>(vel), dir);
        float restitution_seed = is_bouncy * restitution * projected_velocity;
        bouncy_seed |= restitution_seed < 0.0f;
        float info_dist = dist - dot(point - dppv<DPP_FROM_ODD>(point), dir);
        q.rhs = 0.0f; q.cfm = 1.0f; q.lam = warmstart_impulse; q.acc = -warmstart_impulse;
        q.pa = torque_dir; q.r = projected_mass; q.seed = restitution_seed;
        q.pc = ii_torque_dir; q.d0 = info_dist;
        q.rhsR = 0.0f; q.rhsB = 0.0f; q.cfmB = 1.0f;
        LP[k * RP_ISL_NC_MAX + m] = f4(xf_itp(pose, point), 0.0f);
    }
    h.cids = cids;
    float twist_imp = count > 1 ? twist_warmstart : 0.0f;
    V3 dpf = friction_center - world_com;
    float twist_r = 0.0f;
    h.td[0] = 0.0f; h.td[1] = 0.0f; h.td[2] = 0.0f; h.td[3] = 0.0f;
    V3 tw = sym_mul(ii, dir);
    h.stw = odd ? -tw : tw;
    if (count > 1) {
        h.td[0] = len(friction_center - points0);
        h.td[1] = len(friction_center - points1);
        if (count > 2) h.td[2] = len(friction_center - points2);
        if (count > 3) h.td[3] = len(friction_center - points3);
        V3 ii_twist_dir = sym_mul(ii, sdir);
        float Hh = dot(ii_twist_dir, sdir);
        twist_r = rp_inv(Hh + dppf<DPP_FROM_ODD>(Hh));
    }
    float r[3], rhs_wo[2];
    V3 td[2], itd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        V3 tj = j == 0 ? t0 : t1;
        td[j] = cross(dpf, odd ? -tj : tj);
        itd[j] = sym_mul(ii, td[j]);
        float G = dot(itd[j], td[j]);
        r[j] = dot(tj, cmul(imsum, tj)) + G + dppf<DPP_FROM_ODD>(G);
        rhs_wo[j] = dot(tangent_vel, tj);
    }
    {
        float K = dot(itd[0], td[1]);
        r[2] = 2.0f * (K + dppf<DPP_FROM_ODD>(K));
    }
    h.td0 = td[0]; h.td1 = td[1]; h.itd0 = itd[0]; h.itd1 = itd[1];
    h.mu = friction; h.twist_r = twist_r;
    h.rhs_wo0 = rhs_wo[0]; h.rhs_wo1 = rhs_wo[1];
    h.k11 = r[0]; h.k22 = r[1];
    h.tw_imp = twist_imp; h.tw_acc = -twist_imp; h.t_imp0 = tw0; h.t_imp1 = tw1;
    h.t_acc0 = -tw0; h.t_acc1 = -tw1; h.t_rhs0 = rhs_wo[0]; h.t_rhs1 = rhs_wo[1]; h.tb0 = 0.0f; h.tb1 = 0.0f;
    (odd ? L.B1 : L.B0)[m] = f4(xf_itp(pose, friction_center), 0.0f);
    // loop invariants of the sweeps (same expressions the per-colour path re-evaluates every sweep)
    V3 dim = cmul(dir, im);
    h.sdim = odd ? -dim : dim;
    h.k12 = r[2] * 0.5f;
    h.inv_det = rp_inv(h.k11 * h.k22 - h.k12 * h.k12);
    // is_static (a world-attached side on either lane) comes from the island's side tables: no cross-lane read, hence no hazard
    // when the two lanes of the pair disagree on it (a dominated body may sit on the odd side, a fixed body is always even)
    float fstatic = is_static ? 1.0f : 0.0f;
    h.cfm_factor = w.prm.dyn_cfm + fstatic * (w.prm.static_cfm - w.prm.dyn_cfm);
    h.erp_inv_dt = w.prm.dyn_erp_inv_dt + fstatic * (w.prm.static_erp_inv_dt - w.prm.dyn_erp_inv_dt);
    return bouncy_seed;
}

// Pose-dependent half of update / refresh_rhs_wo_bias (contact_with_twist_friction.rs:426-554), for the
// poses currently in LDS.  `solved_dt` only scales the (zero) tangent velocity.  m = manifold index.
RP_DEV void isl_pose_stage(const DevWorld &w, IslSide &h, const IslLds &L, int m, float solved_dt) {
    Xf x = isl_xf(L, h.id);
    V3 tangent_delta = v3(0.0f, 0.0f, 0.0f) * solved_dt;
    float inv_dt = w.prm.inv_dt_sub, maxcv = w.prm.max_corrective_velocity;
    const float4 *LP = h.odd ? L.F : L.E;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= h.n) break;
        SidePoint &p = h.P[k];
        V3 pw = xf_tp(x, v3(LP[k * RP_ISL_NC_MAX + m]));
        pw = sel(h.odd, pw, pw + tangent_delta);          // p1 = T1 lp1 + delta ; p2 = T2 lp2
        V3 p2 = dppv<DPP_FROM_ODD>(pw);
        float dist = p.d0 + dot(pw - p2, h.dir);
        float rhs_wo_bias = rp_max(dist, 0.0f) * inv_dt;
        float rhs_bias = rp_clamp(dist * h.erp_inv_dt, -maxcv, 0.0f);
        p.rhsR = rhs_wo_bias;
        p.rhsB = rhs_wo_bias + rhs_bias;
        p.cfmB = dist <= 0.0f ? h.cfm_factor : 1.0f;
    }
    V3 pf = xf_tp(x, v3((h.odd ? L.B1 : L.B0)[m]));
    pf = sel(h.odd, pf, pf + tangent_delta);
    V3 pf2 = dppv<DPP_FROM_ODD>(pf);
    h.tb0 = dot(pf - pf2, h.t0) * inv_dt; h.tb1 = dot(pf - pf2, h.t1) * inv_dt;
}

// Body-centric warm start (update + warmstart, :426-522 and :633-678).  The impulses a warm start applies do not depend on velocities, only the
// order in which they are ADDED to a body does (colour order, and inside a manifold: the points, the
// tangent part, the twist part).  Every lane therefore writes its terms to LDS in one parallel stage
// (isl_ws_terms) and the thread that owns a body adds them in exactly that order (isl_ws_accumulate):
// 2 stages per substep instead of one per colour, same additions, same order, same bits.
#define WS_SLOTS 11   // per lane: 4 x (lin, ang) point terms, tangent lin, tangent ang, twist ang
#define WS_SLOTS_2PHASE 6 // the two-phase form (dense variant of the kernel): the linear terms (5 slots), then — same rows — the angular ones (6)
#define WS_STRIDE (ISL_LANES + 2) // rows of one slot: every lane's row + two scratch rows for world-attached sides
#include "rp_coulomb_pair.h"      // FrictionModel::Coulomb by the lane pair: 16 dense rows (its own generate / pose stage / terms / solve / write-back)
// every term at once (11 slots); the register-lean form of the kernel writes the linear and the angular terms in two phases into 6 slots
// (lean_ws_terms, rp_islands_lean.h: 31 KB of LDS instead of 57 KB, which is what lets two islands share a CU)
// DENSE: every one of the 11 rows is written — the rows of points a manifold does not have (and the twist row of a one-point manifold)
// hold -0.0, the one float whose addition changes no bit of any accumulator (x + -0.0 == x for every x, +0.0 and -0.0 included) — so
// that the body threads add a toucher's rows without reading its point count: no compare / select on their dependent chain
// (isl_ws_accumulate_*_dense; round 6).
template <bool DENSE = false>
RP_DEV void isl_ws_terms(const DevWorld &w, IslSide &h, float4 *W, int t) { // t = this lane's row in W (its rank in its body's list)
    float wc = w.prm.p.warmstart_coefficient;
    bool ws = wc != 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= h.n) {
            if constexpr (DENSE) { if (ws) { const float4 z = make_float4(-0.0f, -0.0f, -0.0f, 0.0f); for (int j = k; j < 4; ++j) { W[(2 * j) * WS_STRIDE + t] = z; W[(2 * j + 1) * WS_STRIDE + t] = z; } } }
            break;
        }
        SidePoint &p = h.P[k];
        p.rhs = p.rhsB; p.cfm = p.cfmB;
        p.acc += p.lam;
        p.lam *= wc;
        if (ws) {
            float lam = dppf<DPP_FROM_EVEN>(p.lam);
            W[(2 * k) * WS_STRIDE + t] = f4(h.sdim * lam, 0.0f); W[(2 * k + 1) * WS_STRIDE + t] = f4(p.pc * lam, 0.0f);
        }
    }
    h.t_rhs0 = h.rhs_wo0 + h.tb0; h.t_rhs1 = h.rhs_wo1 + h.tb1;
    h.t_acc0 += h.t_imp0; h.t_acc1 += h.t_imp1;
    h.t_imp0 *= wc; h.t_imp1 *= wc;
    h.tw_acc += h.tw_imp;
    h.tw_imp *= wc;
    if (ws) {
        float i0 = dppf<DPP_FROM_EVEN>(h.t_imp0), i1 = dppf<DPP_FROM_EVEN>(h.t_imp1);
        float s0 = h.odd ? -i0 : i0, s1 = h.odd ? -i1 : i1;
        float tw = dppf<DPP_FROM_EVEN>(h.tw_imp);
        W[8 * WS_STRIDE + t] = f4(cmul(h.t0 * s0 + h.t1 * s1, h.im), __int_as_float(h.n));
        W[9 * WS_STRIDE + t] = f4(h.itd0 * i0 + h.itd1 * i1, 0.0f);
        if (h.n > 1) W[10 * WS_STRIDE + t] = f4(h.stw * tw, 0.0f);
        else if constexpr (DENSE) W[10 * WS_STRIDE + t] = make_float4(-0.0f, -0.0f, -0.0f, 0.0f);
    }
}
// the body threads' side of the dense rows: the same additions in the same order, minus the ones of -0.0 rows that change nothing
RP_DEV void isl_ws_accumulate_lin_dense(const float4 *W, int begin, int count, V3 &lin) {
#pragma unroll 2
    for (int e = 0; e < count; ++e) {
        const int row = begin + e;
        const float4 l0 = W[0 * WS_STRIDE + row], l1 = W[2 * WS_STRIDE + row], l2 = W[4 * WS_STRIDE + row], l3 = W[6 * WS_STRIDE + row], tl = W[8 * WS_STRIDE + row];
        lin = lin + v3(l0); lin = lin + v3(l1); lin = lin + v3(l2); lin = lin + v3(l3); lin = lin + v3(tl);
    }
}
RP_DEV void isl_ws_accumulate_ang_dense(const float4 *W, int begin, int count, V3 &ang) {
#pragma unroll 2
    for (int e = 0; e < count; ++e) {
        const int row = begin + e;
        const float4 a0 = W[1 * WS_STRIDE + row], a1 = W[3 * WS_STRIDE + row], a2 = W[5 * WS_STRIDE + row], a3 = W[7 * WS_STRIDE + row], ta = W[9 * WS_STRIDE + row], tw = W[10 * WS_STRIDE + row];
        ang = ang + v3(a0); ang = ang + v3(a1); ang = ang + v3(a2); ang = ang + v3(a3); ang = ang + v3(ta); ang = ang + v3(tw);
    }
}
// one thread adds the linear terms of a body, another one (64 lanes further) its angular terms
template <bool TWO>
RP_DEV void isl_ws_accumulate_lin(const float4 *W, int begin, int count, V3 &lin) {
#pragma unroll 2
    for (int e = 0; e < count; ++e) {
        const int row = begin + e;
        float4 tl = W[(TWO ? 4 : 8) * WS_STRIDE + row];
        float4 l0 = W[0 * WS_STRIDE + row], l1 = W[(TWO ? 1 : 2) * WS_STRIDE + row], l2 = W[(TWO ? 2 : 4) * WS_STRIDE + row], l3 = W[(TWO ? 3 : 6) * WS_STRIDE + row];
        const int n = __float_as_int(tl.w);
        lin = lin + v3(l0);
        if (n > 1) lin = lin + v3(l1);
        if (n > 2) lin = lin + v3(l2);
        if (n > 3) lin = lin + v3(l3);
        lin = lin + v3(tl);
    }
}
template <bool TWO>
RP_DEV void isl_ws_accumulate_ang(const float4 *W, int begin, int count, V3 &ang) {
#pragma unroll 2
    for (int e = 0; e < count; ++e) {
        const int row = begin + e;
        float4 tl = W[(TWO ? 4 : 8) * WS_STRIDE + row], ta = TWO ? tl : W[9 * WS_STRIDE + row], tw = W[(TWO ? 5 : 10) * WS_STRIDE + row];
        float4 a0 = W[(TWO ? 0 : 1) * WS_STRIDE + row], a1 = W[(TWO ? 1 : 3) * WS_STRIDE + row], a2 = W[(TWO ? 2 : 5) * WS_STRIDE + row], a3 = W[(TWO ? 3 : 7) * WS_STRIDE + row];
        const int n = __float_as_int(tl.w);
        ang = ang + v3(a0);
        if (n > 1) ang = ang + v3(a1);
        if (n > 2) ang = ang + v3(a2);
        if (n > 3) ang = ang + v3(a3);
        ang = ang + v3(ta);
        if (n > 1) ang = ang + v3(tw);
    }
}

// apply_restitution (:568-597; the Coulomb model's is the same sweep over its normal parts, contact_with_coulomb_friction.rs:520-559)
template <class Side> RP_DEV void isl_restitution(Side &h, const IslLds &L) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (k >= h.n) break; any |= dppf<DPP_FROM_EVEN>(h.P[k].seed) < 0.0f; }
    if (!any) return;
    Vel v = isl_vel(L, h.id);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= h.n) break;
        auto &p = h.P[k];
        float X = dot(h.dir, v.lin), Y = dot(p.pa, v.ang);
        float S = X + Y;
        float dvel = S - dppf<DPP_FROM_ODD>(X) + dppf<DPP_FROM_ODD>(Y) + p.seed;
        bool gate = p.seed < 0.0f && (p.acc + p.lam) > 0.0f;
        float new_impulse = gate ? rp_max(p.lam - p.r * dvel, 0.0f) : p.lam;
        float dl = dppf<DPP_FROM_EVEN>(new_impulse - p.lam);
        p.lam = new_impulse;
        v.lin = v.lin + h.sdim * dl;
        v.ang = v.ang + p.pc * dl;
    }
    isl_set_vel(L, h.id, v);
}

// writeback_impulses (:783-829) — even lane
RP_DEV void isl_writeback(const DevWorld &w, const IslSide &h, int s) {
    V3 wtw = h.t0 * rp_canon0(h.t_imp0) + h.t1 * rp_canon0(h.t_imp1); // canonicalised zeros, as in cons_writeback
    wtw = v3(rp_canon0(wtw.x), rp_canon0(wtw.y), rp_canon0(wtw.z));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= h.n) break;
        int cid = (h.cids >> (8 * k)) & 0xff;
        PT(w.pt_imp, cid, s) = make_float4(rp_canon0(h.P[k].acc + h.P[k].lam), rp_canon0(h.P[k].lam), rp_canon0(h.tw_imp), 0.0f);
        PT(w.pt_wst, cid, s) = f4(wtw, 0.0f);
    }
}

#ifdef RP_ISL_PROFILE
#define ISL_STAMP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) w.dbg[slot] += (long long)__builtin_readcyclecounter() - t_prev, t_prev = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define ISL_STAMP(slot) do { } while (0)
#endif

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
        // ties broken by the (collider1, collider2) key so the serial overflow order depends neither on atomics nor on the order in
        // which the broad phase handed out the pair slots (the oracle sweeps its overflow colour in the same order)
        const unsigned long long key = pair_order_key(w, s);
        for (int j = 0; j < nc; ++j) {
            int rj = T_rank[j];
            if (rj < r) posn++;
            else if (rj == r && j != t) { int sj = T_slot[j]; unsigned long long kj = pair_order_key(w, sj); posn += kj < key; }
        }
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
            T_rank[i] = q;
        }
        w.isl_nstages[isl] = q + 1;
    }
    __syncthreads();
    // Per sorted manifold: the solver-attached body of each side (arena and island-local index);
    // per body: the lanes (2m + side) that touch it, in sweep order — the body-centric warm start
    // walks this list.  I_body / I_cnt reuse the scratch arrays.
    int *I_body0 = T_slot, *I_body1 = K_rank, *I_cnt = K_slot; // T_rank keeps the stage of manifold m
    if (t < RP_ISL_NB_MAX) I_cnt[t] = 0;
    __syncthreads();
    if (t < nc) {
        int slot = w.isl_cons[cb + t];
        int rb1 = w.c_parent[w.p_c1[slot]], rb2 = w.c_parent[w.p_c2[slot]];
        int rel_dom = w.p_reldom[slot];
        int g1 = (is_dyn(w, rb1) && rel_dom <= 0) ? rb1 : -1;
        int g2 = (is_dyn(w, rb2) && rel_dom >= 0) ? rb2 : -1;
        int l1 = g1 >= 0 ? w.b_local[g1] : -1, l2 = g2 >= 0 ? w.b_local[g2] : -1;
        w.isl_cg1[cb + t] = g1; w.isl_cg2[cb + t] = g2; w.isl_cl1[cb + t] = l1; w.isl_cl2[cb + t] = l2;
        I_body0[t] = l1; I_body1[t] = l2;
        if (l1 >= 0) atomicAdd(&I_cnt[l1], 1);
        if (l2 >= 0) atomicAdd(&I_cnt[l2], 1);
    }
    __syncthreads();
    const int bb = w.isl_body_begin[isl], nb = w.isl_nb[isl];
    if (t == 0) { int pos = 0; for (int b = 0; b < nb; ++b) { w.isl_inc_begin[bb + b] = pos; w.isl_inc_cnt[bb + b] = I_cnt[b]; pos += I_cnt[b]; } }
    __threadfence(); __syncthreads();
    if (t < 2 * nc) {
        int m = t >> 1, side = t & 1;
        int b = side ? I_body1[m] : I_body0[m];
        if (b >= 0) {
            int q = T_rank[m], rank = 0;
            for (int m2 = 0; m2 < nc; ++m2) { int q2 = T_rank[m2]; rank += (q2 < q) && (I_body0[m2] == b || I_body1[m2] == b); }
            w.isl_inc_pos[2 * cb + t] = w.isl_inc_begin[bb + b] + rank;
        } else w.isl_inc_pos[2 * cb + t] = ISL_LANES + (t & 1); // world-attached side: scratch rows nobody reads
    }
    __threadfence(); __syncthreads();
    if (t == 0) w.isl_sorted[isl] = 1;
    __syncthreads();
}

// ---- the fused step's validation (k_island_solve / k_island_solve_dense with `fused`): does island `isl` need the broad phase, the
// narrow phase or the sleep commit this step?  Lane vt of vn validating lanes takes every vn-th item: a body (EVERY collider of it —
// compound bodies chain theirs through c_sibling — against its fat AABB, and in a sleep-enabled world the body's sleep observation), an
// active pair, a pair without solver contacts (recycle test; in a sleep-enabled world also the hint a body that fell asleep cleared).
// `slp` collects sleep_observe_fused's bits over the lane's bodies; fused_sleep_abort() reads their OR over the whole island.
// WIDE = false: the form for worlds of one-collider bodies that never sleep (the benchmark scenes): exactly one collider and one test
// per item, nothing else in the validators' dependent chain (A/B on C3: the wide form costs 5 us of a 73 us step).
// `part`: 0 = every item; 1 / 2 = the two halves of a split at `total - tail` items: part 1 the items in front of it, part 2 the last `tail`
// ones (k_island_solve, round 6: the wavefronts that also hold the body roles take one item per lane behind their body loads, the
// wavefronts that only validate take the rest and start with the kernel).
// `cross` (a launch of several steps, narrow form): set when an item names a body of ANOTHER island (a pair without solver contacts is
// listed with the island of its first dynamic body): such an island may not take a second step inside the launch.
template <bool WIDE> RP_DEV bool fused_validate_island(const DevWorld &w, int isl, int vt, int vn, int stamp_before, int &slp, int part = 0, int tail = 0, bool *cross = nullptr) {
    const int nb = w.isl_nb[isl], nc = w.isl_nc[isl], ni = w.isl_ni[isl];
    const int bb = w.isl_body_begin[isl], cb = w.isl_cons_begin[isl], ib = w.isl_icons_begin[isl];
    const int ns = (WIDE && w.sleep_enabled) ? nb : 0; // the sleep observations are items of their own: other lanes than the fat-AABB tests of the same bodies
    bool bad = false;
    const int total = nb + ns + nc + ni, split = total > tail ? total - tail : 0;
    const int i_begin = part == 2 ? split : 0, i_end = part == 1 ? split : total;
    if constexpr (!WIDE) {
        // the benchmark form: pairs first, bodies behind them (a lane that gets two items gets the cheaper kind second), flat loads
        for (int i = i_begin + vt; i < i_end; i += vn) {
            if (i < nc + ni) {
                const int s = i < nc ? w.isl_cons[cb + i] : w.isl_icons[ib + i - nc];
                if (pair_needs_narrow_phase_flat(w, s, cross ? isl : -1, cross)) bad = true;
            } else {
                const int b = w.isl_bodies[bb + i - nc - ni];
                const int c = w.b_collider[b];
                if (c >= 0 && collider_left_fat_aabb_flat(w, c, b)) bad = true;
            }
        }
        return bad;
    }
    for (int i = i_begin + vt; i < i_end; i += vn) {
        if (i < nb) {
            const int b = w.isl_bodies[bb + i];
            if constexpr (WIDE) {
                for (int c = w.b_collider[b]; c >= 0; c = w.c_sibling[c]) if (collider_left_fat_aabb(w, c)) bad = true;
            } else {
                const int c = w.b_collider[b];
                if (c >= 0 && collider_left_fat_aabb(w, c)) bad = true;
            }
        } else if (i < nb + ns) {
            if constexpr (WIDE) { const int o = sleep_observe_fused(w, w.isl_bodies[bb + i - nb], stamp_before); slp |= o; if (o & 4) bad = true; }
        } else {
            const int k = i - nb - ns;
            const int s = k < nc ? w.isl_cons[cb + k] : w.isl_icons[ib + k - nc];
            if constexpr (WIDE) if (w.sleep_enabled && w.p_c1[s] >= 0 && pair_hint_cleared(w, s, w.p_rb[s])) bad = true; // (k_fast_front's test: the narrow phase recomputes a count-cleared hint)
            if (pair_needs_narrow_phase(w, s)) bad = true;
        }
    }
    return bad;
}
// an island none of whose members keeps its persistent island awake may fall asleep in this step's commit: that is the full graph's
RP_DEV bool fused_sleep_abort(int slp_all) { return (slp_all & 1) && !(slp_all & 2); }
// what k_fast_front checks for the whole world before a fast step (workgroup 0)
template <bool WIDE> RP_DEV bool fused_world_abort(const DevWorld &w) {
    if (w.flags[FL_BP_DIRTY] || w.flags[FL_N_CONS] > 0 || w.flags[FL_N_GLOB_BODIES] > 0 || w.n_joints > 0) return true;
    if constexpr (!WIDE) return false;
    // sleep-enabled worlds: the awake set must be settled — no layout change, wake-up request or island bookkeeping waiting for a full step
    return w.sleep_enabled && (w.flags[FL_LAYOUT_DIRTY] || w.flags[FL_WAKE_PENDING] || w.flags[FL_N_AWAKE] == 0 || w.flags[FL_PI_PENDING] || w.flags[FL_PJ_COUNT] || w.flags[FL_PI_JLINK]);
}
