// rp_global.h — the HBM-resident (global) solver path shared by rp_solver.hip (MULTI mode: one launch
// per colour stage) and rp_islands.hip (SINGLE mode: one extra workgroup of the island launch).
#pragma once
#include "rp_coulomb.h"
#include "rp_pairs.h"
#include "rp_joints.h"

RP_DEV bool global_body(const DevWorld &w, int i) {
    return flags_active(w.b_flags[i]) && w.b_island[i] < 0;
}

// ---- per-body device steps over the HBM solver-body arrays --------------------------------------
RP_DEV void g_body_begin(const DevWorld &w, int i) {
    V3 lin, ang, trans, incl, inca; Q4 rot;
    body_begin(w, i, lin, ang, rot, trans, incl, inca);
    w.s_inca[i] = f4(inca, 0.0f); w.s_incl[i] = f4(incl, 0.0f);
    w.s_lin[i] = f4(lin, 0.0f); w.s_ang[i] = f4(ang, 0.0f);
    w.s_rot[i] = f4(rot); w.s_trans[i] = f4(trans, 0.0f);
}
RP_DEV void g_body_increment(const DevWorld &w, int i) {
    V3 lin = v3(w.s_lin[i]), ang = v3(w.s_ang[i]);
    body_increment(w, w.b_flags[i], lin, ang, q4(w.s_rot[i]), v3(w.s_incl[i]), v3(w.s_inca[i]), v3(w.b_invpi[i]), q4(w.b_pframe[i]));
    w.s_lin[i] = f4(lin, 0.0f); w.s_ang[i] = f4(ang, 0.0f);
}
RP_DEV void g_body_integrate(const DevWorld &w, int i) {
    V3 lin = v3(w.s_lin[i]), ang = v3(w.s_ang[i]), trans = v3(w.s_trans[i]); Q4 rot = q4(w.s_rot[i]);
    body_integrate(w, w.b_flags[i], lin, ang, rot, trans);
    w.s_lin[i] = f4(lin, 0.0f); w.s_ang[i] = f4(ang, 0.0f); w.s_rot[i] = f4(rot); w.s_trans[i] = f4(trans, 0.0f);
}
// kinematic bodies (worker.rs:826-842): damped velocity; a velocity-based body takes the integrated solver pose, a
// position-based one lands exactly on the pose the user asked for; their inverse mass / inertia stay zero.
// Kept out of line: kinematic bodies are rare and the call keeps the dynamic path's register / scratch budget unchanged.
// (the arrays are handed over as plain pointers: a DevWorld reference in this out-of-line path makes the compiler keep a
// 1.4 KB stack copy of the kernel argument, and that much scratch costs ~40 us of launch latency on every step)
struct KinWb { float4 *b_damp, *s_lin, *s_ang, *s_rot, *s_trans, *b_lcom_invm, *b_next_rot, *b_next_pos, *b_linvel, *b_angvel, *b_pos, *b_rot, *b_wcom; int *flags, *b_quar; float dt; };
__device__ __noinline__ void g_kinematic_writeback(KinWb k, int i, int type) {
    float4 damp = k.b_damp[i];
    float dt = k.dt;
    V3 lin = v3(k.s_lin[i]) * (1.0f / (1.0f + dt * damp.x));
    V3 ang = v3(k.s_ang[i]) * (1.0f / (1.0f + dt * damp.y));
    V3 lcom = v3(k.b_lcom_invm[i]);
    Q4 rot = q4(k.s_rot[i]);
    V3 t = v3(k.s_trans[i]) + qrot(rot, -lcom);
    if (type == RP_BODY_KINEMATIC_POSITION) { rot = q4(k.b_next_rot[i]); t = v3(k.b_next_pos[i]); }
    bool finite = isfinite(t.x) && isfinite(t.y) && isfinite(t.z) && isfinite(rot.x) && isfinite(rot.y) && isfinite(rot.z) && isfinite(rot.w) &&
                  isfinite(lin.x) && isfinite(lin.y) && isfinite(lin.z) && isfinite(ang.x) && isfinite(ang.y) && isfinite(ang.z);
    if (!finite) { atomicAdd(&k.flags[FL_QUARANTINE], 1); k.b_quar[i] = 1; k.b_linvel[i] = make_float4(0, 0, 0, 0); k.b_angvel[i] = make_float4(0, 0, 0, 0); return; }
    k.b_linvel[i] = f4(lin, 0.0f); k.b_angvel[i] = f4(ang, 0.0f);
    k.b_pos[i] = f4(t, 0.0f); k.b_rot[i] = f4(rot);
    k.b_next_pos[i] = f4(t, 0.0f); k.b_next_rot[i] = f4(rot);
    k.b_wcom[i] = f4(qrot(rot, lcom) + t, 0.0f);
}
RP_DEV void g_body_writeback(const DevWorld &w, int i) {
    int type = w.b_flags[i] & RP_BF_TYPE_MASK;
    if (type == RP_BODY_DYNAMIC) body_writeback(w, i, v3(w.s_lin[i]), v3(w.s_ang[i]), q4(w.s_rot[i]), v3(w.s_trans[i]));
    else { KinWb k = {w.b_damp, w.s_lin, w.s_ang, w.s_rot, w.s_trans, w.b_lcom_invm, w.b_next_rot, w.b_next_pos, w.b_linvel, w.b_angvel, w.b_pos, w.b_rot, w.b_wcom, w.flags, w.b_quar, w.prm.p.dt}; g_kinematic_writeback(k, i, type); }
}
template <bool COUL, bool PRE = false>
RP_DEV bool g_generate(const DevWorld &w, int pos) {
    int s = w.cons_pair[pos];
    int rb1 = w.c_parent[w.p_c1[s]], rb2 = w.c_parent[w.p_c2[s]];
    int rel_dom = w.p_reldom[s];
    bool dyn1 = body_active(w, rb1), dyn2 = body_active(w, rb2); // solver bodies = the active set (solver_body.rs:114)
    int id1 = (dyn1 && rel_dom <= 0) ? rb1 : -1;
    int id2 = (dyn2 && rel_dom >= 0) ? rb2 : -1;
    if (COUL) return coul_generate(w, GlobalAccT<PRE>(w, pos), s, id1, id2, id1, id2);
    return cons_generate(w, GlobalAccT<PRE>(w, pos), s, id1, id2, id1, id2);
}

// the solver bodies g_generate will attach to the manifold at `pos` (toucher lists of rp_flow.hip, tiling of rp_tiles.hip)
RP_DEV void flow_ids(const DevWorld &w, int pos, int &id1, int &id2) {
    int s = w.cons_pair[pos];
    int rb1 = w.c_parent[w.p_c1[s]], rb2 = w.c_parent[w.p_c2[s]];
    int rel_dom = w.p_reldom[s];
    id1 = (body_active(w, rb1) && rel_dom <= 0) ? rb1 : -1;
    id2 = (body_active(w, rb2) && rel_dom >= 0) ? rb2 : -1;
}

// ---- body-centric warm start (rp_solver.hip: k_ws_prepare writes the terms, k_increment_ws / the tile sweeps add them) ----
#define WS_TERMS 11
// A 16-byte load that reads past this XCD's L2 (sc1): the reader's half of "sc1 stores AND sc1 loads" (MI355X guide, inter-workgroup
// visibility) — what another workgroup of the SAME launch stored write-through is read from the memory side without an agent-scope
// acquire, i.e. without invalidating the L2 every tile of the XCD shares (k_tile_step, rp_tiles.hip).  A buffer load, so that the compiler
// counts it (a hand-issued global_load is invisible to its waitcnt pass); `base` is wave-uniform, `idx` per lane.
RP_DEV float4 ld16_sc1(const float4 *base, unsigned idx) {
    typedef unsigned ld_v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0xffffffff, 0x00027000);
    const ld_v4u x = __builtin_amdgcn_raw_buffer_load_b128(r, idx * 16u, 0, 16); // (aux bit 4 = sc1 on gfx94x / gfx950)
    return make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
}
template <bool SC1LD> RP_DEV float4 ld16_t(const float4 *base, unsigned idx) { return SC1LD ? ld16_sc1(base, idx) : base[idx]; }
// ... and the writer's half: a 16-byte write-through store (sc1: nothing of it stays dirty in the XCD's L2), as a buffer store the
// compiler schedules and counts (the hand-issued form — asm volatile + "memory" — kept every store in program order between the loads)
RP_DEV void st16_sc1(float4 *base, unsigned idx, float4 v) {
    typedef unsigned st_v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0xffffffff, 0x00027000);
    const st_v4u x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(x, r, idx * 16u, 0, 16);
}
// ws_terms = [11][2 * cons_cap] planes indexed by 2 * position + side: the writes of k_ws_prepare are coalesced plane by plane (a
// per-body layout, one contiguous run of terms per body, was measured: the scattered 176-byte writes doubled k_ws_prepare and
// bought the accumulation nothing).  `row` = 2 * pos + side, -1 for a world-attached side.
RP_DEV void ws_put(const DevWorld &w, int slot, int row, V3 v) { if (row >= 0) w.ws_terms[(size_t)slot * (2 * (size_t)w.cons_cap) + row] = f4(v, 0.0f); }
// (SC1: the write-through store of a launch that hands the terms to other workgroups behind a flag instead of a kernel boundary — k_tile_step)
template <bool SC1>
RP_DEV void ws_put_t(const DevWorld &w, int slot, int row, V3 v) {
    if (row < 0) return;
    if (SC1) st16_sc1(w.ws_terms + (size_t)slot * (2 * (size_t)w.cons_cap), (unsigned)row, f4(v, 0.0f));
    else w.ws_terms[(size_t)slot * (2 * (size_t)w.cons_cap) + row] = f4(v, 0.0f);
}
RP_DEV V3 ws_get(const DevWorld &w, int slot, int row) { return v3(w.ws_terms[(size_t)slot * (2 * (size_t)w.cons_cap) + row]); }
// update (contact_with_twist_friction.rs:426-522) of one manifold + the velocity terms its warm start adds to either body, for the
// body-centric warm start: the right-hand sides and cfm of the coming substep from the current poses, the banked impulses, the 11 terms
// of either side into ws_terms (row 2 * pos + side).  Called by k_ws_prepare for every manifold, and by the owner instance of a
// manifold at the end of its RELAXED solve on LDS tiles (rp_tiles.hip): its rows are in registers there and the poses are the ones the
// next substep's prepare would read — three launches and a pass over every row saved per step.
template <class Acc, bool SC1 = false>
RP_DEV void ws_prepare_one(const DevWorld &w, const Acc &A, int pos, float solved_dt) {
    const int id1 = A.id1(), id2 = A.id2(), n = A.n();
    const bool is_static = id1 < 0 || id2 < 0;
    const float fstatic = is_static ? 1.0f : 0.0f;
    const float cfm_factor = w.prm.dyn_cfm + fstatic * (w.prm.static_cfm - w.prm.dyn_cfm);
    const float erp_inv_dt = w.prm.dyn_erp_inv_dt + fstatic * (w.prm.static_erp_inv_dt - w.prm.dyn_erp_inv_dt);
    const float inv_dt = w.prm.inv_dt_sub, maxcv = w.prm.max_corrective_velocity, wc = w.prm.p.warmstart_coefficient;
    const Xf x1 = A.xf(id1), x2 = A.xf(id2);
    const float4 h0 = A.ld(CP_H0), h6 = A.ld(CP_H6);
    const V3 dir1 = v3(h0), t0 = v3(h6), t1 = cross(dir1, t0);
    const V3 tangent_delta = v3(A.ld(CP_B2)) * solved_dt;
    const V3 im1 = v3(A.ld(CP_H1)), im2 = v3(A.ld(CP_H2));
    const bool ws = wc != 0.0f;
    const int s1 = id1 >= 0 ? 2 * pos : -1, s2 = id2 >= 0 ? 2 * pos + 1 : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= n) break;
        float4 m = A.ld(NPL(k, NP_M));
        float4 c = A.ld(NPL(k, NP_C)), d = A.ld(NPL(k, NP_D));
        V3 p1 = xf_tp(x1, v3(A.ld(NPL(k, NP_E)))) + tangent_delta;
        V3 p2 = xf_tp(x2, v3(A.ld(NPL(k, NP_F))));
        float dist = c.w + dot(p1 - p2, dir1);
        float rhs_wo_bias = rp_max(dist, 0.0f) * inv_dt;
        float rhs_bias = rp_clamp(dist * erp_inv_dt, -maxcv, 0.0f);
        m.x = rhs_wo_bias + rhs_bias;
        m.y = dist <= 0.0f ? cfm_factor : 1.0f;
        m.w += m.z;
        m.z *= wc;
        A.st(NPL(k, NP_M), m);
        if (ws) { // ContactConstraintNormalPartSlim::warmstart, contact_constraint_element.rs:465-478
            ws_put_t<SC1>(w, k, s1, cmul(dir1, im1) * m.z); ws_put_t<SC1>(w, 5 + k, s1, v3(c) * m.z);
            ws_put_t<SC1>(w, k, s2, cmul(dir1, im2) * (-m.z)); ws_put_t<SC1>(w, 5 + k, s2, v3(d) * m.z);
        }
    }
    float4 hm0 = A.ld(CP_HM0), hm1 = A.ld(CP_HM1), h7 = A.ld(CP_H7);
    {
        V3 p1 = xf_tp(x1, v3(A.ld(CP_B0))) + tangent_delta;
        V3 p2 = xf_tp(x2, v3(A.ld(CP_B1)));
        float bias0 = dot(p1 - p2, t0) * inv_dt, bias1 = dot(p1 - p2, t1) * inv_dt;
        hm1.z = h6.w + bias0; hm1.w = h7.x + bias1;
        hm1.x += hm0.z; hm1.y += hm0.w;
        hm0.z *= wc; hm0.w *= wc;
        hm0.y += hm0.x;
        hm0.x *= wc;
    }
    A.st(CP_HM0, hm0); A.st(CP_HM1, hm1);
    if (ws) {
        const float i0 = hm0.z, i1 = hm0.w;
        ws_put_t<SC1>(w, 4, s1, cmul(t0 * i0 + t1 * i1, im1)); ws_put_t<SC1>(w, 9, s1, v3(A.ld(CP_T4)) * i0 + v3(A.ld(CP_T5)) * i1);
        ws_put_t<SC1>(w, 4, s2, cmul(t0 * (-i0) + t1 * (-i1), im2)); ws_put_t<SC1>(w, 9, s2, v3(A.ld(CP_T6)) * i0 + v3(A.ld(CP_T7)) * i1);
        if (n > 1) {
            float4 h3 = A.ld(CP_H3), h4 = A.ld(CP_H4), h5 = A.ld(CP_H5);
            Sym3 ii1 = {h3.x, h3.y, h3.z, h3.w, h4.x, h4.y}, ii2 = {h4.z, h4.w, h5.x, h5.y, h5.z, h5.w};
            ws_put_t<SC1>(w, 10, s1, sym_mul(ii1, dir1) * hm0.x);
            ws_put_t<SC1>(w, 10, s2, -(sym_mul(ii2, dir1) * hm0.x)); // v2.ang - y == v2.ang + (-y), exactly
        }
    }
}

// S2 + the warm start of one global-path body: increment (+ gyroscopic term), then its touchers' terms in sweep order (f_sorted holds the
// term row 2 * position + side of every toucher), each added exactly as the colour sweep would have added it (same operands, same order
// per accumulator).  The loads are batched — every row index of up to eight touchers, then their point counts and the terms two
// touchers at a time (all eleven terms of a side: the slots of unused points hold stale values that are fetched and ignored) — so the
// chain is ~5 round trips instead of three per toucher (measured as the prologue of the tile sweeps: 24 us before).
template <bool SC1LD = false>
RP_DEV void ws_fetch(const DevWorld &w, int row, float4 (&tm)[WS_TERMS]) {
#pragma unroll
    for (int s = 0; s < WS_TERMS; ++s) tm[s] = ld16_t<SC1LD>(w.ws_terms + (size_t)s * (2 * (size_t)w.cons_cap), (unsigned)row);
}
RP_DEV void ws_accumulate(V3 &lin, V3 &ang, const float4 (&tm)[WS_TERMS], int n) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (k >= n) break; lin = lin + v3(tm[k]); ang = ang + v3(tm[5 + k]); }
    lin = lin + v3(tm[4]);
    ang = ang + v3(tm[9]);
    if (n > 1) ang = ang + v3(tm[10]);
}
// (vs / as / rs: the copies of the solver velocities and rotations that are current — the tile launches alternate between two)
template <bool SC1LD = false>
RP_DEV void body_increment_ws_at(const DevWorld &w, int i, const float4 *vs, const float4 *as, const float4 *rs, V3 &lin, V3 &ang) {
    lin = v3(ld16_t<SC1LD>(vs, (unsigned)i)); ang = v3(ld16_t<SC1LD>(as, (unsigned)i));
    const int2 beg2 = w.fb_begin[i], deg2 = w.fb_deg[i];
    body_increment(w, w.b_flags[i], lin, ang, q4(ld16_t<SC1LD>(rs, (unsigned)i)), v3(w.s_incl[i]), v3(w.s_inca[i]), v3(w.b_invpi[i]), q4(w.b_pframe[i]));
    if (w.prm.p.warmstart_coefficient == 0.0f) return;
    const int beg = beg2.x, deg = deg2.x;
    for (int r0 = 0; r0 < deg; r0 += 8) { // this body's constraints in sweep order, eight at a time
        int rows[8], ns[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) rows[j] = r0 + j < deg ? w.f_sorted[beg + r0 + j] : -1;
#pragma unroll
        for (int j = 0; j < 8; ++j) ns[j] = rows[j] >= 0 ? w.k_n[rows[j] >> 1] : 0;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            if (rows[j] < 0) break;
            float4 ta[WS_TERMS], tb[WS_TERMS];
            ws_fetch<SC1LD>(w, rows[j], ta);
            if (rows[j + 1] >= 0) ws_fetch<SC1LD>(w, rows[j + 1], tb);
            ws_accumulate(lin, ang, ta, ns[j]);
            if (rows[j + 1] >= 0) ws_accumulate(lin, ang, tb, ns[j + 1]);
        }
    }
}
RP_DEV void body_increment_ws(const DevWorld &w, int i, V3 &lin, V3 &ang) { body_increment_ws_at(w, i, w.s_lin, w.s_ang, w.s_rot, lin, ang); }

// Serial tail of one sweep (worker 0 of the reference): stages [first, n_stages) one after the other
// inside one workgroup, then the overflow colour on lane 0.
template <int MODE, bool COUL, bool PRE = false>
RP_DEV void tail_sweep(const DevWorld &w, int first, bool fib, float solved_dt) {
    int nst = w.flags[FL_N_STAGES];
    for (int st = first; st < nst; ++st) {
        int beg = w.stage_begin[st], cnt = w.stage_count[st];
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) cons_apply_model<COUL>(w, GlobalAccT<PRE>(w, beg + i), MODE, fib, solved_dt);
        __threadfence();
        __syncthreads();
    }
    if (w.flags[FL_HAS_OVERFLOW_COLOR]) {
        const int beg = w.stage_begin[nst], cnt = w.stage_count[nst];
        if (w.lay_state[5] && cnt > 8) {
            // every overflow manifold has one dynamic side (lay_rank_overflow, rp_islands.hip): each thread sweeps, in key order, the
            // manifolds of the bodies it owns — what the serial sweep computes, since manifolds of different owners share no written side
            const unsigned nt = blockDim.x;
            for (int i = 0; i < cnt; ++i) if ((unsigned)w.ov_owner[i] % nt == threadIdx.x) cons_apply_model<COUL>(w, GlobalAccT<PRE>(w, beg + i), MODE, fib, solved_dt);
        } else if (threadIdx.x == 0) {
            for (int i = 0; i < cnt; ++i) { cons_apply_model<COUL>(w, GlobalAccT<PRE>(w, beg + i), MODE, fib, solved_dt); __threadfence(); }
        }
        __threadfence();
        __syncthreads();
    }
}
// ---- SINGLE mode: the whole global path in ONE workgroup (k_global_single); the step is retired
// (FL_SEQ / FL_STEP + hint publication) by workgroup 0 of k_island_solve.
// Publish the device scalars to the host-mapped hint buffer (posted PCIe writes; the host only ever
// uses them as hints, or re-reads them after a stream sync).
RP_DEV void publish_flags(const DevWorld &w) {
    for (int k = threadIdx.x; k < FL_COUNT; k += blockDim.x) {
        int v = __hip_atomic_load(&w.flags[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&w.host_flags[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
template <bool COUL>
RP_DEV void global_single_block(const DevWorld &w, int has_restitution, int fast) {
    const int t = threadIdx.x, nt = blockDim.x;
    __shared__ int bouncy;
    if (fast && w.flags[FL_FAST_ABORT]) return; // fast graph gave up: the step is replayed by the full graph
    if (t == 0) bouncy = 0;
    int M = w.flags[FL_N_CONS];
    if (M > w.cons_cap) M = w.cons_cap;
    int ngb = w.flags[FL_N_GLOB_BODIES];
    const int nj = w.n_joints;
    if (M == 0 && ngb == 0 && nj == 0) return; // everything lives in LDS islands
    const rp_integration_params &prm = w.prm.p;
    const bool fib = prm.friction_in_bias_pass || prm.num_internal_stabilization_iterations == 0;
    const int nb = w.n_bodies;
    __syncthreads();
    for (int i = t; i < nb; i += nt) if (global_body(w, i)) g_body_begin(w, i);
    __threadfence(); __syncthreads();
    for (int pos = t; pos < M; pos += nt) if (g_generate<COUL>(w, pos)) bouncy = 1;
    __threadfence(); __syncthreads();
    for (int sub = 0; sub < w.prm.num_substeps; ++sub) {
        float solved_dt = (float)sub * w.prm.dt_sub;
        for (int i = t; i < nb; i += nt) if (global_body(w, i)) g_body_increment(w, i);
        for (int j = t; j < nj; j += nt) if (joint_live(w, j)) joint_update_one(w, j, sub); // reads poses only
        __threadfence(); __syncthreads();
        tail_sweep<MODE_WARMSTART, COUL>(w, 0, fib, solved_dt);
        for (int it = 0; it < prm.num_internal_pgs_iterations; ++it) {
            joint_tail_sweep(w, 0, false, prm.warmstart_joints && it == 0); // every joint before any contact
            tail_sweep<MODE_BIAS, COUL>(w, 0, fib, solved_dt);
        }
        for (int i = t; i < nb; i += nt) if (global_body(w, i)) g_body_integrate(w, i);
        __threadfence(); __syncthreads();
        for (int it = 0; it < prm.num_internal_stabilization_iterations; ++it) {
            joint_tail_sweep(w, 0, true, false);
            tail_sweep<MODE_RELAX, COUL>(w, 0, fib, solved_dt + w.prm.dt_sub);
        }
    }
    if (has_restitution && bouncy) tail_sweep<MODE_RESTITUTION, COUL>(w, 0, fib, 0.0f);
    for (int pos = t; pos < M; pos += nt) { if (COUL) coul_writeback(w, GlobalAcc(w, pos), w.cons_pair[pos]); else cons_writeback(w, GlobalAcc(w, pos), w.cons_pair[pos]); }
    for (int j = t; j < nj; j += nt) if (joint_live(w, j)) joint_writeback_one(w, j);
    for (int i = t; i < nb; i += nt) if (global_body(w, i)) g_body_writeback(w, i);
}

