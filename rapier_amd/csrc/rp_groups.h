// rp_groups.h — substep solve-groups (RigidBody::additional_solver_iterations) on the device.
//
// Reference: /root/reference/src/dynamics/island_manager/substep_groups.rs:44-229 (the partition) and
// dynamics/solver/staged_island_solver/init.rs:52-100, 163-420 (one GroupLayout per distinct count: its own substep count and dt,
// its own chunk layout and joint layout, the groups solved one after the other in descending cadence).
//
//   * connected components of the awake DYNAMIC bodies over the pairs that hold an active contact and over the impulse joints
//     take the largest additional_solver_iterations of their members; a kinematic body is lifted to the largest count among the
//     dynamic bodies it touches; one group per distinct count (the host keeps the table of distinct counts, descending:
//     DevWorld::grp_extra / grp_sub — a count no awake body holds this step is simply an empty group);
//   * group g runs num_solver_iterations + extra[g] substeps of length dt / that, over its own bodies, joints and constraints
//     (a constraint or joint follows the highest group index — the lowest cadence — among its solver bodies); the >= 32-chunk
//     threshold that orders the colour stages (init.rs:169) and the >= 64-joint threshold of the joint layout (joints.rs:352) are
//     applied to the group's own share of every colour.
//
// Worlds that hold an elevated body are solved by ONE workgroup (k_global_groups below): every body is kept out of the LDS
// islands (k_isl_count) and the per-stage / dataflow launches are not used.  Such scenes are small assemblies (chains with a
// high mass ratio, robot arms); the arithmetic is the very same cons_* / joint_* code with the group's substep parameters.
#pragma once
#include "rp_global.h"

RP_DEV int grp_ld(int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RP_DEV int grp_find(int *label, int x) {
    int p = grp_ld(&label[x]);
    while (p != x) { x = p; p = grp_ld(&label[x]); }
    return x;
}
RP_DEV void grp_union(int *label, int a, int b) { // lock-free: the larger root is hooked under the smaller one
    for (;;) {
        a = grp_find(label, a); b = grp_find(label, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        if (atomicCAS(&label[a], a, b) == a) return;
    }
}
// substep_groups.rs:66-70 `slot`: an awake dynamic body; :150-157 `kinematic_slot`: an awake non-dynamic, non-fixed body
RP_DEV bool grp_dyn_slot(const DevWorld &w, int b) { if (b < 0) return false; int fl = w.b_flags[b]; return flags_active(fl) && (fl & RP_BF_TYPE_MASK) == RP_BODY_DYNAMIC; }
RP_DEV bool grp_kin_slot(const DevWorld &w, int b) { if (b < 0) return false; int fl = w.b_flags[b]; return flags_active(fl) && (fl & RP_BF_TYPE_MASK) != RP_BODY_DYNAMIC; }
RP_DEV int grp_of_active(const DevWorld &w, int b) { return (b >= 0 && flags_active(w.b_flags[b])) ? w.b_group[b] : 0; }
RP_DEV DevWorld grp_world(const DevWorld &w0, int g) { // the world as group g sees it: its own substep count, dt and spring coefficients
    DevWorld w = w0;
    const SubParams sp = w0.grp_sub[g];
    w.prm.dt_sub = sp.dt_sub; w.prm.inv_dt_sub = sp.inv_dt_sub;
    w.prm.dyn_cfm = sp.dyn_cfm; w.prm.static_cfm = sp.static_cfm; w.prm.dyn_erp_inv_dt = sp.dyn_erp_inv_dt; w.prm.static_erp_inv_dt = sp.static_erp_inv_dt;
    w.prm.joint_erp_inv_dt = sp.joint_erp_inv_dt; w.prm.joint_cfm_coeff = sp.joint_cfm_coeff; w.prm.num_substeps = sp.num_substeps;
    return w;
}

// one sweep of group g over its contact constraints, in the group's own stage order (init.rs:163-254 on the group's colour counts)
template <int MODE, bool COUL>
RP_DEV void grp_contact_sweep(const DevWorld &w, int g, const int *cg, const int *st_of_color, bool fib, float solved_dt) {
    for (int pass = 0; pass < 2; ++pass)
        for (int c = 0; c < RP_COLOR_OVERFLOW; ++c) {
            const int n = cg[g * RP_NUM_COLORS + c];
            if (n == 0 || ((n + 3) / 4 >= 32) != (pass == 0)) continue;
            const int st = st_of_color[c];
            if (st < 0) continue;
            const int beg = w.stage_begin[st], cnt = w.stage_count[st];
            for (int i = threadIdx.x; i < cnt; i += blockDim.x) if (w.k_group[beg + i] == g) cons_apply_model<COUL>(w, GlobalAcc(w, beg + i), MODE, fib, solved_dt);
            __threadfence(); __syncthreads();
        }
    if (cg[g * RP_NUM_COLORS + RP_COLOR_OVERFLOW] > 0) { // the overflow colour is not body-disjoint: serially, in bucket order
        const int st = st_of_color[RP_COLOR_OVERFLOW];
        if (st >= 0 && threadIdx.x == 0) {
            const int beg = w.stage_begin[st], cnt = w.stage_count[st];
            for (int i = 0; i < cnt; ++i) if (w.k_group[beg + i] == g) { cons_apply_model<COUL>(w, GlobalAcc(w, beg + i), MODE, fib, solved_dt); __threadfence(); }
        }
        __threadfence(); __syncthreads();
    }
}
// one sweep of group g over its joints: the group's parallel colours (>= 64 joints of the group) ascending, then the others ascending
RP_DEV void grp_joint_sweep(const DevWorld &w, int g, const int *cj, bool wo_bias, bool warmstart) {
    if (w.n_joints == 0) return;
    const int n_live = w.flags[FL_NJ_OVF_BEGIN] + w.flags[FL_NJ_OVF_COUNT];
    for (int pass = 0; pass < 2; ++pass)
        for (int c = 0; c < RP_NUM_COLORS; ++c) {
            const int n = cj[g * RP_NUM_COLORS + c];
            const bool parallel = c < RP_COLOR_OVERFLOW && n >= 64;
            if (n == 0 || parallel != (pass == 0)) continue;
            if (c < RP_COLOR_OVERFLOW) { // joints of one colour are body-disjoint
                for (int idx = threadIdx.x; idx < n_live; idx += blockDim.x) { int j = w.j_order[idx]; if (w.j_color[j] == c && w.j_group[j] == g) joint_solve_one(w, j, wo_bias, warmstart); }
            } else if (threadIdx.x == 0) { // uncoloured joints: serially, in edge order (j_order keeps them colour-major in edge order)
                for (int idx = 0; idx < n_live; ++idx) { int j = w.j_order[idx]; if (w.j_color[j] == c && w.j_group[j] == g) { joint_solve_one(w, j, wo_bias, warmstart); __threadfence(); } }
            }
            __threadfence(); __syncthreads();
        }
}

template <bool COUL>
RP_DEV void global_groups_block(const DevWorld &w0, int has_restitution, int fast) {
    const int t = threadIdx.x, nt = blockDim.x;
    if (fast && w0.flags[FL_FAST_ABORT]) return;
    __shared__ int bouncy, cg[RP_MAX_GROUPS * RP_NUM_COLORS], cj[RP_MAX_GROUPS * RP_NUM_COLORS], st_of_color[RP_NUM_COLORS], nbg[RP_MAX_GROUPS];
    int M = w0.flags[FL_N_CONS]; if (M > w0.cons_cap) M = w0.cons_cap;
    int top = w0.flags[FL_POOL_TOP]; if (top > w0.pool_cap) top = w0.pool_cap;
    const int nb = w0.n_bodies, nj = w0.n_joints, G = w0.n_groups;
    const rp_integration_params &prm = w0.prm.p;
    const bool fib = prm.friction_in_bias_pass || prm.num_internal_stabilization_iterations == 0;
    if (t == 0) bouncy = 0;
    for (int k = t; k < RP_MAX_GROUPS * RP_NUM_COLORS; k += nt) { cg[k] = 0; cj[k] = 0; }
    for (int k = t; k < RP_NUM_COLORS; k += nt) st_of_color[k] = -1;
    for (int k = t; k < RP_MAX_GROUPS; k += nt) nbg[k] = 0;
    // ---- the partition (substep_groups.rs:44-229) ----
    for (int i = t; i < nb; i += nt) { w0.g_parent[i] = i; w0.g_key[i] = 0; w0.b_group[i] = G - 1; }
    __threadfence(); __syncthreads();
    for (int s = t; s < top; s += nt) { // contact edges: pairs with an active contact between two awake dynamic bodies
        if (w0.p_c1[s] < 0 || w0.p_nsc[s] == 0) continue;
        int b1 = w0.c_parent[w0.p_c1[s]], b2 = w0.c_parent[w0.p_c2[s]];
        if (grp_dyn_slot(w0, b1) && grp_dyn_slot(w0, b2)) grp_union(w0.g_parent, b1, b2);
    }
    for (int j = t; j < nj; j += nt) { // every impulse joint between two awake dynamic bodies
        int b1 = w0.j_b1[j], b2 = w0.j_b2[j];
        if (grp_dyn_slot(w0, b1) && grp_dyn_slot(w0, b2)) grp_union(w0.g_parent, b1, b2);
    }
    __threadfence(); __syncthreads();
    for (int i = t; i < nb; i += nt) { // the component's count = the largest count among its members
        if (!grp_dyn_slot(w0, i)) continue;
        int extra = w0.b_extra[i];
        if (extra > 0) atomicMax(&w0.g_key[grp_find(w0.g_parent, i)], extra);
    }
    __threadfence(); __syncthreads();
    for (int i = t; i < nb; i += nt) { // (only non-roots are written, only roots are read)
        if (!grp_dyn_slot(w0, i)) continue;
        int root = grp_find(w0.g_parent, i);
        if (root != i) w0.g_key[i] = grp_ld(&w0.g_key[root]);
    }
    __threadfence(); __syncthreads();
    for (int s = t; s < top; s += nt) { // a kinematic body joins the highest-cadence group it touches
        if (w0.p_c1[s] < 0 || w0.p_nsc[s] == 0) continue;
        int b1 = w0.c_parent[w0.p_c1[s]], b2 = w0.c_parent[w0.p_c2[s]];
        if (grp_kin_slot(w0, b1) && grp_dyn_slot(w0, b2)) atomicMax(&w0.g_key[b1], grp_ld(&w0.g_key[b2]));
        if (grp_kin_slot(w0, b2) && grp_dyn_slot(w0, b1)) atomicMax(&w0.g_key[b2], grp_ld(&w0.g_key[b1]));
    }
    for (int j = t; j < nj; j += nt) {
        int b1 = w0.j_b1[j], b2 = w0.j_b2[j];
        if (grp_kin_slot(w0, b1) && grp_dyn_slot(w0, b2)) atomicMax(&w0.g_key[b1], grp_ld(&w0.g_key[b2]));
        if (grp_kin_slot(w0, b2) && grp_dyn_slot(w0, b1)) atomicMax(&w0.g_key[b2], grp_ld(&w0.g_key[b1]));
    }
    __threadfence(); __syncthreads();
    for (int i = t; i < nb; i += nt) {
        if (!flags_active(w0.b_flags[i])) continue;
        const int key = grp_ld(&w0.g_key[i]);
        int g = G - 1;
        for (int q = 0; q < G; ++q) if (w0.grp_extra[q] == key) { g = q; break; }
        w0.b_group[i] = g;
        if (global_body(w0, i)) atomicAdd(&nbg[g], 1);
    }
    __threadfence(); __syncthreads();
    if (M == 0 && w0.flags[FL_N_GLOB_BODIES] == 0 && nj == 0) return;
    // ---- S0 solver bodies (the increments use the substep length of the body's group), S1 generate ----
    for (int i = t; i < nb; i += nt) if (global_body(w0, i)) { const DevWorld wg = grp_world(w0, w0.b_group[i]); g_body_begin(wg, i); }
    __threadfence(); __syncthreads();
    for (int pos = t; pos < M; pos += nt) if (g_generate<COUL>(w0, pos)) bouncy = 1;
    {   // colour -> stage of the global position layout; the overflow colour is the stage behind the last one
        const int nst = w0.flags[FL_N_STAGES];
        for (int st = t; st < nst; st += nt) st_of_color[w0.stage_color[st]] = st;
        if (t == 0 && w0.flags[FL_HAS_OVERFLOW_COLOR]) st_of_color[RP_COLOR_OVERFLOW] = nst;
    }
    for (int pos = t; pos < M; pos += nt) { // a constraint follows the lowest cadence among its solver bodies
        const int s = w0.cons_pair[pos];
        const int g1 = grp_of_active(w0, w0.c_parent[w0.p_c1[s]]), g2 = grp_of_active(w0, w0.c_parent[w0.p_c2[s]]);
        const int g = g1 > g2 ? g1 : g2;
        w0.k_group[pos] = g;
        atomicAdd(&cg[g * RP_NUM_COLORS + w0.p_color[s]], 1);
    }
    for (int j = t; j < nj; j += nt) {
        if (!joint_live(w0, j)) { w0.j_group[j] = -1; continue; }
        const int g1 = grp_of_active(w0, w0.j_b1[j]), g2 = grp_of_active(w0, w0.j_b2[j]);
        const int g = g1 > g2 ? g1 : g2;
        w0.j_group[j] = g;
        atomicAdd(&cj[g * RP_NUM_COLORS + w0.j_color[j]], 1);
    }
    __threadfence(); __syncthreads();
    // ---- the groups, in descending cadence, each with its whole substep loop ----
    for (int g = 0; g < G; ++g) {
        if (nbg[g] == 0) continue; // no awake body holds this count
        const DevWorld w = grp_world(w0, g);
        for (int sub = 0; sub < w.prm.num_substeps; ++sub) {
            const float solved_dt = (float)sub * w.prm.dt_sub;
            for (int i = t; i < nb; i += nt) if (global_body(w, i) && w.b_group[i] == g) g_body_increment(w, i);
            for (int j = t; j < nj; j += nt) if (w.j_group[j] == g) joint_update_one(w, j, sub); // reads poses only
            __threadfence(); __syncthreads();
            grp_contact_sweep<MODE_WARMSTART, COUL>(w, g, cg, st_of_color, fib, solved_dt);
            for (int it = 0; it < prm.num_internal_pgs_iterations; ++it) {
                grp_joint_sweep(w, g, cj, false, prm.warmstart_joints && it == 0); // every joint before any contact
                grp_contact_sweep<MODE_BIAS, COUL>(w, g, cg, st_of_color, fib, solved_dt);
            }
            for (int i = t; i < nb; i += nt) if (global_body(w, i) && w.b_group[i] == g) g_body_integrate(w, i);
            __threadfence(); __syncthreads();
            for (int it = 0; it < prm.num_internal_stabilization_iterations; ++it) {
                grp_joint_sweep(w, g, cj, true, false);
                grp_contact_sweep<MODE_RELAX, COUL>(w, g, cg, st_of_color, fib, solved_dt + w.prm.dt_sub);
            }
        }
    }
    if (has_restitution && bouncy)
        for (int g = 0; g < G; ++g) if (nbg[g] != 0) grp_contact_sweep<MODE_RESTITUTION, COUL>(w0, g, cg, st_of_color, fib, 0.0f);
    for (int pos = t; pos < M; pos += nt) { if (COUL) coul_writeback(w0, GlobalAcc(w0, pos), w0.cons_pair[pos]); else cons_writeback(w0, GlobalAcc(w0, pos), w0.cons_pair[pos]); }
    for (int j = t; j < nj; j += nt) if (joint_live(w0, j)) joint_writeback_one(w0, j);
    for (int i = t; i < nb; i += nt) if (global_body(w0, i)) g_body_writeback(w0, i);
}
