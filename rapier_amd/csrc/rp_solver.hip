// rp_solver.hip — the global (HBM-resident) colour-ordered TGS-soft contact solver + integrator.
//
// Restates StagedIslandSolver::init_and_solve / run_worker
// (/root/reference/src/dynamics/solver/staged_island_solver/{init.rs:30-545, worker.rs:32-898,
// solve.rs:12-209}) for everything the LDS island kernel (rp_islands.hip) does not own: islands too
// large for one CU's LDS and bodies without contacts.  The reference's 4-lane AoSoA chunks + worker
// stage machine become:
//   * one thread per solver manifold, constraint data in float4 planes C[plane][position] so a
//     wavefront's 64 consecutive positions load 1 KiB per plane (coalesced, HBM/L2 streaming);
//   * MULTI mode: one launch per (sweep, colour stage) for colours with >= 32 chunks ("parallel"
//     colours, init.rs:169): same-colour manifolds touch disjoint dynamic bodies, so the body
//     gather/scatter needs no atomics; plus one single-workgroup "tail" launch
//     per sweep for the small colours (ascending) and the overflow colour (serial, lane 0) — the
//     reference runs exactly those on worker 0 (init.rs:192-254).  The tail also absorbs any parallel
//     stage the host did not launch, so the Gauss-Seidel order never depends on host knowledge;
//   * SINGLE mode: when the global path holds little or no work (the usual case once every island
//     fits in LDS) ONE single-workgroup launch runs the whole assembly/loop/write-back sequence.
// Both modes are correct for any amount of work; the host picks by lazily read hints.
// No FMA contraction (-ffp-contract=off), IEEE divide/sqrt: same arithmetic as the reference's lanes.
#include "rp_global.h"
#include "rp_groups.h"
#include <utility>
#include <cstdlib>
#include <algorithm>

// ---- MULTI mode kernels ---------------------------------------------------------------------------
// S0 + S1 in one launch: generate reads the solver bodies straight from the body arrays (lin / ang = the body velocities, solver pose =
// (rotation, world centre of mass): the very expressions of g_body_begin, recomputed per manifold side), so the two need no kernel
// boundary between them
template <bool PRE>
struct GenAccT : GlobalAccT<PRE> {
    RP_DEV GenAccT(const DevWorld &w_, int pos_) : GlobalAccT<PRE>(w_, pos_) {}
    RP_DEV Vel vel(int id) const {
        const DevWorld &w = this->w;
        Vel v;
        if (id < 0) { v.lin = v3(0, 0, 0); v.ang = v3(0, 0, 0); } else { v.lin = v3(w.b_linvel[id]); v.ang = v3(w.b_angvel[id]); }
        return v;
    }
    RP_DEV Xf xf(int id) const {
        const DevWorld &w = this->w;
        Xf x;
        if (id < 0) { x.r = q4(0, 0, 0, 1); x.t = v3(0, 0, 0); }
        else { x.r = q4(w.b_rot[id]); x.t = qrot(x.r, v3(w.b_lcom_invm[id])) + v3(w.b_pos[id]); }
        return x;
    }
};
template <bool COUL>
__global__ void __launch_bounds__(256) k_begin_generate(DevWorld w) { // (FL_ANY_BOUNCY was reset by k_flow_ranks, the launch before)
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (lean_dead(w)) return; // (a live lean step has FL_FLOW_DIRTY == 0 already: the store below changes nothing for the lanes still evaluating this)
    if (gid == 0) w.flags[FL_FLOW_DIRTY] = 0; // (the toucher ranks and the tiling were rebuilt by the launches before this one)
    for (int i = gid; i < w.n_bodies; i += stride) if (global_body(w, i)) g_body_begin(w, i);
    int M = w.flags[FL_N_CONS];
    if (M > w.cons_cap) M = w.cons_cap;
    for (int pos = gid; pos < M; pos += stride) {
        const int s = w.cons_pair[pos];
        const int rb1 = w.c_parent[w.p_c1[s]], rb2 = w.c_parent[w.p_c2[s]], rel_dom = w.p_reldom[s];
        const int id1 = (body_active(w, rb1) && rel_dom <= 0) ? rb1 : -1, id2 = (body_active(w, rb2) && rel_dom >= 0) ? rb2 : -1; // (g_generate)
        const bool bouncy = COUL ? coul_generate(w, GenAccT<true>(w, pos), s, id1, id2, id1, id2) : cons_generate(w, GenAccT<true>(w, pos), s, id1, id2, id1, id2);
        if (bouncy) w.flags[FL_ANY_BOUNCY] = 1;
    }
}
__global__ void k_increment(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (lean_dead(w)) return;
    if (i >= w.n_bodies || !global_body(w, i)) return;
    g_body_increment(w, i);
}
__global__ void k_integrate(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (lean_dead(w)) return;
    if (i >= w.n_bodies || !global_body(w, i)) return;
    g_body_integrate(w, i);
}
// ---- body-centric warm start (twist model) -------------------------------------------------------------------------------
// The reference fuses `update` + `warmstart` into its colour sweep (contact_with_twist_friction.rs:426-522, :633-678; stage
// worker.rs:438-488) and pays one stage per colour for it.  Neither half needs that: `update` reads poses only, and the warm start
// adds velocity terms that do not depend on the velocities — per body, a fixed sequence of float adds in sweep order.  So the
// per-stage path runs them as TWO launches per substep instead of one per colour (9 on b3d_large_pyramid):
//   k_ws_prepare    every manifold in parallel: update (rhs, cfm, banked impulses) and the 11 velocity terms of either side
//                   (5 linear: the <= 4 normal terms + the tangent term; 6 angular: + the twist term) into ws_terms;
//   k_increment_ws  every body: increment (+ gyroscopic term), then its touchers' terms in sweep order (f_sorted, built with the
//                   dataflow solver's toucher ranks), each added exactly as the colour sweep would have added it.
// Same operands, same order per accumulator (-ffp-contract=off): bit-identical to the per-colour sweep and to the oracle.
// (WS_TERMS, ws_put / ws_get, body_increment_ws: rp_global.h — shared with the tile sweeps of rp_tiles.hip)
// joint_substep >= 0: the launch also rebuilds the rows of every impulse joint from the current poses (k_joint_update folded in: both
// are the pose-dependent, fully parallel preparation of a substep; jointed worlds on tiles save a launch per substep)
// (JOINTS = false: the contact half alone — the general joint update keeps its rows in scratch, 592 B per lane, and its registers would
// set the occupancy of a launch that b3d_large_pyramid makes four times a step without a single joint)
template <bool JOINTS>
__global__ void __launch_bounds__(256) k_ws_prepare(DevWorld w, float solved_dt, int joint_substep) {
    if (lean_dead(w)) return;
    if (JOINTS && joint_substep >= 0) {
        const int jstride = gridDim.x * blockDim.x;
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < w.n_joints; j += jstride) if (joint_live(w, j)) joint_update_one(w, j, joint_substep);
    }
    int M = w.flags[FL_N_CONS];
    if (M > w.cons_cap) M = w.cons_cap;
    const int stride = gridDim.x * blockDim.x;
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < M; pos += stride) ws_prepare_one(w, GlobalAcc(w, pos), pos, solved_dt);
}
// The linear and the angular half of a body are independent chains (increment + five terms per toucher / gyroscopic increment + six
// terms per toucher): the first half of the grid does the one, the second half the other — twice the wavefronts, about half the
// dependent loads per lane (each half computes body_increment_ws and keeps one result: the other chain, its loads included, is dead
// code the compiler removes).  The island kernel splits its bodies the same way (role_lin / role_ang).
__global__ void __launch_bounds__(256) k_increment_ws(DevWorld w) {
    const int half_blocks = gridDim.x >> 1;
    const bool angular = (int)blockIdx.x >= half_blocks;
    const int i = ((int)blockIdx.x - (angular ? half_blocks : 0)) * blockDim.x + threadIdx.x;
    if (lean_dead(w)) return;
    if (i >= w.n_bodies || !global_body(w, i)) return;
    V3 lin, ang;
    if (angular) { body_increment_ws(w, i, lin, ang); w.s_ang[i] = f4(ang, 0.0f); }
    else { body_increment_ws(w, i, lin, ang); w.s_lin[i] = f4(lin, 0.0f); }
}

// One parallel colour stage: the reference's claim/steal chunk loop becomes a grid-stride loop.
template <int MODE, bool COUL>
__global__ void __launch_bounds__(256) k_stage(DevWorld w, int stage, int friction_in_bias, float solved_dt) {
    if (stage >= w.flags[FL_N_PARALLEL]) return;
    if (MODE == MODE_RESTITUTION && !w.flags[FL_ANY_BOUNCY]) return;
    int beg = w.stage_begin[stage], cnt = w.stage_count[stage];
    int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride)
        cons_apply_model<COUL>(w, GlobalAcc(w, beg + i), MODE, friction_in_bias != 0, solved_dt);
}
template <int MODE, bool COUL>
__global__ void __launch_bounds__(1024) k_tail(DevWorld w, int first, int friction_in_bias, float solved_dt) {
    if (MODE == MODE_RESTITUTION && !w.flags[FL_ANY_BOUNCY]) return;
    int npar = w.flags[FL_N_PARALLEL];
    tail_sweep<MODE, COUL>(w, first < npar ? first : npar, friction_in_bias != 0, solved_dt);
}
template <bool COUL>
__global__ void k_writeback_impulses(DevWorld w) {
    if (lean_dead(w)) return;
    int M = w.flags[FL_N_CONS];
    if (M > w.cons_cap) M = w.cons_cap;
    int stride = gridDim.x * blockDim.x;
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < M; pos += stride) { if (COUL) coul_writeback(w, GlobalAcc(w, pos), w.cons_pair[pos]); else cons_writeback(w, GlobalAcc(w, pos), w.cons_pair[pos]); }
    // (the joints' impulses ride the same launch: an independent item-parallel loop — one launch less per step in jointed worlds)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < w.n_joints; j += stride) if (joint_live(w, j)) joint_writeback_one(w, j);
}
__global__ void k_writeback_bodies(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (lean_dead(w)) return; // (the step did not happen: k_ccd counts the graph — FL_SEQ — and raises the marker)
    if (i == 0) { w.flags[FL_STEP] += 1; w.flags[FL_SEQ] += 1; }
    // a bare lean graph (no manifold in the world: rp_world.h) has no contact impulse to write back: the joints' impulses — what is left
    // of k_writeback_impulses — ride this launch (one launch less per step on b3d_joint_grid)
    if (w.lean & 2) { const int stride = gridDim.x * blockDim.x; for (int j = i; j < w.n_joints; j += stride) if (joint_live(w, j)) joint_writeback_one(w, j); }
    if (i >= w.n_bodies || !global_body(w, i)) return;
    g_body_writeback(w, i);
}

// The two write-backs as ONE launch (round 6): they share no word — the first body_blocks workgroups write the bodies back, the rest the
// impulses (grid-stride, as k_writeback_impulses) — so a step of a contact world pays one launch and the longer of the two instead of
// two launches end to end (b3d_large_pyramid: 8.6 + 6.5 us + a gap)
template <bool COUL>
__global__ void k_writeback(DevWorld w, int body_blocks) {
    if (lean_dead(w)) return; // (the step did not happen: k_ccd counts the graph — FL_SEQ — and raises the marker)
    if ((int)blockIdx.x < body_blocks) {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i == 0) { w.flags[FL_STEP] += 1; w.flags[FL_SEQ] += 1; }
        if (i >= w.n_bodies || !global_body(w, i)) return;
        g_body_writeback(w, i);
        return;
    }
    int M = w.flags[FL_N_CONS];
    if (M > w.cons_cap) M = w.cons_cap;
    const int stride = ((int)gridDim.x - body_blocks) * blockDim.x, first = ((int)blockIdx.x - body_blocks) * blockDim.x + threadIdx.x;
    for (int pos = first; pos < M; pos += stride) { if (COUL) coul_writeback(w, GlobalAcc(w, pos), w.cons_pair[pos]); else cons_writeback(w, GlobalAcc(w, pos), w.cons_pair[pos]); }
    for (int j = first; j < w.n_joints; j += stride) if (joint_live(w, j)) joint_writeback_one(w, j);
}

// NarrowPhase::emit_contact_force_events (solver_graph.rs:462-498) + ContactForceEvent::from_contact_pair
// (geometry/mod.rs:223-258): one thread per pair slot, after the impulses of the step were written back.
__global__ void k_force_events(DevWorld w, int fast) {
    if (fast && w.flags[FL_FAST_ABORT]) return; // the fast graph gave up on this step: it is replayed (events included) on the full graph
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    const float dt = w.prm.p.dt, inv_dt = dt == 0.0f ? 0.0f : 1.0f / dt;
    const int step = w.flags[FL_STEP]; // the step that just retired
    int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < top; s += stride) {
        int c1 = w.p_c1[s];
        if (c1 < 0) continue;
        int c2 = w.p_c2[s];
        float2 e1 = w.c_events[c1], e2 = w.c_events[c2];
        float ta = (__float_as_int(e1.x) & RP_EVENTS_CONTACT_FORCE) ? e1.y : 3.402823466e+38f;
        float tb = (__float_as_int(e2.x) & RP_EVENTS_CONTACT_FORCE) ? e2.y : 3.402823466e+38f;
        float threshold = ta < tb ? ta : tb;
        if (w.has_composite && pair_is_aux(w, s)) continue; // (a cluster of a composite pair: summed with its parent below)
        if (!(threshold < 3.402823466e+38f) || !pair_selected(w, s)) continue;
        // every solver manifold of the pair (ContactPair::solver_manifolds: the plain manifold, or the clusters: this slot + its aux slots)
        const int nsm = (w.has_composite && w.p_aux[s].w > 1) ? w.p_aux[s].w : 1;
        float total = 0.0f;
        for (int q = 0; q < nsm; ++q) { const int sq = sm_slot(w, s, q); if (sq < 0) continue; const int npts = w.p_npts[sq]; for (int k = 0; k < npts; ++k) total += PT(w.pt_imp, k, sq).x; } // every tracked point, like ContactManifoldExt::total_impulse
        float total_magnitude = (0.0f + total) * inv_dt;
        int pf = w.p_pflags[s];
        if (total_magnitude > threshold) {
            float max_mag = 0.0f; V3 max_dir = v3(0, 0, 0), total_force = v3(0, 0, 0);
            for (int q = 0; q < nsm; ++q) {
                const int sq = sm_slot(w, s, q);
                if (sq < 0) continue;
                const V3 normal = v3(w.p_normal[sq]);
                const int npts = w.p_npts[sq];
                float tmi = 0.0f;
                for (int k = 0; k < npts; ++k) {
                    float imp = PT(w.pt_imp, k, sq).x;
                    tmi += imp;
                    if (imp > max_mag) { max_mag = imp; max_dir = normal; }
                }
                total_force = total_force + normal * tmi;
            }
            total_force = total_force * inv_dt;
            int k = atomicAdd(&w.flags[FL_EV_FORCE], 1);
            if (k < w.ev_cap) {
                w.ev_force_meta[k] = make_int4(c1, c2, step, (pf & RP_PF_FORCE_EMITTED) ? 0 : 1);
                w.ev_force_a[k] = f4(total_force, total_magnitude);
                w.ev_force_b[k] = f4(max_dir, max_mag * inv_dt);
            }
            w.p_pflags[s] = pf | RP_PF_FORCE_EMITTED;
        } else if (pf & RP_PF_FORCE_EMITTED) w.p_pflags[s] = pf & ~RP_PF_FORCE_EMITTED;
    }
}
void rp_launch_force_events(const DevWorld &w, hipStream_t st, int fast) {
    if (!w.has_force_events || w.n_colliders == 0) return;
    int blocks = (w.pool_cap + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_force_events, dim3(blocks), dim3(256), 0, st, w, fast);
}

__global__ void k_publish(DevWorld w) { publish_flags(w); }
template <bool COUL>
__global__ void __launch_bounds__(512) k_global_single(DevWorld w, int has_restitution, int fast) { global_single_block<COUL>(w, has_restitution, fast); }
// worlds with substep solve-groups: the whole global path, group by group, in one workgroup (rp_groups.h)
template <bool COUL>
__global__ void __launch_bounds__(512) k_global_groups(DevWorld w, int has_restitution, int fast) { global_groups_block<COUL>(w, has_restitution, fast); }

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
        int la = (fl >> RP_BF_LOCK_SHIFT) & 0x3f;
        w.b_eim[i] = make_float4((la & 1) ? 0.0f : li.w, (la & 2) ? 0.0f : li.w, (la & 4) ? 0.0f : li.w, 0.0f);
        Sym3 ii = world_inv_inertia(v3(w.b_invpi[i]), q4(w.b_pframe[i]), rot);
        apply_locked_rotations(la, ii);
        w.b_eii0[i] = make_float4(ii.m11, ii.m12, ii.m13, ii.m22);
        w.b_eii1[i] = make_float4(ii.m23, ii.m33, 0.0f, 0.0f);
    } else {
        w.b_eim[i] = make_float4(0, 0, 0, 0); w.b_eii0[i] = make_float4(0, 0, 0, 0); w.b_eii1[i] = make_float4(0, 0, 0, 0);
    }
}

void rp_launch_joint_update(const DevWorld &w, hipStream_t st, int substep_id);
void rp_launch_joint_sweep(const DevWorld &w, hipStream_t st, int parallel_stages, int wo_bias, int warmstart);
void rp_launch_joint_writeback(const DevWorld &w, hipStream_t st);

// ---- host-side launch sequences -------------------------------------------------------------------
struct SolverLaunchPlan { int parallel_stages; int stage_blocks; };

static bool host_coulomb(const DevWorld &w) { return w.prm.p.friction_model == RP_FRICTION_COULOMB; }
template <int MODE, bool COUL>
static void launch_sweep_model(const DevWorld &w, hipStream_t st, const SolverLaunchPlan &plan, int fib, float solved_dt) {
    for (int s = 0; s < plan.parallel_stages; ++s)
        hipLaunchKernelGGL((k_stage<MODE, COUL>), dim3(plan.stage_blocks * 4), dim3(64), 0, st, w, s, fib, solved_dt); // one wave per workgroup: a colour stage of ~10k manifolds then spreads over ~150 CUs instead of ~40
    hipLaunchKernelGGL((k_tail<MODE, COUL>), dim3(1), dim3(1024), 0, st, w, plan.parallel_stages, fib, solved_dt);
}
template <int MODE>
static void launch_sweep(const DevWorld &w, hipStream_t st, const SolverLaunchPlan &plan, int fib, float solved_dt) {
    if (host_coulomb(w)) launch_sweep_model<MODE, true>(w, st, plan, fib, solved_dt); else launch_sweep_model<MODE, false>(w, st, plan, fib, solved_dt);
}
static int body_blocks(const DevWorld &w) { int nb = (w.n_bodies + 255) / 256; return nb < 1 ? 1 : nb; }
static int cons_blocks(const DevWorld &w) { int cb = (w.cons_cap + 255) / 256; if (cb > 2048) cb = 2048; return cb < 1 ? 1 : cb; }

// the flags an edit of a live world raises (rp_api.hip after_topology_edit): one thread
__global__ void k_edit_flags(DevWorld w, int keep_grid) {
    if (!keep_grid) w.flags[FL_BP_GRID_OK] = 0;
    w.lay_state[0] = 0; // bodies came, went or changed their kind: the next layout rebuild finds its components from scratch (rp_islands.hip)
    w.flags[FL_BP_DIRTY] = 1; w.flags[FL_LAYOUT_DIRTY] = 1; w.flags[FL_JOINT_DIRTY] = 1; w.flags[FL_FLOW_DIRTY] = 1;
}
// The device's step stamps are 32-bit (FL_STEP and everything stamped with cur_step): long before they could wrap — the host asks for
// it once FL_STEP passes 2^29, ~9 h at the headline rate — every stamp moves back by `delta` steps.  Stamps that are only compared for
// EQUALITY with the step in progress (sleep observation, island marks) and lie more than delta steps back become 0 = never; the two
// that are ORDERED against each other (a body's last fall-asleep step against the step a pair's solver hint was computed in) keep
// their order across the line (older ones collapse onto 1: at worst a hint is recomputed once).  The sleep-scan stamp (one per step)
// and the islands' split cooldowns move with it.  Runs between steps, on an idle stream.
__global__ void k_rebase_stamps(DevWorld w, int delta) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    auto eq = [&](int v) { return v > delta ? v - delta : 0; };
    auto ord = [&](int v) { return v > delta ? v - delta : (v > 0 ? 1 : 0); };
    const unsigned long long scan = w.pi_w64[0];
    const int scan_stamp = (int)(unsigned)(scan & 0xffffffffull), scan_step = (int)(scan >> 32);
    const bool move_scan = scan_stamp > delta;
    for (int i = gid; i < w.n_bodies; i += stride) {
        w.b_sleep_stamp[i] = eq(w.b_sleep_stamp[i]); w.lab_wake[i] = eq(w.lab_wake[i]); w.lab_awake[i] = eq(w.lab_awake[i]);
        w.b_slept_at[i] = ord(w.b_slept_at[i]);
        if (move_scan) { const int d = w.pi_denied[i]; w.pi_denied[i] = d > delta ? d - delta : 0; }
    }
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    for (int s = gid; s < top; s += stride) w.p_hint_seq[s] = ord(w.p_hint_seq[s]);
    // events still queued keep their ORDER against the ones to come: their stamps move by delta as they are (signed: an event from before
    // the line gets a stamp <= 0; the host adds the steps moved so far when it hands events out, rp_collision_events_read)
    if (w.ev_col) { int n = w.flags[FL_EV_COL]; if (n > w.ev_cap) n = w.ev_cap; for (int k = gid; k < n; k += stride) w.ev_col[k].w -= delta; }
    if (w.ev_force_meta) { int n = w.flags[FL_EV_FORCE]; if (n > w.ev_cap) n = w.ev_cap; for (int k = gid; k < n; k += stride) w.ev_force_meta[k].z -= delta; }
    if (gid == 0) {
        w.flags[FL_STEP] -= delta; w.flags[FL_WAKE_STAMP] = 0;
        w.pi_w64[0] = ((unsigned long long)(unsigned)eq(scan_step) << 32) | (unsigned)(move_scan ? scan_stamp - delta : scan_stamp);
    }
}
void rp_launch_rebase_stamps(const DevWorld &w, hipStream_t st, int delta) {
    int n = w.n_bodies > w.pool_cap ? w.n_bodies : w.pool_cap, blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_rebase_stamps, dim3(blocks), dim3(256), 0, st, w, delta);
}
void rp_launch_edit_flags(const DevWorld &w, hipStream_t st, int keep_grid) { hipLaunchKernelGGL(k_edit_flags, dim3(1), dim3(1), 0, st, w, keep_grid); }
void rp_launch_init_bodies(const DevWorld &w, hipStream_t st) {
    if (w.n_bodies == 0) return;
    hipLaunchKernelGGL(k_init_bodies, dim3(body_blocks(w)), dim3(256), 0, st, w);
}
void rp_launch_global_single(const DevWorld &w, hipStream_t st, int has_restitution, int fast) {
    if (w.n_groups > 1) {
        if (host_coulomb(w)) hipLaunchKernelGGL(k_global_groups<true>, dim3(1), dim3(512), 0, st, w, has_restitution, fast);
        else hipLaunchKernelGGL(k_global_groups<false>, dim3(1), dim3(512), 0, st, w, has_restitution, fast);
        return;
    }
    if (host_coulomb(w)) hipLaunchKernelGGL(k_global_single<true>, dim3(1), dim3(512), 0, st, w, has_restitution, fast);
    else hipLaunchKernelGGL(k_global_single<false>, dim3(1), dim3(512), 0, st, w, has_restitution, fast);
}
void rp_launch_flow_ranks(const DevWorld &w, hipStream_t st);
void rp_launch_tiles_build(const DevWorld &w, hipStream_t st);
void rp_launch_tile_sweep(const DevWorld &w, hipStream_t st, int mode, int grid, int friction_in_bias, float solved_dt, int fuse, int joint_warmstart);
void rp_launch_joint_net_step(const DevWorld &w, hipStream_t st, int grid, int joint_warmstart);
void rp_launch_tile_step(const DevWorld &w, hipStream_t st, int grid, int friction_in_bias);
// lean: the graph runs only while FL_FLOW_DIRTY is clear (rp_world.h "lean step graphs") — both rebuilds would exit at once: left out
void rp_launch_solver_assembly(const DevWorld &w, hipStream_t st, int lean) {
    if (!lean) {
        rp_launch_flow_ranks(w, st); // the per-body toucher lists of the body-centric warm start (only rebuilt when the layout changed)
        rp_launch_tiles_build(w, st); // ... and the LDS tiling of the big component (rp_tiles.hip; same gate)
    }
    const int blocks = std::max(body_blocks(w), cons_blocks(w)) > 2048 ? 2048 : std::max(body_blocks(w), cons_blocks(w));
    if (host_coulomb(w)) hipLaunchKernelGGL(k_begin_generate<true>, dim3(blocks), dim3(256), 0, st, w);
    else hipLaunchKernelGGL(k_begin_generate<false>, dim3(blocks), dim3(256), 0, st, w);
}
// The TGS loop proper: S2..S7 for every substep (+ S8 restitution) — worker.rs:207-734.
// tile_grid > 0: every biased / relaxed sweep is ONE launch over the LDS tiles (rp_tiles.hip) instead of one per colour stage.  A tile
// sweep reads the solver velocities from one buffer and writes the other, so the kernels that follow get a DevWorld with the two
// pointer pairs swapped; returns the parity (1 = the velocities ended in t_lin / t_ang) for rp_launch_solver_writeback.
int rp_launch_solver_loop(const DevWorld &w0, hipStream_t st, int parallel_stages, int stage_blocks, int has_restitution, int joint_stages, int tile_grid, int no_contacts_hint) {
    SolverLaunchPlan plan = {parallel_stages, stage_blocks < 1 ? 1 : stage_blocks};
    DevWorld w = w0;
    int parity = 0; // bit 0: velocities + mutable constraint planes in the other copy, bit 1: poses in the other copy
    const bool tiles = tile_grid > 0 && w.tile_cap > 0 && !host_coulomb(w) && w.ws_terms; // (a tile sweep = the joint stages, then the contact stages)
    int nb = body_blocks(w);
    const rp_integration_params &p = w.prm.p;
    int fib = (p.friction_in_bias_pass || p.num_internal_stabilization_iterations == 0) ? 1 : 0;
    // with tiles the first biased sweep of a substep also increments + warm-starts the bodies, the last one also integrates them
    const int fuse_mask = 2; // (bit 0, the increment folded into the sweep's prologue, measured slower on worlds with contacts: every halo body pays the gather again)
    const bool can_fuse = tiles && p.num_internal_pgs_iterations >= 1;
    // (a world without contact manifolds — a hint: the folded form is correct either way — has no warm-start terms to gather: the
    // increment rides the sweep's prologue for free; with contacts the gather of every halo body costs more than the launch, measured)
    // (worlds whose joints are all spherical: the first biased sweep of a substep rebuilds the joint rows itself — rp_tiles.hip, tile_joint_build)
    static const bool joint_inline_ok = getenv("RP_NO_JOINT_INLINE") == nullptr;
    const bool jinline = can_fuse && w.n_joints > 0 && w.joints_spherical && joint_inline_ok;
    const bool fuse_inc = can_fuse && ((fuse_mask & 1) || ((fuse_mask & 4) == 0 && no_contacts_hint)), fuse_int = can_fuse && (fuse_mask & 2);
    // a bare lean graph in the joint-net form (DevWorld::lean bit 2, planned by the host, verified by lean_dead): the whole loop is ONE
    // launch that keeps every tile's joints in registers (k_joint_net_step, rp_tiles.hip) — b3d_joint_grid
    if (w.lean & 4) {
        rp_launch_joint_net_step(w, st, w.lean >> 8, p.warmstart_joints ? 1 : 0);
        return (w.prm.num_substeps & 1) ? 2 : 0; // (velocities where they began, poses in the other copy after an odd number of substeps)
    }
    // a lean graph of a tiled contact world in the one-launch form (DevWorld::lean bit 3, planned by the host, verified by lean_dead): the
    // four launches of every substep are phases of k_tile_step (rp_tiles.hip) — b3d_large_pyramid
    if (w.lean & 8) {
        rp_launch_tile_step(w, st, w.lean >> 8, fib);
        return (w.prm.num_substeps & 1) ? 2 : 0; // (velocities and mutable planes where they began, poses in the other copy after an odd number of substeps)
    }
#define TILE_SWEEP(MODE, SDT, FUSE, JWS) do { const int fuse_ = (FUSE); rp_launch_tile_sweep(w, st, MODE, tile_grid, fib, SDT, fuse_, JWS); \
        std::swap(w.s_lin, w.t_lin); std::swap(w.s_ang, w.t_ang); w.c_par ^= 1; parity ^= 1; \
        if (fuse_ & 2) { std::swap(w.s_rot, w.t_rot); std::swap(w.s_trans, w.t_trans); parity ^= 2; } } while (0)
    // (Folding k_ws_prepare into the owner instances of the relaxed sweep — the rows are in registers there — was built and measured:
    // the extra arithmetic and 22 term stores per manifold inside the issue-bound stage loop cost more than the three launches it
    // removes: b3d_large_pyramid solver 0.472 -> 0.497 ms.  It stays a launch.)
    for (int s = 0; s < w.prm.num_substeps; ++s) {
        float solved_dt = (float)s * w.prm.dt_sub;
        if (!host_coulomb(w) && w.ws_terms) { // body-centric warm start: two launches instead of one per colour
            if (tiles && w.n_joints > 0 && !jinline) hipLaunchKernelGGL(k_ws_prepare<true>, dim3(cons_blocks(w)), dim3(256), 0, st, w, solved_dt, s);
            else if (!(w.lean & 2)) hipLaunchKernelGGL(k_ws_prepare<false>, dim3(cons_blocks(w)), dim3(256), 0, st, w, solved_dt, -1); // (a bare lean graph holds no manifold: rp_world.h)
            if (!fuse_inc) hipLaunchKernelGGL(k_increment_ws, dim3(2 * nb), dim3(256), 0, st, w); // (linear halves, then angular halves)
            if (!tiles) rp_launch_joint_update(w, st, s);
        } else {
            hipLaunchKernelGGL(k_increment, dim3(nb), dim3(256), 0, st, w);
            rp_launch_joint_update(w, st, s); // rows rebuilt from the current poses (worker.rs:287-357)
            launch_sweep<MODE_WARMSTART>(w, st, plan, fib, solved_dt);
        }
        for (int it = 0; it < p.num_internal_pgs_iterations; ++it) {
            const int jws = (p.warmstart_joints && it == 0) ? 1 : 0;
            if (tiles) TILE_SWEEP(MODE_BIAS, solved_dt, ((fuse_inc && it == 0) ? 1 : 0) | ((fuse_int && it == p.num_internal_pgs_iterations - 1) ? 2 : 0) | ((jinline && it == 0) ? (4 | (s << 8)) : 0), jws);
            else { rp_launch_joint_sweep(w, st, joint_stages, 0, jws); launch_sweep<MODE_BIAS>(w, st, plan, fib, solved_dt); } // all joints before any contact
        }
        if (!fuse_int) hipLaunchKernelGGL(k_integrate, dim3(nb), dim3(256), 0, st, w);
        for (int it = 0; it < p.num_internal_stabilization_iterations; ++it) {
            if (tiles) TILE_SWEEP(MODE_RELAX, solved_dt + w.prm.dt_sub, 0, 0);
            else { rp_launch_joint_sweep(w, st, joint_stages, 1, 0); launch_sweep<MODE_RELAX>(w, st, plan, fib, solved_dt + w.prm.dt_sub); }
        }
    }
#undef TILE_SWEEP
    if (has_restitution) launch_sweep<MODE_RESTITUTION>(w, st, plan, fib, 0.0f);
    return parity;
}
void rp_launch_solver_writeback(const DevWorld &w0, hipStream_t st, int parity, int publish) {
    DevWorld w = w0;
    if (parity & 1) { std::swap(w.s_lin, w.t_lin); std::swap(w.s_ang, w.t_ang); w.c_par = 1; } // the tile sweeps left the velocities (and the mutable constraint planes) in the other copy
    if (parity & 2) { std::swap(w.s_rot, w.t_rot); std::swap(w.s_trans, w.t_trans); } // ... and the poses
    if (w.lean & 2) hipLaunchKernelGGL(k_writeback_bodies, dim3(body_blocks(w)), dim3(256), 0, st, w); // (bare lean graph: the joints' write-back rides k_writeback_bodies)
    else if (host_coulomb(w)) hipLaunchKernelGGL(k_writeback<true>, dim3(body_blocks(w) + cons_blocks(w)), dim3(256), 0, st, w, body_blocks(w));
    else hipLaunchKernelGGL(k_writeback<false>, dim3(body_blocks(w) + cons_blocks(w)), dim3(256), 0, st, w, body_blocks(w));
    if (publish) hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, w); // hint buffer (MULTI mode: after the step; else k_ccd carries it)
}
