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

// ---- MULTI mode kernels ---------------------------------------------------------------------------
__global__ void k_solver_begin(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) w.flags[FL_ANY_BOUNCY] = 0;
    if (i >= w.n_bodies || !global_body(w, i)) return;
    g_body_begin(w, i);
}
template <bool COUL>
__global__ void k_generate(DevWorld w) {
    int M = w.flags[FL_N_CONS];
    if (M > w.cons_cap) M = w.cons_cap;
    int stride = gridDim.x * blockDim.x;
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < M; pos += stride)
        if (g_generate<COUL>(w, pos)) w.flags[FL_ANY_BOUNCY] = 1;
}
__global__ void k_increment(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w.n_bodies || !global_body(w, i)) return;
    g_body_increment(w, i);
}
__global__ void k_integrate(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w.n_bodies || !global_body(w, i)) return;
    g_body_integrate(w, i);
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
    int M = w.flags[FL_N_CONS];
    if (M > w.cons_cap) M = w.cons_cap;
    int stride = gridDim.x * blockDim.x;
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < M; pos += stride) { if (COUL) coul_writeback(w, GlobalAcc(w, pos), w.cons_pair[pos]); else cons_writeback(w, GlobalAcc(w, pos), w.cons_pair[pos]); }
}
__global__ void k_writeback_bodies(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { w.flags[FL_STEP] += 1; w.flags[FL_SEQ] += 1; }
    if (i >= w.n_bodies || !global_body(w, i)) return;
    g_body_writeback(w, i);
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
        if (!(threshold < 3.402823466e+38f) || !pair_selected(w, s)) continue;
        int npts = w.p_npts[s];
        float total = 0.0f;
        for (int k = 0; k < npts; ++k) total += PT(w.pt_imp, k, s).x; // every tracked point, like ContactManifoldExt::total_impulse
        float total_magnitude = (0.0f + total) * inv_dt;
        int pf = w.p_pflags[s];
        if (total_magnitude > threshold) {
            V3 normal = v3(w.p_normal[s]);
            float max_mag = 0.0f, tmi = 0.0f; V3 max_dir = v3(0, 0, 0);
            for (int k = 0; k < npts; ++k) {
                float imp = PT(w.pt_imp, k, s).x;
                tmi += imp;
                if (imp > max_mag) { max_mag = imp; max_dir = normal; }
            }
            V3 total_force = (v3(0, 0, 0) + normal * tmi) * inv_dt;
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
void rp_launch_solver_assembly(const DevWorld &w, hipStream_t st) {
    hipLaunchKernelGGL(k_solver_begin, dim3(body_blocks(w)), dim3(256), 0, st, w);
    if (host_coulomb(w)) hipLaunchKernelGGL(k_generate<true>, dim3(cons_blocks(w)), dim3(256), 0, st, w);
    else hipLaunchKernelGGL(k_generate<false>, dim3(cons_blocks(w)), dim3(256), 0, st, w);
}
// The TGS loop proper: S2..S7 for every substep (+ S8 restitution) — worker.rs:207-734.
void rp_launch_solver_loop(const DevWorld &w, hipStream_t st, int parallel_stages, int stage_blocks, int has_restitution, int joint_stages) {
    SolverLaunchPlan plan = {parallel_stages, stage_blocks < 1 ? 1 : stage_blocks};
    int nb = body_blocks(w);
    const rp_integration_params &p = w.prm.p;
    int fib = (p.friction_in_bias_pass || p.num_internal_stabilization_iterations == 0) ? 1 : 0;
    for (int s = 0; s < w.prm.num_substeps; ++s) {
        float solved_dt = (float)s * w.prm.dt_sub;
        hipLaunchKernelGGL(k_increment, dim3(nb), dim3(256), 0, st, w);
        rp_launch_joint_update(w, st, s); // rows rebuilt from the current poses (worker.rs:287-357)
        launch_sweep<MODE_WARMSTART>(w, st, plan, fib, solved_dt);
        for (int it = 0; it < p.num_internal_pgs_iterations; ++it) {
            rp_launch_joint_sweep(w, st, joint_stages, 0, (p.warmstart_joints && it == 0) ? 1 : 0); // all joints before any contact
            launch_sweep<MODE_BIAS>(w, st, plan, fib, solved_dt);
        }
        hipLaunchKernelGGL(k_integrate, dim3(nb), dim3(256), 0, st, w);
        for (int it = 0; it < p.num_internal_stabilization_iterations; ++it) {
            rp_launch_joint_sweep(w, st, joint_stages, 1, 0);
            launch_sweep<MODE_RELAX>(w, st, plan, fib, solved_dt + w.prm.dt_sub);
        }
    }
    if (has_restitution) launch_sweep<MODE_RESTITUTION>(w, st, plan, fib, 0.0f);
}
void rp_launch_solver_writeback(const DevWorld &w, hipStream_t st) {
    if (host_coulomb(w)) hipLaunchKernelGGL(k_writeback_impulses<true>, dim3(cons_blocks(w)), dim3(256), 0, st, w);
    else hipLaunchKernelGGL(k_writeback_impulses<false>, dim3(cons_blocks(w)), dim3(256), 0, st, w);
    rp_launch_joint_writeback(w, st);
    hipLaunchKernelGGL(k_writeback_bodies, dim3(body_blocks(w)), dim3(256), 0, st, w);
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, w); // hint buffer (MULTI mode: after the step)
}
