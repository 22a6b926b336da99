// rp_flow.hip — the global solver path as ONE dataflow launch per step.
//
// What it replaces: rp_solver.hip runs StagedIslandSolver's stage machine
// (/root/reference/src/dynamics/solver/staged_island_solver/{worker.rs:207-734, solve.rs:12-209, sync.rs:39-186}) as one kernel
// launch per (sweep, colour stage): ~110 dependent launches per step for an 8-colour scene, each one a chain of dependent memory
// round trips (kernel arguments -> stage table -> constraint -> body gather -> scatter), 5-9 us apiece whatever the stage holds.
// The reference pays a worker barrier at the same places.  Neither the barrier nor the launch boundary is required by the
// algorithm: a Gauss-Seidel sweep in colour order only demands that the constraints touching ONE body are applied in sweep
// order.  So on MI355X the stage barrier becomes a per-body hand-off:
//
//   * every solver body's velocity record carries a ticket: f_rec[2b].w = f_rec[2b + 1].w = (epoch << 21) | events completed on b;
//   * every event on a body — increment, each contact's warm start / biased / relaxed / restitution application, each joint
//     solve, integrate, write-back — knows its ticket from the body's toucher lists (ranks in sweep order, built once per
//     layout change by the k_flow_* kernels below): it polls the body records of its (at most two) bodies until they show
//     its tickets, applies itself, and stores the records back with ticket + 1;
//   * records travel between CUs / XCDs as 16-byte write-through (sc1) stores and L1-bypassing (sc1) loads, the data IS the
//     flag (MI355X guide, Guideline 16 form R2: one naturally aligned granule per store, no fence, no separate flag);
//     poses follow the velocity record of the integrate event (sc1 stores, vmcnt(0), then the record: form R1).
//
// The critical path of a step is then the longest dependency chain (~ colours x sweeps hand-offs of ~1-2 us) instead of
// (colours x sweeps) grid-wide barriers, different parts of the scene run ahead of each other, and the arithmetic is the very
// same cons_* / joint_* code the per-stage path instantiates, on the same operands in the same per-body order: the result is
// bit-identical to rp_solver.hip and to the oracle.
//
// Progress: all threads walk the same phase sequence and, inside a phase, their items in ascending sweep position, so the
// globally smallest unfinished (phase, position) is always runnable by a thread that has reached it; lanes of one wavefront
// never wait for each other (a lane that is ready proceeds while its neighbours keep polling).  That argument needs every
// workgroup resident: the grid is sized from the occupancy query (rp_api.hip) and every wait is bounded — a timeout raises
// FL_FLOW_ABORT / RP_OVF_FLOW, all waves leave their loops and the host reports the step as failed.
#include "rp_global.h"
#include "rp_gridbar.h"

typedef unsigned int flow_u4 __attribute__((ext_vector_type(4)));
#define FLOW_SC1 16                  // cache policy bits of the raw buffer builtins on gfx950: sc1 = agent scope (write-through / L1 bypass)
#define FLOW_TICKET_BITS 21
#define FLOW_TICKET_MASK ((1u << FLOW_TICKET_BITS) - 1u)
#define FLOW_TIMEOUT_TICKS 300000000ll // wall_clock64 runs at 100 MHz: 3 s

struct FlowBufs { __amdgpu_buffer_rsrc_t rec, rot, trans; }; // rec: [2 * body] = (lin, tag), (ang, tag): both halves of a record share a 32-byte slot
RP_DEV __amdgpu_buffer_rsrc_t flow_rsrc(void *p, int n_items) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, n_items * 16, 0x00020000); }
RP_DEV flow_u4 flow_ld(__amdgpu_buffer_rsrc_t r, int i) { return __builtin_amdgcn_raw_buffer_load_b128(r, i * 16, 0, FLOW_SC1); }
RP_DEV void flow_st(__amdgpu_buffer_rsrc_t r, int i, V3 v, unsigned tag) {
    flow_u4 x; x.x = (unsigned)__float_as_int(v.x); x.y = (unsigned)__float_as_int(v.y); x.z = (unsigned)__float_as_int(v.z); x.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(x, r, i * 16, 0, FLOW_SC1);
}
RP_DEV void flow_st4(__amdgpu_buffer_rsrc_t r, int i, float4 v) {
    flow_u4 x; x.x = (unsigned)__float_as_int(v.x); x.y = (unsigned)__float_as_int(v.y); x.z = (unsigned)__float_as_int(v.z); x.w = (unsigned)__float_as_int(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(x, r, i * 16, 0, FLOW_SC1);
}
#ifdef RP_FLOW_TRACE
__device__ long long *g_flow_trace; __device__ int g_flow_trace_body;
#define FLOW_TRACE(i, tag) do { if ((i) == g_flow_trace_body) g_flow_trace[64 + ((tag) & 1023u)] = (long long)wall_clock64(); } while (0)
#else
#define FLOW_TRACE(i, tag) do { } while (0)
#endif
RP_DEV void flow_st_vel(const FlowBufs &B, int i, V3 lin, V3 ang, unsigned tag) { FLOW_TRACE(i, tag); flow_st(B.rec, 2 * i, lin, tag); flow_st(B.rec, 2 * i + 1, ang, tag); }
RP_DEV V3 flow_v3(flow_u4 x) { return v3(__int_as_float((int)x.x), __int_as_float((int)x.y), __int_as_float((int)x.z)); }
RP_DEV Q4 flow_q4(flow_u4 x) { return q4(__int_as_float((int)x.x), __int_as_float((int)x.y), __int_as_float((int)x.z), __int_as_float((int)x.w)); }

struct FlowCtx {
    FlowBufs B;
    unsigned epoch;       // (epoch << FLOW_TICKET_BITS) of this step: tags of other steps never match
    int *abort_flag;      // flags[FL_FLOW_ABORT]
    int *ovf_flag;        // flags[FL_OVERFLOW]
    long long t0;
    bool dead;            // this wave saw the abort flag: skip every remaining wait
    unsigned n_items, n_polls, n_applies; // per wave: wait loops entered, poll rounds, APPLY executions (flushed to dbg[20..22])
    long long t_apply, t_wait;            // wall-clock ticks (10 ns) spent inside APPLY / inside the wait loops (dbg[24], dbg[25])
};

// The event schedule of one body inside a step (see the file header): tickets are event indices.
struct FlowSched { int npgs, nstab, nsub, hr; };
RP_DEV int fs_events(const FlowSched &s, int dc, int dj) { return 2 + dc + (s.npgs + s.nstab) * (dj + dc); }
RP_DEV int fs_incr(const FlowSched &s, int dc, int dj, int sub) { return sub * fs_events(s, dc, dj); }
RP_DEV int fs_ws(const FlowSched &s, int dc, int dj, int sub, int r) { return sub * fs_events(s, dc, dj) + 1 + r; }
RP_DEV int fs_jbias(const FlowSched &s, int dc, int dj, int sub, int it, int rj) { return sub * fs_events(s, dc, dj) + 1 + dc + it * (dj + dc) + rj; }
RP_DEV int fs_cbias(const FlowSched &s, int dc, int dj, int sub, int it, int r) { return fs_jbias(s, dc, dj, sub, it, 0) + dj + r; }
RP_DEV int fs_integ(const FlowSched &s, int dc, int dj, int sub) { return sub * fs_events(s, dc, dj) + 1 + dc + s.npgs * (dj + dc); }
RP_DEV int fs_jrelax(const FlowSched &s, int dc, int dj, int sub, int it, int rj) { return fs_integ(s, dc, dj, sub) + 1 + it * (dj + dc) + rj; }
RP_DEV int fs_crelax(const FlowSched &s, int dc, int dj, int sub, int it, int r) { return fs_jrelax(s, dc, dj, sub, it, 0) + dj + r; }
RP_DEV int fs_rest(const FlowSched &s, int dc, int dj, int r) { return s.nsub * fs_events(s, dc, dj) + r; }
RP_DEV int fs_final(const FlowSched &s, int dc, int dj) { return s.nsub * fs_events(s, dc, dj) + (s.hr ? dc : 0); }

// One poll of the records of up to two bodies: 0 when both show their expected tags (then v1 / v2 hold the velocities), else how
// many events the slower body still is away from this one (the polling cadence adapts to it).
RP_DEV unsigned flow_gap(unsigned lw, unsigned aw, unsigned e) {
    if (lw == e && aw == e) return 0u;
    const unsigned c = lw < aw ? lw : aw;
    return ((c ^ e) & ~FLOW_TICKET_MASK) ? 1024u : (e - c > 0u ? e - c : 1u); // another step's record: its owner has not begun yet
}
RP_DEV unsigned flow_poll(const FlowCtx &cx, int i1, unsigned e1, int i2, unsigned e2, Vel &v1, Vel &v2) {
    flow_u4 l1 = {0, 0, 0, e1}, a1 = {0, 0, 0, e1}, l2 = {0, 0, 0, e2}, a2 = {0, 0, 0, e2};
    if (i1 >= 0) { l1 = flow_ld(cx.B.rec, 2 * i1); a1 = flow_ld(cx.B.rec, 2 * i1 + 1); }
    if (i2 >= 0) { l2 = flow_ld(cx.B.rec, 2 * i2); a2 = flow_ld(cx.B.rec, 2 * i2 + 1); }
    v1.lin = flow_v3(l1); v1.ang = flow_v3(a1); v2.lin = flow_v3(l2); v2.ang = flow_v3(a2);
    const unsigned g1 = flow_gap(l1.w, a1.w, e1), g2 = flow_gap(l2.w, a2.w, e2);
    return g1 > g2 ? g1 : g2;
}
// rounds a lane sits out after a poll that found it `gap` events away from its ticket (an event takes a microsecond or more)
// (the chip-wide request rate bounds a poll round: ~1 us with a thousand fully polling wavefronts, ~0.2 us when a quarter of the
// lanes poll — tools/ubench/pollcost.hip; so lanes that cannot be next stay quiet, and wake up together when a neighbour fires)
RP_DEV unsigned flow_skip(unsigned gap) { return gap <= 1u ? 0u : (gap >= 9u ? 32u : 4u * (gap - 1u)); }
// Wave-uniform back-off after an iteration in which no lane of the wavefront made progress; returns false when the launch is
// being aborted (timeout here or in another wave).
RP_DEV bool flow_backoff(FlowCtx &cx, unsigned &spins) {
    __builtin_amdgcn_s_sleep(6); // ~0.15 us: the cadence of a round in which no (or few) lanes poll
    if ((++spins & 255u) != 0u) return true;
    int ab = __hip_atomic_load(cx.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!ab && (long long)wall_clock64() - cx.t0 > FLOW_TIMEOUT_TICKS) {
        __hip_atomic_store(cx.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicOr(cx.ovf_flag, RP_OVF_FLOW);
        ab = 1;
    }
    if (ab) cx.dead = true;
    return !ab;
}
// pose readers outside the ticket chain (joint rows): wait until the body's record shows this step's epoch and at least `need`
RP_DEV bool flow_reached(const FlowCtx &cx, int i, int need) {
    if (i < 0) return true;
    flow_u4 l = flow_ld(cx.B.rec, 2 * i);
    return (l.w & ~FLOW_TICKET_MASK) == cx.epoch && (int)(l.w & FLOW_TICKET_MASK) >= need;
}

// ---- accessors ------------------------------------------------------------------------------------
// generate reads the start-of-step body state, which nothing modifies before the final write-back: no hand-off needed.
// (same expressions as body_begin: lin / ang = the body velocities, solver pose = (rotation, world centre of mass))
#define FLOW_JROWS 3 // joint rows fetched ahead of the ticket wait (a spherical joint has 3)
struct FlowGenAcc {
    static constexpr bool PRELOAD = false; // persistent kernel at 256 VGPRs already: see GlobalAccT
    const DevWorld &w; int pos;
    RP_DEV FlowGenAcc(const DevWorld &w_, int pos_) : w(w_), pos(pos_) {}
    RP_DEV void st(int plane, float4 v) const { w.C[(size_t)plane * w.cons_cap + pos] = v; }
    RP_DEV void set_meta(int a, int b, int cnt, int cid) const { w.k_b1[pos] = a; w.k_b2[pos] = b; w.k_n[pos] = cnt; w.k_cid[pos] = cid; }
    RP_DEV Vel vel(int id) const {
        Vel v;
        if (id < 0) { v.lin = v3(0, 0, 0); v.ang = v3(0, 0, 0); } else { v.lin = v3(w.b_linvel[id]); v.ang = v3(w.b_angvel[id]); }
        return v;
    }
    RP_DEV Xf xf(int id) const {
        Xf x;
        if (id < 0) { x.r = q4(0, 0, 0, 1); x.t = v3(0, 0, 0); }
        else { x.r = q4(w.b_rot[id]); x.t = qrot(x.r, v3(w.b_lcom_invm[id])) + v3(w.b_pos[id]); }
        return x;
    }
};
// sweeps: constraint planes are private to the owning thread (plain accesses); bodies come from the poll and leave as tagged
// write-through records; the poses are fetched by the wait loop as soon as they are final for the sweep (FLOW_RUN_CONTACT).
// (Preloading the planes into registers ahead of the wait was measured and dropped: the hand-off latency, not the plane loads,
// bounds a hop — DESIGN.md §4.6.)
template <bool PRE>
struct FlowAccT {
    static constexpr bool PRELOAD = PRE; // the twist model has the registers for it (no spill), the Coulomb model does not: see GlobalAccT
    const DevWorld &w; const FlowCtx &cx; int pos, i1, i2, nn; unsigned t1, t2; // t = tag to publish (expected + 1)
    mutable Vel v1, v2;
    mutable Xf x1, x2;     // solver poses of the two bodies, loaded by the wait loop as soon as they are final for this sweep
    mutable bool wrote1, wrote2;
    RP_DEV FlowAccT(const DevWorld &w_, const FlowCtx &cx_, int pos_, int i1_, int i2_, int n_, unsigned t1_, unsigned t2_)
        : w(w_), cx(cx_), pos(pos_), i1(i1_), i2(i2_), nn(n_), t1(t1_), t2(t2_), wrote1(false), wrote2(false) {}
    RP_DEV float4 ld(int plane) const { return w.C[(size_t)plane * w.cons_cap + pos]; }
    RP_DEV void st(int plane, float4 v) const { w.C[(size_t)plane * w.cons_cap + pos] = v; }
    RP_DEV int id1() const { return i1; }
    RP_DEV int id2() const { return i2; }
    RP_DEV int n() const { return nn; }
    RP_DEV int cids() const { return w.k_cid[pos]; }
    RP_DEV Vel vel(int id) const {
        Vel z; z.lin = v3(0, 0, 0); z.ang = v3(0, 0, 0);
        return id < 0 ? z : (id == i1 ? v1 : v2);
    }
    RP_DEV void set_vel(int id, const Vel &v) const {
        if (id < 0) return;
        const bool first = id == i1;
        const unsigned t = first ? t1 : t2;
        flow_st_vel(cx.B, id, v.lin, v.ang, t);
        if (first) wrote1 = true; else wrote2 = true;
    }
    RP_DEV void load_poses() const {
        x1.r = q4(0, 0, 0, 1); x1.t = v3(0, 0, 0); x2 = x1;
        if (i1 >= 0) { x1.r = flow_q4(flow_ld(cx.B.rot, i1)); x1.t = flow_v3(flow_ld(cx.B.trans, i1)); }
        if (i2 >= 0) { x2.r = flow_q4(flow_ld(cx.B.rot, i2)); x2.t = flow_v3(flow_ld(cx.B.trans, i2)); }
    }
    RP_DEV Xf xf(int id) const { Xf z; z.r = q4(0, 0, 0, 1); z.t = v3(0, 0, 0); return id < 0 ? z : (id == i1 ? x1 : x2); }
    // an application that left a body's velocity alone still completes its event on that body
    RP_DEV void finish() const {
        if (i1 >= 0 && !wrote1) { flow_st_vel(cx.B, i1, v1.lin, v1.ang, t1); }
        if (i2 >= 0 && !wrote2) { flow_st_vel(cx.B, i2, v2.lin, v2.ang, t2); }
    }
};
struct FlowJointIO {
    const FlowCtx &cx; Vel v[2]; unsigned t[2];
    RP_DEV void bodies(const DevWorld &w_, int j, int &b1, int &b2) const { b1 = w_.j_b1[j]; b2 = w_.j_b2[j]; }
    RP_DEV void pose(int side, int b, Pose &p) const { p.r = flow_q4(flow_ld(cx.B.rot, b)); p.t = flow_v3(flow_ld(cx.B.trans, b)); }
    RP_DEV void load_vel(int side, int b, V3 &l, V3 &a) const { l = v[side].lin; a = v[side].ang; }
    RP_DEV void store_vel(int side, int b, V3 l, V3 a) const { flow_st_vel(cx.B, b, l, a, t[side]); }
    RP_DEV int jm_out(const DevWorld &w_) const { return w_.c_par; }
    RP_DEV bool jm_store() const { return true; }
};

// ---- toucher lists: ranks of every constraint / joint among the events of its bodies (rebuilt when the layout changed) ---
// (flow_ids: rp_global.h)
RP_DEV int flow_live_joints(const DevWorld &w) { return w.n_joints > 0 ? w.flags[FL_NJ_OVF_BEGIN] + w.flags[FL_NJ_OVF_COUNT] : 0; }
// pass 0: count (into the fill cursors, which rest at zero between rebuilds)
RP_DEV void flow_count(DevWorld &w, int gid, int stride) {
    int M = w.flags[FL_N_CONS]; if (M > w.cons_cap) M = w.cons_cap;
    const int njl = flow_live_joints(w);
    if (gid == 0) { w.flags[FL_FLOW_CURSOR] = 0; w.flags[FL_FLOW_JCURSOR] = 0; }
    for (int pos = gid; pos < M; pos += stride) {
        int a, b; flow_ids(w, pos, a, b);
        if (a >= 0) atomicAdd(&w.fb_fill[a].x, 1);
        if (b >= 0) atomicAdd(&w.fb_fill[b].x, 1);
    }
    for (int idx = gid; idx < njl; idx += stride) {
        int j = w.j_order[idx], a = w.j_b1[j], b = w.j_b2[j];
        if (a >= 0) atomicAdd(&w.fb_fill[a].y, 1);
        if (b >= 0) atomicAdd(&w.fb_fill[b].y, 1);
    }
}
// pass 1: every body reserves its two list ranges
RP_DEV void flow_alloc(DevWorld &w, int gid, int stride) {
  for (int i = gid; i < w.n_bodies; i += stride) {
    int2 d = w.fb_fill[i];
    w.fb_deg[i] = d;
    { // every ticket of a step must fit FLOW_TICKET_BITS
        const rp_integration_params &p = w.prm.p;
        long long ev = 2ll + d.x + (long long)(p.num_internal_pgs_iterations + p.num_internal_stabilization_iterations) * (d.x + d.y);
        if (ev * w.prm.num_substeps + d.x + 2 > (long long)FLOW_TICKET_MASK) atomicOr(&w.flags[FL_OVERFLOW], RP_OVF_FLOW);
    }
    w.fb_begin[i] = make_int2(d.x ? atomicAdd(&w.flags[FL_FLOW_CURSOR], d.x) : 0, d.y ? atomicAdd(&w.flags[FL_FLOW_JCURSOR], d.y) : 0);
    w.fb_fill[i] = make_int2(0, 0);
  }
}
// pass 2: fill
RP_DEV void flow_fill(DevWorld &w, int gid, int stride) {
    int M = w.flags[FL_N_CONS]; if (M > w.cons_cap) M = w.cons_cap;
    const int njl = flow_live_joints(w);
    for (int pos = gid; pos < M; pos += stride) {
        int a, b; flow_ids(w, pos, a, b);
        if (a >= 0) w.f_adj[w.fb_begin[a].x + atomicAdd(&w.fb_fill[a].x, 1)] = pos;
        if (b >= 0) w.f_adj[w.fb_begin[b].x + atomicAdd(&w.fb_fill[b].x, 1)] = pos;
    }
    for (int idx = gid; idx < njl; idx += stride) {
        int j = w.j_order[idx], a = w.j_b1[j], b = w.j_b2[j];
        if (a >= 0) w.f_jadj[w.fb_begin[a].y + atomicAdd(&w.fb_fill[a].y, 1)] = idx;
        if (b >= 0) w.f_jadj[w.fb_begin[b].y + atomicAdd(&w.fb_fill[b].y, 1)] = idx;
    }
}
// pass 3: rank = how many touchers of the same body come earlier in the sweep (constraint position / joint sweep index)
RP_DEV int flow_rank_in(const int *list, int begin, int n, int key) { int r = 0; for (int k = 0; k < n; ++k) r += list[begin + k] < key; return r; }
RP_DEV void flow_rank(DevWorld &w, int gid, int stride) {
    int M = w.flags[FL_N_CONS]; if (M > w.cons_cap) M = w.cons_cap;
    const int njl = flow_live_joints(w);
    for (int pos = gid; pos < M; pos += stride) {
        int a, b; flow_ids(w, pos, a, b);
        int2 r = make_int2(-1, -1);
        if (a >= 0) { r.x = flow_rank_in(w.f_adj, w.fb_begin[a].x, w.fb_deg[a].x, pos); w.f_sorted[w.fb_begin[a].x + r.x] = 2 * pos; w.f_other[w.fb_begin[a].x + r.x] = b; }
        if (b >= 0) { r.y = flow_rank_in(w.f_adj, w.fb_begin[b].x, w.fb_deg[b].x, pos); w.f_sorted[w.fb_begin[b].x + r.y] = 2 * pos + 1; w.f_other[w.fb_begin[b].x + r.y] = a; }
        w.fk_rank[pos] = r;
    }
    for (int idx = gid; idx < njl; idx += stride) {
        int j = w.j_order[idx], a = w.j_b1[j], b = w.j_b2[j];
        int2 r = make_int2(-1, -1);
        if (a >= 0) { r.x = flow_rank_in(w.f_jadj, w.fb_begin[a].y, w.fb_deg[a].y, idx); if (w.f_jsorted) { w.f_jsorted[w.fb_begin[a].y + r.x] = idx; w.f_jother[w.fb_begin[a].y + r.x] = b; } }
        if (b >= 0) { r.y = flow_rank_in(w.f_jadj, w.fb_begin[b].y, w.fb_deg[b].y, idx); if (w.f_jsorted) { w.f_jsorted[w.fb_begin[b].y + r.y] = idx; w.f_jother[w.fb_begin[b].y + r.y] = a; } }
        w.fj_rank[j] = r;
    }
    for (int i = gid; i < w.n_bodies; i += stride) w.fb_fill[i] = make_int2(0, 0); // cursors rest at zero for the next rebuild
}

// the four passes as ONE launch behind grid barriers (rp_gridbar.h): a step whose layout did not change pays a single early exit
__global__ void __launch_bounds__(1024) k_flow_ranks(DevWorld w) {
    if (blockIdx.x == 0 && threadIdx.x == 0) w.flags[FL_ANY_BOUNCY] = 0; // (the first launch of every solver assembly: k_begin_generate / k_flow_begin raise it)
    if (!w.flags[FL_FLOW_DIRTY]) return; // (cleared by the kernel that starts the solve: k_flow_begin / k_begin_generate)
    const int gid = gbar_item(), gstride = gridDim.x * blockDim.x;
    GridBar bar = gbar_begin(w, 3);
    flow_count(w, gid, gstride);
    GBAR_SYNC(bar);
    flow_alloc(w, gid, gstride);
    GBAR_SYNC(bar);
    flow_fill(w, gid, gstride);
    GBAR_SYNC(bar);
    flow_rank(w, gid, gstride);
    GBAR_SYNC(bar);
    gbar_end(bar);
}

// ---- the step ---------------------------------------------------------------------------------------
RP_DEV unsigned flow_epoch(const DevWorld &w) { return (((unsigned)w.flags[FL_SEQ] % 1023u) + 1u) << FLOW_TICKET_BITS; }
// S0 for every body of the global path (the body half of k_begin_generate): records start at ticket 0 of this step's epoch
__global__ void k_flow_begin(DevWorld w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { w.flags[FL_FLOW_DIRTY] = 0; w.flags[FL_FLOW_ABORT] = 0; w.flags[FL_ANY_BOUNCY] = 0; }
    if (i >= w.n_bodies || !global_body(w, i)) return;
    V3 lin, ang, trans, incl, inca; Q4 rot;
    body_begin(w, i, lin, ang, rot, trans, incl, inca);
    const float tag = __int_as_float((int)flow_epoch(w));
    w.s_inca[i] = f4(inca, 0.0f); w.s_incl[i] = f4(incl, 0.0f);
    w.f_rec[2 * i] = f4(lin, tag); w.f_rec[2 * i + 1] = f4(ang, tag);
    w.s_rot[i] = f4(rot); w.s_trans[i] = f4(trans, 0.0f);
}

// kinematic bodies: the rule of g_kinematic_writeback on values handed over by the caller
__device__ __noinline__ void flow_kinematic_writeback(KinWb k, int i, int type, V3 slin, V3 sang, Q4 rot, V3 trans) {
    float4 damp = k.b_damp[i];
    float dt = k.dt;
    V3 lin = slin * (1.0f / (1.0f + dt * damp.x));
    V3 ang = sang * (1.0f / (1.0f + dt * damp.y));
    V3 lcom = v3(k.b_lcom_invm[i]);
    V3 t = trans + qrot(rot, -lcom);
    if (type == RP_BODY_KINEMATIC_POSITION) { rot = q4(k.b_next_rot[i]); t = v3(k.b_next_pos[i]); }
    bool finite = isfinite(t.x) && isfinite(t.y) && isfinite(t.z) && isfinite(rot.x) && isfinite(rot.y) && isfinite(rot.z) && isfinite(rot.w) &&
                  isfinite(lin.x) && isfinite(lin.y) && isfinite(lin.z) && isfinite(ang.x) && isfinite(ang.y) && isfinite(ang.z);
    if (!finite) { atomicAdd(&k.flags[FL_QUARANTINE], 1); k.b_quar[i] = 1; k.b_linvel[i] = make_float4(0, 0, 0, 0); k.b_angvel[i] = make_float4(0, 0, 0, 0); return; }
    k.b_linvel[i] = f4(lin, 0.0f); k.b_angvel[i] = f4(ang, 0.0f);
    k.b_pos[i] = f4(t, 0.0f); k.b_rot[i] = f4(rot);
    k.b_next_pos[i] = f4(t, 0.0f); k.b_next_rot[i] = f4(rot);
    k.b_wcom[i] = f4(qrot(rot, lcom) + t, 0.0f);
}

// The per-item wait-and-apply loop.  `has`: this lane holds an item; i1 / i2 (-1 = none) with expected tags e1 / e2.
// `APPLY` runs once per lane, as soon as that lane's records are ready; lanes do not wait for each other.
#define FLOW_RUN(has, i1, e1, i2, e2, ...)                                                       \
    do {                                                                                         \
        bool pending_ = (has) && !cx.dead;                                                       \
        unsigned spins_ = 0;                                                                     \
        cx.n_items += __any(pending_) ? 1u : 0u;                                                 \
        const long long tw0_ = (long long)wall_clock64();                                        \
        unsigned skip_ = 0;                                                                      \
        while (__any(pending_)) {                                                                \
            unsigned gap_ = 4096u;                                                               \
            Vel v1, v2;                                                                          \
            const bool poll_ = pending_ && skip_ == 0u; /* a lane far from its ticket sits rounds out: its loads would only add to the request stream */ \
            cx.n_polls += __any(poll_) ? 1u : 0u;                                                \
            if (poll_) { gap_ = flow_poll(cx, (i1), (e1), (i2), (e2), v1, v2); skip_ = flow_skip(gap_); } else if (skip_) skip_--; \
            const bool ready_ = gap_ == 0u;                                                      \
            if (__any(ready_) && skip_ > 2u) skip_ = 2u; /* neighbours become ready together */  \
            cx.n_applies += __any(ready_) ? 1u : 0u;                                             \
            if (ready_) { __VA_ARGS__; pending_ = false; }                                       \
            if (!__any(ready_) && !flow_backoff(cx, spins_)) break; \
        }                                                                                        \
        cx.t_wait += (long long)wall_clock64() - tw0_;                                           \
    } while (0)

// The wait loop of a contact application.  POSE_MODE 0: the sweep reads no pose.  2: the poses it reads become final when both
// records have passed an integrate ticket (p1 / p2: this substep's integrate for the relaxed sweep, last substep's for update +
// warm start) — they are fetched at that moment, normally several events before this application's own ticket (for the warm
// start: in the first poll round), so their latency is off the hand-off chain.
#define FLOW_RUN_CONTACT(A, has, e1, e2, POSE_MODE, p1, p2, ...)                                 \
    do {                                                                                         \
        bool pending_ = (has) && !cx.dead, posed_ = (POSE_MODE) != 2;                               \
        unsigned spins_ = 0;                                                                     \
        cx.n_items += __any(pending_) ? 1u : 0u;                                                 \
        const long long tw0_ = (long long)wall_clock64();                                        \
        if (pending_ && (POSE_MODE) == 1) (A).load_poses();                                        \
        unsigned skip_ = 0;                                                                      \
        while (__any(pending_)) {                                                                \
            unsigned gap_ = 4096u;                                                               \
            Vel v1, v2;                                                                          \
            const bool poll_ = pending_ && skip_ == 0u;                                          \
            cx.n_polls += __any(poll_) ? 1u : 0u;                                                \
            if (!poll_ && skip_) skip_--;                                                        \
            if (poll_) {                                                                         \
                gap_ = flow_poll(cx, (A).i1, (e1), (A).i2, (e2), v1, v2);                        \
                skip_ = flow_skip(gap_);                                                         \
                if (!posed_) {                                                                   \
                    const unsigned g1_ = (A).i1 >= 0 ? (e1) - (p1) : 0xffffffffu, g2_ = (A).i2 >= 0 ? (e2) - (p2) : 0xffffffffu; /* events between integrate and this one */ \
                    if (gap_ <= (g1_ < g2_ ? g1_ : g2_)) { (A).load_poses(); posed_ = true; }    \
                }                                                                                \
            }                                                                                    \
            const bool ready_ = gap_ == 0u && posed_;                                            \
            if (__any(ready_) && skip_ > 2u) skip_ = 2u;                                         \
            cx.n_applies += __any(ready_) ? 1u : 0u;                                             \
            if (ready_) { (A).v1 = v1; (A).v2 = v2; __VA_ARGS__; (A).finish(); pending_ = false; } \
            if (!__any(ready_) && !flow_backoff(cx, spins_)) break; \
        }                                                                                        \
        cx.t_wait += (long long)wall_clock64() - tw0_;                                           \
    } while (0)

// JOINTS = false: the instantiation for worlds without impulse joints carries none of the joint code (its row arrays live in
// scratch memory, which a launch pays for whether or not a joint exists)
template <bool COUL, bool JOINTS>
__global__ void __launch_bounds__(256) k_global_flow(DevWorld w, int has_restitution) {
    // Workgroups are dedicated to ONE kind of event — body events (increment, integrate, write-back), joints, or contacts: a
    // wavefront walks its phases in order, so a wavefront that served both bodies and contacts would hold its contacts' warm start
    // back until the slowest of its (unrelated) bodies could be incremented, and chain such artificial waits across the scene.
    // The grid is split in proportion to the three item counts (one item per thread and phase where the grid allows: a thread that
    // holds two items of a phase chains the second behind the first for nothing — b3d_joint_grid's 19,800 joints on a quarter of
    // the grid did).
    const int G = gridDim.x;
    int GB, GJ;
    {
        int Mq = w.flags[FL_N_CONS]; if (Mq > w.cons_cap) Mq = w.cons_cap;
        const int needB = max(1, (w.n_bodies + 255) / 256), needJ = (JOINTS && w.n_joints > 0) ? max(1, (flow_live_joints(w) + 255) / 256) : 0;
        const int needC = max(1, (Mq + 255) / 256), total = needB + needJ + needC;
        if (total <= G) { GB = needB; GJ = needJ; }
        else { GB = max(1, (int)((long long)needB * G / total)); GJ = needJ ? max(1, (int)((long long)needJ * G / total)) : 0; }
        if (GB + GJ > G - 1) { GB = max(1, (G - 1) / 2); GJ = needJ ? max(1, G - 1 - GB) : 0; } // a tiny grid still keeps one contact workgroup
    }
    const int bid = blockIdx.x;
    const int role = bid < GB ? 0 : (bid < GB + GJ ? 1 : 2); // 0 bodies, 1 joints, 2 contacts
    const int rb0 = role == 0 ? 0 : (role == 1 ? GB : GB + GJ), rbn = role == 0 ? GB : (role == 1 ? GJ : G - GB - GJ);
    const int T = rbn * blockDim.x, tid = (bid - rb0) * blockDim.x + threadIdx.x;
    int M = w.flags[FL_N_CONS]; if (M > w.cons_cap) M = w.cons_cap;
    const int nb = w.n_bodies, njl = JOINTS ? flow_live_joints(w) : 0;
    const rp_integration_params &prm = w.prm.p;
    const bool fib = prm.friction_in_bias_pass || prm.num_internal_stabilization_iterations == 0;
    FlowSched sc = {prm.num_internal_pgs_iterations, prm.num_internal_stabilization_iterations, w.prm.num_substeps, has_restitution};
    FlowCtx cx;
    cx.B.rec = flow_rsrc(w.f_rec, 2 * nb); cx.B.rot = flow_rsrc(w.s_rot, nb); cx.B.trans = flow_rsrc(w.s_trans, nb);
    cx.epoch = flow_epoch(w);
#ifdef RP_FLOW_TRACE
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_flow_trace = w.dbg; g_flow_trace_body = w.n_bodies / 2; }
#endif
    cx.abort_flag = &w.flags[FL_FLOW_ABORT]; cx.ovf_flag = &w.flags[FL_OVERFLOW];
    cx.t0 = (long long)wall_clock64();
    cx.n_items = 0; cx.n_polls = 0; cx.n_applies = 0; cx.t_apply = 0; cx.t_wait = 0;
    cx.dead = (w.flags[FL_OVERFLOW] & RP_OVF_FLOW) != 0; // an earlier failure (or a ticket range overflow): no waiting at all
    const unsigned ep = cx.epoch;

    // ---- S1 generate: this thread's constraints (start-of-step state only) ----
    if (role == 2) for (int pos = tid; pos < M; pos += T) {
        int s = w.cons_pair[pos], id1, id2;
        flow_ids(w, pos, id1, id2);
        int2 rk = w.fk_rank[pos];
        if ((id1 >= 0) != (rk.x >= 0) || (id2 >= 0) != (rk.y >= 0)) { // the toucher lists are stale: never wait on them
            __hip_atomic_store(cx.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicOr(cx.ovf_flag, RP_OVF_FLOW);
        }
        if (COUL) coul_generate(w, FlowGenAcc(w, pos), s, id1, id2, id1, id2); else cons_generate(w, FlowGenAcc(w, pos), s, id1, id2, id1, id2);
    }

    for (int sub = 0; sub < sc.nsub; ++sub) {
        const float solved_dt = (float)sub * w.prm.dt_sub;
        // ---- S2 increment (+ gyroscopic term): the owner of each body ----
        if (role == 0) for (int base = 0; base < nb; base += T) {
            const int i = base + tid;
            const bool has = i < nb && global_body(w, i);
            int2 dg = has ? w.fb_deg[i] : make_int2(0, 0);
            const unsigned e = ep | (unsigned)fs_incr(sc, dg.x, dg.y, sub);
            FLOW_RUN(has, i, e, -1, 0u, {
                V3 lin = v1.lin, ang = v1.ang;
                body_increment(w, w.b_flags[i], lin, ang, flow_q4(flow_ld(cx.B.rot, i)), v3(w.s_incl[i]), v3(w.s_inca[i]), v3(w.b_invpi[i]), q4(w.b_pframe[i]));
                flow_st_vel(cx.B, i, lin, ang, e + 1u);
            });
        }
        // ---- joint rows from the current poses (off the ticket chain: only needs last substep's integrate) ----
        if (JOINTS && role == 1) for (int base = 0; base < njl; base += T) {
            const int idx = base + tid;
            const bool has = idx < njl;
            int j = 0, b1 = -1, b2 = -1, need1 = 0, need2 = 0;
            if (has) {
                j = w.j_order[idx]; b1 = w.j_b1[j]; b2 = w.j_b2[j];
                if (sub > 0) {
                    if (b1 >= 0) { int2 d = w.fb_deg[b1]; need1 = fs_integ(sc, d.x, d.y, sub - 1) + 1; }
                    if (b2 >= 0) { int2 d = w.fb_deg[b2]; need2 = fs_integ(sc, d.x, d.y, sub - 1) + 1; }
                }
            }
            bool pending = has && !cx.dead;
            unsigned spins = 0;
            while (__any(pending)) {
                bool ready = false;
                if (pending) ready = flow_reached(cx, b1, need1) && flow_reached(cx, b2, need2);
                if (ready) { FlowJointIO io = {cx}; joint_update_one_t(w, io, j, sub); pending = false; }
                if (!__any(ready) && !flow_backoff(cx, spins)) break;
            }
        }
        // ---- update + warm start of every contact, in sweep order per body ----
        if (role == 2) for (int base = 0; base < M; base += T) {
            const int pos = base + tid;
            const bool has = pos < M;
            int i1 = -1, i2 = -1, nn = 0; unsigned e1 = 0, e2 = 0, pt1 = 0, pt2 = 0;
            if (has) {
                i1 = w.k_b1[pos]; i2 = w.k_b2[pos]; nn = w.k_n[pos];
                int2 rk = w.fk_rank[pos];
                // the poses this sweep reads are last substep's: final once the record has passed that integrate (at once in substep 0)
                if (i1 >= 0) { int2 d = w.fb_deg[i1]; e1 = ep | (unsigned)fs_ws(sc, d.x, d.y, sub, rk.x); pt1 = ep | (unsigned)(sub > 0 ? fs_integ(sc, d.x, d.y, sub - 1) + 1 : 0); }
                if (i2 >= 0) { int2 d = w.fb_deg[i2]; e2 = ep | (unsigned)fs_ws(sc, d.x, d.y, sub, rk.y); pt2 = ep | (unsigned)(sub > 0 ? fs_integ(sc, d.x, d.y, sub - 1) + 1 : 0); }
            }
            FlowAccT<!COUL && !JOINTS> A(w, cx, pos, i1, i2, nn, e1 + 1u, e2 + 1u);
            FLOW_RUN_CONTACT(A, has, e1, e2, 2, pt1, pt2, cons_apply_model<COUL>(w, A, MODE_WARMSTART, fib, solved_dt));
        }
        // ---- biased sweeps: every joint before any contact ----
        for (int it = 0; it < sc.npgs; ++it) {
            if (JOINTS && role == 1) for (int base = 0; base < njl; base += T) {
                const int idx = base + tid;
                const bool has = idx < njl;
                int j = 0, i1 = -1, i2 = -1; unsigned e1 = 0, e2 = 0;
                if (has) {
                    j = w.j_order[idx]; i1 = w.j_b1[j]; i2 = w.j_b2[j];
                    int2 rk = w.fj_rank[j];
                    if (i1 >= 0) { int2 d = w.fb_deg[i1]; e1 = ep | (unsigned)fs_jbias(sc, d.x, d.y, sub, it, rk.x); }
                    if (i2 >= 0) { int2 d = w.fb_deg[i2]; e2 = ep | (unsigned)fs_jbias(sc, d.x, d.y, sub, it, rk.y); }
                }
                // the joint's rows are this thread's own: fetched now, before the wait for its bodies' tickets, they are off the hand-off chain
                int nrows = 0; V3 jim1 = v3(0, 0, 0), jim2 = jim1; JointRowsT<FLOW_JROWS> R0;
                if (has) { nrows = joint_row_count(w.j_locked[j], w.j_limited[j], w.j_motor[j]); jim1 = v3(JRP(JR_IM1, j)); jim2 = v3(JRP(JR_IM2, j)); jrows_load<FLOW_JROWS>(w, j, 0, nrows, R0); }
                FLOW_RUN(has, i1, e1, i2, e2, {
                    FlowJointIO io = {cx, {v1, v2}, {e1 + 1u, e2 + 1u}};
                    joint_solve_fetched<FlowJointIO, FLOW_JROWS>(w, io, j, i1, i2, nrows, jim1, jim2, R0, false, prm.warmstart_joints && it == 0);
                });
            }
            if (role == 2) for (int base = 0; base < M; base += T) {
                const int pos = base + tid;
                const bool has = pos < M;
                int i1 = -1, i2 = -1, nn = 0; unsigned e1 = 0, e2 = 0;
                if (has) {
                    i1 = w.k_b1[pos]; i2 = w.k_b2[pos]; nn = w.k_n[pos];
                    int2 rk = w.fk_rank[pos];
                    if (i1 >= 0) { int2 d = w.fb_deg[i1]; e1 = ep | (unsigned)fs_cbias(sc, d.x, d.y, sub, it, rk.x); }
                    if (i2 >= 0) { int2 d = w.fb_deg[i2]; e2 = ep | (unsigned)fs_cbias(sc, d.x, d.y, sub, it, rk.y); }
                }
                FlowAccT<!COUL && !JOINTS> A(w, cx, pos, i1, i2, nn, e1 + 1u, e2 + 1u);
                FLOW_RUN_CONTACT(A, has, e1, e2, 0, 0u, 0u, cons_apply_model<COUL>(w, A, MODE_BIAS, fib, solved_dt));
            }
        }
        // ---- S6 integrate: poses first (write-through), then the record that announces them ----
        if (role == 0) for (int base = 0; base < nb; base += T) {
            const int i = base + tid;
            const bool has = i < nb && global_body(w, i);
            int2 dg = has ? w.fb_deg[i] : make_int2(0, 0);
            const unsigned e = ep | (unsigned)fs_integ(sc, dg.x, dg.y, sub);
            FLOW_RUN(has, i, e, -1, 0u, {
                V3 lin = v1.lin, ang = v1.ang, trans = flow_v3(flow_ld(cx.B.trans, i)); Q4 rot = flow_q4(flow_ld(cx.B.rot, i));
                body_integrate(w, w.b_flags[i], lin, ang, rot, trans);
                flow_st4(cx.B.rot, i, f4(rot)); flow_st4(cx.B.trans, i, f4(trans, 0.0f));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                flow_st_vel(cx.B, i, lin, ang, e + 1u);
            });
        }
        // ---- relaxed sweeps (refresh_rhs_wo_bias + solve with friction) ----
        for (int it = 0; it < sc.nstab; ++it) {
            if (JOINTS && role == 1) for (int base = 0; base < njl; base += T) {
                const int idx = base + tid;
                const bool has = idx < njl;
                int j = 0, i1 = -1, i2 = -1; unsigned e1 = 0, e2 = 0;
                if (has) {
                    j = w.j_order[idx]; i1 = w.j_b1[j]; i2 = w.j_b2[j];
                    int2 rk = w.fj_rank[j];
                    if (i1 >= 0) { int2 d = w.fb_deg[i1]; e1 = ep | (unsigned)fs_jrelax(sc, d.x, d.y, sub, it, rk.x); }
                    if (i2 >= 0) { int2 d = w.fb_deg[i2]; e2 = ep | (unsigned)fs_jrelax(sc, d.x, d.y, sub, it, rk.y); }
                }
                int nrows = 0; V3 jim1 = v3(0, 0, 0), jim2 = jim1; JointRowsT<FLOW_JROWS> R0;
                if (has) { nrows = joint_row_count(w.j_locked[j], w.j_limited[j], w.j_motor[j]); jim1 = v3(JRP(JR_IM1, j)); jim2 = v3(JRP(JR_IM2, j)); jrows_load<FLOW_JROWS>(w, j, 0, nrows, R0); }
                FLOW_RUN(has, i1, e1, i2, e2, {
                    FlowJointIO io = {cx, {v1, v2}, {e1 + 1u, e2 + 1u}};
                    joint_solve_fetched<FlowJointIO, FLOW_JROWS>(w, io, j, i1, i2, nrows, jim1, jim2, R0, true, false);
                });
            }
            if (role == 2) for (int base = 0; base < M; base += T) {
                const int pos = base + tid;
                const bool has = pos < M;
                int i1 = -1, i2 = -1, nn = 0; unsigned e1 = 0, e2 = 0, pt1 = 0, pt2 = 0;
                if (has) {
                    i1 = w.k_b1[pos]; i2 = w.k_b2[pos]; nn = w.k_n[pos];
                    int2 rk = w.fk_rank[pos];
                    if (i1 >= 0) { int2 d = w.fb_deg[i1]; e1 = ep | (unsigned)fs_crelax(sc, d.x, d.y, sub, it, rk.x); pt1 = ep | (unsigned)(fs_integ(sc, d.x, d.y, sub) + 1); }
                    if (i2 >= 0) { int2 d = w.fb_deg[i2]; e2 = ep | (unsigned)fs_crelax(sc, d.x, d.y, sub, it, rk.y); pt2 = ep | (unsigned)(fs_integ(sc, d.x, d.y, sub) + 1); }
                }
                FlowAccT<!COUL && !JOINTS> A(w, cx, pos, i1, i2, nn, e1 + 1u, e2 + 1u);
                FLOW_RUN_CONTACT(A, has, e1, e2, 2, pt1, pt2, cons_apply_model<COUL>(w, A, MODE_RELAX, fib, solved_dt + w.prm.dt_sub));
            }
        }
    }
    // ---- S8 restitution (a no-op for constraints without a seed) ----
    if (has_restitution && role == 2) {
        for (int base = 0; base < M; base += T) {
            const int pos = base + tid;
            const bool has = pos < M;
            int i1 = -1, i2 = -1, nn = 0; unsigned e1 = 0, e2 = 0;
            if (has) {
                i1 = w.k_b1[pos]; i2 = w.k_b2[pos]; nn = w.k_n[pos];
                int2 rk = w.fk_rank[pos];
                if (i1 >= 0) { int2 d = w.fb_deg[i1]; e1 = ep | (unsigned)fs_rest(sc, d.x, d.y, rk.x); }
                if (i2 >= 0) { int2 d = w.fb_deg[i2]; e2 = ep | (unsigned)fs_rest(sc, d.x, d.y, rk.y); }
            }
            FlowAccT<!COUL && !JOINTS> A(w, cx, pos, i1, i2, nn, e1 + 1u, e2 + 1u);
            FLOW_RUN_CONTACT(A, has, e1, e2, 0, 0u, 0u, cons_apply_model<COUL>(w, A, MODE_RESTITUTION, fib, 0.0f));
        }
    }
    // ---- S9 impulse write-back (this thread's own constraints and joints), S10 body write-back ----
    if (!cx.dead && role == 2) {
        for (int pos = tid; pos < M; pos += T) { if (COUL) coul_writeback(w, GlobalAcc(w, pos), w.cons_pair[pos]); else cons_writeback(w, GlobalAcc(w, pos), w.cons_pair[pos]); }
    }
    if (JOINTS && !cx.dead && role == 1) for (int idx = tid; idx < njl; idx += T) joint_writeback_one(w, w.j_order[idx]);
    if (role == 0) for (int base = 0; base < nb; base += T) {
        const int i = base + tid;
        const bool has = i < nb && global_body(w, i);
        int2 dg = has ? w.fb_deg[i] : make_int2(0, 0);
        const unsigned e = ep | (unsigned)fs_final(sc, dg.x, dg.y);
        FLOW_RUN(has, i, e, -1, 0u, {
            V3 trans = flow_v3(flow_ld(cx.B.trans, i)); Q4 rot = flow_q4(flow_ld(cx.B.rot, i));
            const int type = w.b_flags[i] & RP_BF_TYPE_MASK;
            if (type == RP_BODY_DYNAMIC) body_writeback(w, i, v1.lin, v1.ang, rot, trans);
            else {
                KinWb k = {w.b_damp, w.s_lin, w.s_ang, w.s_rot, w.s_trans, w.b_lcom_invm, w.b_next_rot, w.b_next_pos, w.b_linvel, w.b_angvel, w.b_pos, w.b_rot, w.b_wcom, w.flags, w.b_quar, w.prm.p.dt};
                flow_kinematic_writeback(k, i, type, v1.lin, v1.ang, rot, trans);
            }
        });
    }
    if ((threadIdx.x & 63) == 0 && cx.n_items) { // hand-off statistics of this wavefront (rp_debug_cycles slots 20..23)
        atomicAdd((unsigned long long *)&w.dbg[20], (unsigned long long)cx.n_items);
        atomicAdd((unsigned long long *)&w.dbg[21], (unsigned long long)cx.n_polls);
        atomicAdd((unsigned long long *)&w.dbg[22], (unsigned long long)cx.n_applies);
        atomicAdd((unsigned long long *)&w.dbg[23], 1ull);
        atomicAdd((unsigned long long *)&w.dbg[24], (unsigned long long)cx.t_apply);
        atomicAdd((unsigned long long *)&w.dbg[25], (unsigned long long)cx.t_wait);
        atomicAdd((unsigned long long *)&w.dbg[26], (unsigned long long)((long long)wall_clock64() - cx.t0));
    }
}

// the step is retired (and the hint record published) by its own small launch: every wave of k_global_flow reads FL_SEQ for its epoch
__global__ void k_flow_retire(DevWorld w) {
    if (threadIdx.x == 0) { w.flags[FL_SEQ] += 1; if (!(w.flags[FL_OVERFLOW] & RP_OVF_FLOW)) w.flags[FL_STEP] += 1; }
    __threadfence(); __syncthreads();
    publish_flags(w);
}

void rp_launch_joint_writeback(const DevWorld &w, hipStream_t st);

// Grid of the dataflow launch: every workgroup must be resident at once.
int rp_flow_grid(int device) {
    static int cached[64] = {0};
    if (device >= 0 && device < 64 && cached[device]) return cached[device];
    hipDeviceProp_t prop;
    int per_cu = 0, cus = 0;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_global_flow<true, true>, 256, 0) != hipSuccess) per_cu = 0;
    int per_cu2 = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, k_global_flow<false, true>, 256, 0) != hipSuccess) per_cu2 = 0;
    if (per_cu2 < per_cu) per_cu = per_cu2;
    int want = 1; // workgroups per CU; stays well below the occupancy answer: the hardware may admit one fewer
    if (want > per_cu - 1 && per_cu > 4) want = per_cu - 1; // near the hardware's own limit the occupancy answer can be one too many
    if (per_cu < 1 || cus < 1) return 0;
    if (want > per_cu) want = per_cu;
    int g = cus * want;
    if (device >= 0 && device < 64) cached[device] = g;
    return g;
}
// per-body toucher lists in sweep order (rebuilt only when FL_FLOW_DIRTY; the flag is cleared by the kernel that starts the solve)
void rp_launch_flow_ranks(const DevWorld &w, hipStream_t st) {
    int n = w.cons_cap > w.n_joints ? w.cons_cap : w.n_joints; if (n < w.n_bodies) n = w.n_bodies;
    int blocks = (n + 255) / 256; if (blocks > w.gbar_blocks) blocks = w.gbar_blocks; if (blocks < 1) blocks = 1; // all resident (grid barriers)
    hipLaunchKernelGGL(k_flow_ranks, dim3(blocks), dim3(1024), 0, st, w);
}
void rp_launch_tiles_build(const DevWorld &w, hipStream_t st);
void rp_launch_global_flow(const DevWorld &w, hipStream_t st, int grid, int has_restitution) {
    int nbb = (w.n_bodies + 255) / 256; if (nbb < 1) nbb = 1;
    rp_launch_flow_ranks(w, st);
    rp_launch_tiles_build(w, st); // worlds that may tile (rp_tiles.hip): the tiling follows the ranks, same gate; once it is valid the host moves the sweeps onto tiles
    hipLaunchKernelGGL(k_flow_begin, dim3(nbb), dim3(256), 0, st, w);
    const bool coul = w.prm.p.friction_model == RP_FRICTION_COULOMB, joints = w.n_joints > 0;
#define FLOW_LAUNCH(C, J) hipLaunchKernelGGL((k_global_flow<C, J>), dim3(grid), dim3(256), 0, st, w, has_restitution)
    if (coul) { if (joints) FLOW_LAUNCH(true, true); else FLOW_LAUNCH(true, false); }
    else { if (joints) FLOW_LAUNCH(false, true); else FLOW_LAUNCH(false, false); }
#undef FLOW_LAUNCH
    hipLaunchKernelGGL(k_flow_retire, dim3(1), dim3(64), 0, st, w);
}

// workgroups of k_flow_ranks (1024 threads) one CU holds at once (0 = the query failed): input of DevWorld::gbar_blocks (rp_api.hip)
int rp_occ_flow_ranks(void) { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_flow_ranks, 1024, 0) != hipSuccess) n = 0; return n; }
