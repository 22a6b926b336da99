// rp_api.hip — the C ABI of librapier_hip.so (include/rapier_hip.h) and the per-step launch driver.
//
// Mirrors PhysicsWorld / PhysicsPipeline::step (physics_world.rs:61-157, physics_pipeline/mod.rs:196-246,
// substep.rs:267-581): detect_collisions -> build_islands_and_solve_velocity_constraints ->
// advance_to_final_positions -> refresh_moved_collider_aabbs.  A step is a fixed sequence of kernel
// launches on the world's HIP stream, captured once into a hipGraph and replayed (the reference's
// per-colour spin barriers become kernel boundaries; graph replay removes the host launch cost).
// The host never needs device-side counts to be correct: kernels size themselves from device
// scalars and the solver's single-workgroup tail absorbs any colour stage the host did not launch.
// Layout hints (parallel colour count, largest stage) are read lazily from pinned memory.
//
// Two graphs exist per world:
//   * the FULL graph: collider poses/AABBs -> broad phase (one launch, rp_gridbar.h) -> narrow phase -> colouring -> sleep pass ->
//     solver-graph / island layout (one launch) -> solver.  Always correct; ~13 launches.
//   * the FAST graph (steady state): k_fast_front [-> k_sleep_pass -> k_sleep_check in sleep-enabled worlds] -> k_island_solve ->
//     k_global_single [-> k_force_events]; when every body lives in an LDS island and nothing else is asked for, the single fused
//     launch of k_island_solve instead.  k_fast_front
//     proves on the device that the broad phase and the narrow phase would be no-ops this step (no fat
//     AABB left, every pair passes its recycle test); if not, it raises FL_FAST_ABORT and the other two
//     kernels exit without touching the world.  The device counts executed steps (FL_STEP); every host
//     entry point that observes the world first replays the missing steps through the FULL graph
//     (settle()).  The physics is identical either way: the fast graph only skips work that the full
//     graph would have found to be empty.
#include "rp_world.h"
#include "rp_polyhedron.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>

// HIP stream capture is process-wide in effect: while ANY thread captures, a synchronising runtime call from another thread
// (hipMemcpy, hipMalloc, hipFree, hipHostMalloc ...) fails and poisons the capture, whatever the capture mode (measured on ROCm
// 7.2: worlds stepped from several host threads broke each other's graph captures).  So the capture sections and those calls take
// one process-wide lock; kernels, async copies on a world's own stream and stream / event waits stay outside it, so worlds still
// overlap on the device.
static std::mutex g_hip_unsafe_api;
// A "synchronous" copy of a world is a copy ON THE WORLD'S STREAM followed by a wait for that stream.  The stream is non-blocking, so the
// legacy-stream hipMemcpy the first three rounds used here was not ordered against work still queued on it (the zero fills of
// finalize(), kernels of an edit in progress): see the note at the b_order upload in finalize().
static inline hipError_t stream_hipMemcpy(hipStream_t st, void *dst, const void *src, size_t n, hipMemcpyKind k) {
    hipError_t e = hipMemcpyAsync(dst, src, n, k, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}
static inline hipError_t locked_hipMallocBytes(void **p, size_t n) { std::lock_guard<std::mutex> g(g_hip_unsafe_api); return hipMalloc(p, n); }
static inline hipError_t locked_hipFree(void *p) { std::lock_guard<std::mutex> g(g_hip_unsafe_api); return hipFree(p); }
static inline hipError_t locked_hipHostMalloc(void **p, size_t n, unsigned flags) { std::lock_guard<std::mutex> g(g_hip_unsafe_api); return hipHostMalloc(p, n, flags); }
static inline hipError_t locked_hipHostFree(void *p) { std::lock_guard<std::mutex> g(g_hip_unsafe_api); return hipHostFree(p); }
#define hipMemcpy(dst, src, n, k) stream_hipMemcpy(w->stream, (void *)(dst), (const void *)(src), (n), (k)) // (every caller has its world in scope as `w`)
#define hipFree(p) locked_hipFree((void *)(p))
#define hipHostFree(p) locked_hipHostFree((void *)(p))

void rp_launch_collider_update(const DevWorld &w, hipStream_t st);
void rp_launch_broadphase(const DevWorld &w, hipStream_t st);
void rp_launch_bp_rehash(const DevWorld &w, hipStream_t st);
void rp_launch_narrowphase(const DevWorld &w, hipStream_t st);
void rp_launch_narrowphase_part(const DevWorld &w, hipStream_t st, int part);
void rp_launch_init_bodies(const DevWorld &w, hipStream_t st);
void rp_launch_solver_assembly(const DevWorld &w, hipStream_t st, int lean);
int rp_launch_solver_loop(const DevWorld &w, hipStream_t st, int parallel_stages, int stage_blocks, int has_restitution, int joint_stages, int tile_grid, int no_contacts_hint);
void rp_launch_solver_writeback(const DevWorld &w, hipStream_t st, int parity, int publish);
bool rp_ccd_launches(const DevWorld &w);
void rp_launch_island_solve(const DevWorld &w, hipStream_t st, int grid, int has_restitution, int fast, int retire, int fused, int dense, int wide);
void rp_launch_global_single(const DevWorld &w, hipStream_t st, int has_restitution, int fast);
void rp_launch_fast_front(const DevWorld &w, hipStream_t st, int no_global_kernel);
void rp_launch_wake(const DevWorld &w, hipStream_t st, int phase);
void rp_launch_wake_partners(const DevWorld &w, hipStream_t st);
void rp_launch_force_events(const DevWorld &w, hipStream_t st, int fast);
void rp_launch_idle_step(const DevWorld &w, hipStream_t st);
void rp_launch_sleep_fast(const DevWorld &w, hipStream_t st);
void rp_launch_sensor_fast(const DevWorld &w, hipStream_t st);
void rp_launch_sensor_check(const DevWorld &w, hipStream_t st);
void rp_launch_rebase_stamps(const DevWorld &w, hipStream_t st, int delta);
void rp_launch_clear_no_contact(const DevWorld &w, hipStream_t st);
void rp_launch_global_flow(const DevWorld &w, hipStream_t st, int grid, int has_restitution);
void rp_launch_ccd(const DevWorld &w, hipStream_t st, int has_bullets, int publish);
void rp_launch_pi_ensure(const DevWorld &w, hipStream_t st, int first, int count, int reset);
void rp_launch_pi_remove_body(const DevWorld &w, hipStream_t st, int b);
void rp_launch_pj_append_joint(const DevWorld &w, hipStream_t st, int dev_joint, int b1, int b2, int key);
int rp_flow_grid(int device);
int rp_occ_bp_rebuild(void); int rp_occ_layout_rebuild(void); int rp_occ_sleep_pass(void); int rp_occ_flow_ranks(void); int rp_occ_tiles_build(void);
// Grid of the kernels that synchronise through device-side grid barriers (rp_gridbar.h): every workgroup must be resident at once, so
// the cap is what THIS device holds of the hungriest of them — CUs x the smallest occupancy answer — with a quarter left free for
// whatever else runs on the device (other worlds' rebuilds, other streams); never more than 192 (more workgroups only lengthen the
// barriers), at least 1 (a single workgroup needs no co-residency at all: the passes are grid-stride).
static int gbar_grid_for_device(int device) {
    static int cached[64] = {0};
    if (device >= 0 && device < 64 && cached[device]) return cached[device];
    hipDeviceProp_t prop;
    int cus = 0;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
    int occ = std::min(std::min(rp_occ_bp_rebuild(), rp_occ_layout_rebuild()), std::min(std::min(rp_occ_sleep_pass(), rp_occ_flow_ranks()), rp_occ_tiles_build()));
    int g = (cus * occ * 3) / 4;
    const char *e = getenv("RP_GBAR_BLOCKS"); // test hook: a small part / a busy device
    if (e && atoi(e) > 0) g = std::min(g > 0 ? g : atoi(e), atoi(e));
    g = std::max(1, std::min(g, 192));
    if (device >= 0 && device < 64) cached[device] = g;
    return g;
}
int rp_fused_grid(int device); int rp_fused_grid_dense(int device);

struct HostBody {
    rp_body_desc d; float inv_mass; float inv_pi[3]; float lcom[3]; int ncolliders; bool removed;
    // RigidBodyActivation state carried across device rebuilds (rp_sleep.hip)
    float pframe[4] = {0, 0, 0, 1};   // principal inertia frame (MassProperties::principal_inertia_local_frame)
    float max_extent = 0.0f, sleep_timer = 0.0f, sprev[7] = {0, 0, 0, 0, 0, 0, 1};
    float ccd_thickness = 3.402823466e+38f; // RigidBodyCcd::ccd_thickness: the thinnest attached shape (Real::MAX without colliders)
    int sleeping = 0, slabel = 0, next_ord = 0;
    std::vector<int> cols;            // its colliders (removed ones included) in attachment order = ascending collider index
    int isl = -1;                     // RigidBodyIds::island_id (persistent islands, rp_sleep.hip), mirrored across device rebuilds
    bool has_next = false; float next[7] = {0, 0, 0, 0, 0, 0, 1}; // RigidBodyPosition::next_position of a kinematic body
    bool quarantined = false;         // disabled by the quarantine (quarantine.rs): inert like a removed body, handle still readable
};

// Device allocations of finalize(): which DevWorld member they back and how they are indexed, so that a world that outgrows its
// capacities (or receives a joint) can move to larger arrays and keep every persistent row: `per` contiguous elements per item,
// `planes` planes of `stride` items each, items indexed by body / collider / pair slot / device joint.
enum { DOM_NONE = 0, DOM_BODY, DOM_COLL, DOM_PAIR, DOM_JOINT, DOM_FIXED /* fixed-size array, carried whole */ };
struct AllocRec { void *ptr; size_t off, elem, per, stride; int planes, dom, fill; };

struct rp_world {
    int device = 0;
    hipStream_t stream = nullptr;
    rp_integration_params params;
    float gravity[3];
    std::vector<HostBody> bodies;
    std::vector<rp_collider_desc> colliders;
    std::vector<int> collider_parent, collider_ord; // ord: ordinal among the colliders of the same parent (attachment order)
    std::vector<int> collider_sub; int cur_sub = 0, n_sub = 1; // sub-worlds (rp_world_begin_subworld): the one every collider was inserted into
    int next_free_ord = 0;
    // Arena slots (data/arena.rs:28-90, 260-380): a removed body / collider slot is handed out again, LIFO, before a fresh index is;
    // a handle = generation << 32 | index, the generation being the arena's removal count at insertion time
    std::vector<uint32_t> body_gen, coll_gen; uint32_t body_arena_gen = 0, coll_arena_gen = 0;
    std::vector<int> body_free, coll_free;
    bool dead_pairs_possible = false; // colliders were removed since the last step: their pairs are still in the device pair set
    std::vector<char> collider_removed, joint_removed;
    // convex polyhedra (rp_polyhedron.h): registered shapes, the polyhedron of every collider row (-1: another shape), their device tables
    std::vector<HostPolyhedron> polys; std::vector<int> collider_poly; bool polys_uploaded = false;
    // composite shapes (rp_compound_create / rp_trimesh_create / rp_heightfield_create; device twin: rp_composite.h)
    struct HostComposite {
        int kind = 0;                                   // RP_SHAPE_COMPOUND | RP_SHAPE_TRIMESH
        std::vector<rp_collider_desc> parts; std::vector<int> part_poly; // compound: the part descriptors (translation recentred, quaternion normalised)
        std::vector<float> tri;                         // mesh: 9 floats per triangle (vertices recentred)
        std::vector<float> smin, smax;                  // 3 floats per sub-shape: its AABB in the composite frame
        float centre[3] = {0, 0, 0}, half[3] = {0, 0, 0};
        int count() const { return kind == RP_SHAPE_COMPOUND ? (int)parts.size() : (int)(tri.size() / 9); }
    };
    std::vector<HostComposite> comps; std::vector<int> collider_comp;
    void *cm_dev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void *cv_dev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<rp_joint_desc> joints;
    std::vector<int> active_joint_ids; // device joint index -> index into `joints`
    std::vector<int> quarantine_log;   // bodies disabled by the quarantine, in detection order
    int quar_seen = 0;                 // value of FL_QUARANTINE the host has already acted on
    // persistent-island tables saved by download_state() for a rebuild without carry-over (friction-model change)
    struct { bool valid = false; std::vector<int> used, nb, dirty, denied, sleeping, freel, stats; unsigned long long w64[4] = {0, 0, 0, 0}; int next = 0, nfree = 0, pending = 0; } pi_saved;
    std::vector<int> pending_wake;     // bodies to wake once the device world exists again (joints inserted: insert(.., wake_up = true))
    bool finalized = false;
    int cap_bodies = 0, cap_colliders = 0; // device array capacities (rows beyond n_bodies / n_colliders are spare)
    bool hints_valid = false;
    std::vector<AllocRec> allocs;
    DevWorld dw;
    int *pinned_flags = nullptr; // FL_COUNT ints, written by an async D2H copy at the end of each step
    // growth with state carry-over (grow_begin -> finalize -> carry_over): the previous device world, kept until its rows were copied
    bool carry = false;
    std::vector<AllocRec> old_allocs;
    DevWorld old_dw;
    int *old_pinned = nullptr;
    std::vector<int> old_active_joint_ids;
    unsigned *d_speed = nullptr; // rp_world_max_linear_speed's reduction cell
    float guard_horizon = 0.0f;  // rp_world_set_shard_guard_horizon
    // shard guard (rp_world_set_shard_guard): host copy, re-uploaded whenever the device world is rebuilt
    std::vector<float4> guard_min, guard_max; std::vector<int> guard_start, guard_items; float guard_origin[3] = {0, 0, 0}, guard_cell = 0.0f; int guard_dims[3] = {0, 0, 0};
    // launch plan + graph
    int plan_stages = 0, plan_blocks = 1, plan_single = 1, plan_island_grid = 1, plan_joint_stages = 0, plan_no_global = 0, plan_fused = 0, plan_tile_grid = 0, plan_no_contacts = 0, plan_bare = 0, plan_dense = 0, plan_wide = 0;
    bool has_restitution = false;
    // [0] = full path, [1] = fast path, [2] = lean path; "whole" = one graph per step, col/loop/fin = timed thirds (full / fast only)
    hipGraph_t g_whole[3] = {nullptr, nullptr, nullptr}, g_col[3] = {nullptr, nullptr, nullptr}, g_loop[3] = {nullptr, nullptr, nullptr}, g_fin[3] = {nullptr, nullptr, nullptr};
    hipGraphExec_t ge_whole[3] = {nullptr, nullptr, nullptr}, ge_col[3] = {nullptr, nullptr, nullptr}, ge_loop[3] = {nullptr, nullptr, nullptr}, ge_fin[3] = {nullptr, nullptr, nullptr};
    int graph_stages = -1, graph_blocks = -1, graph_single = -1, graph_island_grid = -1, graph_dense = -1, graph_wide = -1, graph_joint_stages = -1, graph_no_global = -1, graph_fused = -1, graph_tile_grid = -1, graph_no_contacts = -1, graph_bare = -1;
    bool use_graph = true, use_fast = true, use_fused = true;
    bool auto_dense = true;        // RP_ISL_DENSE=0: never
    bool force_dense = false;      // RP_ISL_DENSE=1 (tests): the dense form of k_island_solve whatever the island count
    bool use_lean = true;          // the lean step graph of MULTI-mode worlds (below: "lean graph"); RP_NO_LEAN=1: never
    int cur_lean = 0;              // the enqueue_* callbacks capture / launch the lean graph (with dw_lean)
    DevWorld dw_lean;              // dw with lean = 1: the kernel argument of a lean graph's launches
    long long lean_steps = 0;      // lean graphs enqueued
    long long lean_backoff = 3;    // full steps after a lean step died (doubles per death up to 256, back to 3 after 64 clean lean steps)
    int lean_streak = 0; bool lean_death_seen = false;
    bool use_flow = true;          // the dataflow launch (rp_flow.hip) is available; RP_NO_FLOW=1: never, RP_FLOW=1: for every large world
    bool force_flow = false;
    int flow_grid = 0;             // workgroups of the dataflow launch (all resident at once), 0 = unavailable
    int fused_grid = 0;            // most workgroups a fused fast step may use (all resident at once), 0 = no fused step on this device
    int fused_grid_dense = 0;      // the same for the dense form of k_island_solve (two islands per CU), 0 = that form is not used
    bool has_bullets = false;      // some dynamic body has ccd_enabled: the continuous-collision pass runs its second tier
    float min_ccd_thickness = 3.402823466e+38f; // thinnest dynamic body (the fused single-kernel step needs it above the fat-AABB margin)
    bool compound = false;         // some dynamic body carries several colliders or an offset collider (no fused fast step)
    bool timed_ready[2] = {false, false};
    bool never_stepped = true;     // no step has retired and the device world was never rebuilt / grown: what the host mirrors hold is the whole state
    int pairs_scale = 1;           // the pair pool holds RP_PAIRS_PER_COLLIDER x pairs_scale slots per collider row: doubled when the pool fills up (rp_step)
    int cur_fast = 0;              // mode the enqueue_* callbacks capture
    long long steps_requested = 0; // steps asked for since finalize (device FL_STEP counts the executed ones)
    int rebase_at = 1 << 30;       // FL_STEP beyond which the device's 32-bit step stamps move back (k_rebase_stamps)
    long long rebases = 0;
    long long seq_enqueued = 0;    // step graphs enqueued since finalize (device FL_SEQ counts the retired ones)
    long long full_until = 0;      // stay on the full graph until this many steps were requested
    long long eager_until = 0;     // launch the kernels directly until this many steps were requested: a world that is being edited (bodies /
                                   // colliders / joints coming and going every few steps) would re-capture its graphs — ~10 ms — after every edit
    long long fast_steps = 0, full_steps = 0, replayed_steps = 0, fused_steps = 0;
    // timers
    bool timers = false;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double acc_isl_ms = 0.0, acc_glob_ms = 0.0, acc_col_ms = 0.0, acc_step_ms = 0.0;
    double acc_bp_ms = 0.0, acc_np_ms = 0.0, acc_islc_ms = 0.0; int acc_full_steps = 0; // full steps only: broad phase, narrow phase, island construction
    int acc_steps = 0;
    double loop_ms_since_read = 0.0; int loop_steps_since_read = 0;
    std::string err;
};

#define HIPCHK(w, expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            (w)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                                 \
            return RP_ERR_DEVICE;                                                                         \
        }                                                                                                 \
    } while (0)

static int settle(rp_world *w);
static int upload_body_row(rp_world *w, int i);
static int upload_body_row_mass(rp_world *w, int i);
static int upload_collider_row(rp_world *w, int i);
static int upload_collider_chain(rp_world *w, int parent);
static int after_topology_edit(rp_world *w, bool keep_grid = false);
static bool world_sleep_enabled(const rp_world *w);
static bool world_has_kinematic_pos(const rp_world *w);
static bool world_has_force_events(const rp_world *w);
static bool world_has_sensors(const rp_world *w);
static void refresh_ccd_facts(rp_world *w);
static bool world_has_compound_bodies(const rp_world *w);
static std::vector<unsigned long long> no_contact_keys(const rp_world *w);
static int check_sleep_scope(rp_world *w);
static int rebuild_begin(rp_world *w);
static int grow_begin(rp_world *w);
static int refresh_joint_frames(rp_world *w, int b);
static int carry_over(rp_world *w);
static int queue_wake(rp_world *w, int b, int lvl);
static int finalize(rp_world *w);
static int quarantine_body_at(rp_world *w, int b);
static bool all_finite(const float *v, int n);
static int handle_index(uint64_t h);
static int body_of(const rp_world *w, uint64_t h, bool allow_removed = false);
static int collider_of(const rp_world *w, uint64_t h);
static int reset_row(rp_world *w, int dom, int i);
static int purge_dead_pairs(rp_world *w);
template <typename T> static int poke(rp_world *w, T *dst, const T &v);
template <typename T> static int poke(rp_world *w, T *dst, const T &v);

extern "C" void rp_default_params(rp_integration_params *p) {
    // IntegrationParameters::default() — integration_parameters.rs:379-408
    p->dt = 1.0f / 60.0f;
    p->contact_natural_frequency = 30.0f; p->contact_damping_ratio = 10.0f;
    p->static_contact_natural_frequency = 60.0f; p->static_contact_damping_ratio = 10.0f;
    p->joint_natural_frequency = 1.0e6f; p->joint_damping_ratio = 1.0f;
    p->warmstart_coefficient = 1.0f;
    p->normalized_allowed_linear_error = 0.005f;
    p->normalized_max_corrective_velocity = 3.0f;
    p->normalized_prediction_distance = 0.02f;
    p->normalized_max_linear_velocity = 400.0f;
    p->normalized_contact_recycle_distance = 0.05f;
    p->length_unit = 1.0f;
    p->num_solver_iterations = 4;
    p->num_internal_pgs_iterations = 1;
    p->num_internal_stabilization_iterations = 1;
    p->contact_recycling = 1;
    p->friction_in_bias_pass = 0;
    p->friction_model = RP_FRICTION_SIMPLIFIED;
    p->warmstart_joints = 0;
    p->max_ccd_substeps = 1;
}

// SpringCoefficients::{erp_inv_dt, cfm_factor} — integration_parameters.rs:86-149 (f32, no FMA)
static float spring_erp_inv_dt(float freq, float damping, float dt) {
    volatile float ang_freq = freq * 6.283185307179586f;
    volatile float a = dt * ang_freq;
    volatile float b = 2.0f * damping;
    volatile float den = a + b;
    return ang_freq / den;
}
static float spring_cfm_factor(float freq, float damping, float dt) {
    volatile float erp = dt * spring_erp_inv_dt(freq, damping, dt);
    volatile float cfm_coeff = 0.0f;
    if (erp != 0.0f) {
        volatile float q = 1.0f / erp;
        volatile float iem1 = q - 1.0f;
        volatile float num = iem1 * iem1;
        volatile float d0 = 1.0f + iem1;
        volatile float d1 = d0 * 4.0f;
        volatile float d2 = d1 * damping;
        volatile float d3 = d2 * damping;
        cfm_coeff = num / d3;
    }
    volatile float den = 1.0f + cfm_coeff;
    return 1.0f / den;
}

static float spring_cfm_coeff(float freq, float damping, float dt) {
    volatile float erp = dt * spring_erp_inv_dt(freq, damping, dt);
    if (erp == 0.0f) return 0.0f;
    volatile float q = 1.0f / erp;
    volatile float iem1 = q - 1.0f;
    volatile float num = iem1 * iem1;
    volatile float d0 = 1.0f + iem1;
    volatile float d1 = d0 * 4.0f;
    volatile float d2 = d1 * damping;
    volatile float d3 = d2 * damping;
    return num / d3;
}

// SubParams of a solve group with `extra` additional substeps (init.rs:52-100: dt / (num_solver_iterations + extra))
static SubParams group_sub_params(const rp_integration_params &p, int extra) {
    SubParams s;
    s.extra = extra;
    s.num_substeps = p.num_solver_iterations + extra;
    volatile float dts = p.dt / (float)s.num_substeps;
    s.dt_sub = dts;
    s.inv_dt_sub = dts == 0.0f ? 0.0f : 1.0f / dts;
    s.dyn_cfm = spring_cfm_factor(p.contact_natural_frequency, p.contact_damping_ratio, dts);
    s.static_cfm = spring_cfm_factor(p.static_contact_natural_frequency, p.static_contact_damping_ratio, dts);
    s.dyn_erp_inv_dt = spring_erp_inv_dt(p.contact_natural_frequency, p.contact_damping_ratio, dts);
    s.static_erp_inv_dt = spring_erp_inv_dt(p.static_contact_natural_frequency, p.static_contact_damping_ratio, dts);
    s.joint_erp_inv_dt = spring_erp_inv_dt(p.joint_natural_frequency, p.joint_damping_ratio, dts);
    s.joint_cfm_coeff = spring_cfm_coeff(p.joint_natural_frequency, p.joint_damping_ratio, dts);
    return s;
}
// The distinct additional_solver_iterations counts of the live bodies, descending, 0 last (substep_groups.rs: one solve group per
// distinct count).  Returns the number of groups (1 = no elevated body), or -1 when there are more than RP_MAX_GROUPS.
static int group_table(const rp_world *w, std::vector<int> &extras) {
    extras.clear();
    for (const HostBody &b : w->bodies) {
        if (b.removed || b.quarantined || b.d.additional_solver_iterations <= 0) continue;
        if (std::find(extras.begin(), extras.end(), b.d.additional_solver_iterations) == extras.end()) extras.push_back(b.d.additional_solver_iterations);
    }
    std::sort(extras.begin(), extras.end(), [](int a, int b) { return a > b; });
    extras.push_back(0);
    return (int)extras.size() > RP_MAX_GROUPS ? -1 : (int)extras.size();
}
// (re)builds the group table of a finalized world; the caller destroys the step graphs when n_groups changed
static int upload_group_table(rp_world *w) {
    std::vector<int> extras;
    int ng = group_table(w, extras);
    if (ng < 0) { w->err = "more than 15 distinct positive additional_solver_iterations values in one world"; return RP_ERR_CAPACITY; }
    std::vector<SubParams> subs(RP_MAX_GROUPS, group_sub_params(w->params, 0));
    std::vector<int> ex(RP_MAX_GROUPS, 0);
    for (int g = 0; g < ng; ++g) { subs[g] = group_sub_params(w->params, extras[g]); ex[g] = extras[g]; }
    HIPCHK(w, hipMemcpy(w->dw.grp_sub, subs.data(), RP_MAX_GROUPS * sizeof(SubParams), hipMemcpyHostToDevice));
    HIPCHK(w, hipMemcpy(w->dw.grp_extra, ex.data(), RP_MAX_GROUPS * sizeof(int), hipMemcpyHostToDevice));
    w->dw.n_groups = ng;
    return RP_OK;
}
static void fill_sim_params(rp_world *w, SimParams &sp, float cell) {
    const rp_integration_params &p = w->params;
    sp.p = p;
    for (int k = 0; k < 3; ++k) sp.gravity[k] = w->gravity[k];
    sp.num_substeps = p.num_solver_iterations;
    volatile float dts = p.dt / (float)sp.num_substeps;
    sp.dt_sub = dts;
    sp.inv_dt_sub = dts == 0.0f ? 0.0f : 1.0f / dts;
    sp.dyn_cfm = spring_cfm_factor(p.contact_natural_frequency, p.contact_damping_ratio, dts);
    sp.static_cfm = spring_cfm_factor(p.static_contact_natural_frequency, p.static_contact_damping_ratio, dts);
    sp.dyn_erp_inv_dt = spring_erp_inv_dt(p.contact_natural_frequency, p.contact_damping_ratio, dts);
    sp.static_erp_inv_dt = spring_erp_inv_dt(p.static_contact_natural_frequency, p.static_contact_damping_ratio, dts);
    sp.joint_erp_inv_dt = spring_erp_inv_dt(p.joint_natural_frequency, p.joint_damping_ratio, dts);
    sp.joint_cfm_coeff = spring_cfm_coeff(p.joint_natural_frequency, p.joint_damping_ratio, dts);
    sp.prediction = p.normalized_prediction_distance * p.length_unit;
    sp.recycle_distance = p.contact_recycling ? p.normalized_contact_recycle_distance * p.length_unit : 0.0f;
    sp.max_corrective_velocity = p.normalized_max_corrective_velocity * p.length_unit;
    sp.max_lin = p.normalized_max_linear_velocity * p.length_unit;
    volatile float inv_dt = p.dt == 0.0f ? 0.0f : 1.0f / p.dt;
    sp.max_ang = 0.78539816339744830962f * inv_dt; // MAX_ROTATION * inv_dt, worker.rs:29,575
    sp.bp_skin = 4.0e-2f * p.length_unit;           // CHANGE_DETECTION_FACTOR, broad_phase_bvh/mod.rs:175
    sp.cell_size = cell;
    sp.inv_cell_size = 1.0f / cell;
}

// IntegrationParameters as the reference types them (integration_parameters.rs:181-304): num_solver_iterations is a NonZeroUsize,
// the other counts are usize, the lengths and frequencies are positive reals.  One check shared by create and set, so that a
// zero-iteration substep loop (dt_sub = inf, a fused step whose arrival never happens) cannot reach the device.
static const char *params_problem(const rp_integration_params &p) {
    auto real = [](float x) { return std::isfinite(x); };
    if (!real(p.dt) || p.dt < 0.0f) return "dt must be finite and >= 0";
    if (p.num_solver_iterations < 1) return "num_solver_iterations must be >= 1";
    if (p.num_internal_pgs_iterations < 0 || p.num_internal_stabilization_iterations < 0) return "the internal iteration counts must be >= 0";
    if (p.num_solver_iterations > 4096 || p.num_internal_pgs_iterations > 4096 || p.num_internal_stabilization_iterations > 4096) return "iteration counts above 4096";
    if (!real(p.length_unit) || !(p.length_unit > 0.0f)) return "length_unit must be finite and > 0";
    if (p.friction_model != RP_FRICTION_SIMPLIFIED && p.friction_model != RP_FRICTION_COULOMB) return "unknown friction_model";
    if (!real(p.contact_natural_frequency) || !real(p.contact_damping_ratio) || !real(p.static_contact_natural_frequency) || !real(p.static_contact_damping_ratio) ||
        !real(p.joint_natural_frequency) || !real(p.joint_damping_ratio) || p.contact_natural_frequency < 0.0f || p.static_contact_natural_frequency < 0.0f || p.joint_natural_frequency < 0.0f)
        return "spring frequencies / damping ratios must be finite (frequencies >= 0)";
    if (!real(p.warmstart_coefficient) || !real(p.normalized_allowed_linear_error) || !real(p.normalized_max_corrective_velocity) || !real(p.normalized_prediction_distance) ||
        !real(p.normalized_max_linear_velocity) || !real(p.normalized_contact_recycle_distance))
        return "non-finite parameter";
    if (p.normalized_prediction_distance < 0.0f || p.normalized_allowed_linear_error < 0.0f) return "negative distance parameter";
    return nullptr;
}
extern "C" int32_t rp_world_create(const rp_integration_params *params, const float gravity[3], int32_t device, rp_world **out) {
    if (!out) return RP_ERR_INVALID;
    *out = nullptr;
    if (params && params_problem(*params)) return RP_ERR_INVALID;
    if (gravity && !(std::isfinite(gravity[0]) && std::isfinite(gravity[1]) && std::isfinite(gravity[2]))) return RP_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RP_ERR_DEVICE; // no CPU fallback
    if (device < 0 || device >= ndev) return RP_ERR_DEVICE;
    std::lock_guard<std::mutex> guard(g_hip_unsafe_api); // device queries and stream creation next to another world's graph capture
    rp_world *w = new rp_world();
    w->device = device;
    if (params) w->params = *params; else rp_default_params(&w->params);
    for (int k = 0; k < 3; ++k) w->gravity[k] = gravity ? gravity[k] : 0.0f;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) { delete w; return RP_ERR_DEVICE; }
    const char *g = getenv("RP_NO_GRAPH");
    if (g && g[0] == '1') w->use_graph = false;
    g = getenv("RP_NO_FAST");
    if (g && g[0] == '1') w->use_fast = false;
    g = getenv("RP_NO_FUSED");
    if (g && g[0] == '1') w->use_fused = false;
    g = getenv("RP_NO_LEAN");
    if (g && g[0] == '1') w->use_lean = false;
    g = getenv("RP_NO_FLOW");
    if (g && g[0] == '1') w->use_flow = false;
    g = getenv("RP_FLOW");
    if (g && g[0] == '1') w->force_flow = true;
    if (w->use_flow) { w->flow_grid = rp_flow_grid(device); if (w->flow_grid <= 0) w->use_flow = false; }
    if (w->use_fused) { w->fused_grid = rp_fused_grid(device); if (w->fused_grid <= 0) w->use_fused = false; }
#ifdef RP_TESTING
    if (const char *ra = getenv("RP_TEST_REBASE_AT")) w->rebase_at = std::max(8, atoi(ra)); // (test hook: the stamps move back every few steps)
#endif
    { const char *nd = getenv("RP_NO_ISL_DENSE"); w->fused_grid_dense = (nd && nd[0] == '1') ? 0 : rp_fused_grid_dense(device); if (const char *fd = getenv("RP_ISL_DENSE")) { if (fd[0] == '1' && w->fused_grid_dense > 0) w->force_dense = true; if (fd[0] == '0') w->auto_dense = false; } }
    memset(&w->dw, 0, sizeof(w->dw));
    *out = w;
    return RP_OK;
}

static void destroy_graphs(rp_world *w) {
    for (int m = 0; m < 3; ++m) {
        hipGraphExec_t *ex[] = {&w->ge_whole[m], &w->ge_col[m], &w->ge_loop[m], &w->ge_fin[m]};
        hipGraph_t *gr[] = {&w->g_whole[m], &w->g_col[m], &w->g_loop[m], &w->g_fin[m]};
        for (auto e : ex) if (*e) { hipGraphExecDestroy(*e); *e = nullptr; }
        for (auto g : gr) if (*g) { hipGraphDestroy(*g); *g = nullptr; }
    }
    w->graph_stages = -1; w->graph_blocks = -1; w->graph_single = -1; w->graph_island_grid = -1; w->graph_dense = -1; w->graph_wide = -1; w->graph_joint_stages = -1; w->graph_no_global = -1; w->graph_fused = -1; w->graph_tile_grid = -1; w->graph_no_contacts = -1; w->graph_bare = -1;
    w->timed_ready[0] = w->timed_ready[1] = false;
}
static void free_device(rp_world *w) {
    destroy_graphs(w);
    for (const AllocRec &a : w->allocs) hipFree(a.ptr);
    w->allocs.clear();
    for (const AllocRec &a : w->old_allocs) hipFree(a.ptr);
    w->old_allocs.clear(); w->carry = false;
    if (w->pinned_flags) { hipHostFree(w->pinned_flags); w->pinned_flags = nullptr; }
    if (w->old_pinned) { hipHostFree(w->old_pinned); w->old_pinned = nullptr; }
    w->finalized = false;
}

extern "C" int32_t rp_world_destroy(rp_world *w) {
    if (!w) return RP_ERR_INVALID;
    hipSetDevice(w->device);
    if (w->stream) hipStreamSynchronize(w->stream);
    free_device(w);
    for (void *&b : w->cv_dev) if (b) { hipFree(b); b = nullptr; }
    for (void *&b : w->cm_dev) if (b) { hipFree(b); b = nullptr; }
    if (w->d_speed) { hipFree(w->d_speed); w->d_speed = nullptr; }
    for (auto &e : w->ev) if (e) hipEventDestroy(e);
    if (w->stream) hipStreamDestroy(w->stream);
    delete w;
    return RP_OK;
}
extern "C" const char *rp_last_error(const rp_world *w) { return w ? w->err.c_str() : "null world"; }
extern "C" int32_t rp_params_get(const rp_world *w, rp_integration_params *out) { if (!w || !out) return RP_ERR_INVALID; *out = w->params; return RP_OK; }
extern "C" int32_t rp_params_set(rp_world *w, const rp_integration_params *in) {
    if (!w || !in) return RP_ERR_INVALID;
    if (w->finalized) { int r = settle(w); if (r != RP_OK) return r; }
    if (const char *why = params_problem(*in)) { w->err = std::string("rp_params_set: ") + why; return RP_ERR_INVALID; }
    if (w->finalized && in->friction_model != w->params.friction_model) {
        // the constraint planes are sized per friction model: rebuild the device world from the current state
        int r = rebuild_begin(w);
        if (r != RP_OK) return r;
    }
    w->params = *in;
    if (w->finalized) {
        fill_sim_params(w, w->dw.prm, w->dw.prm.cell_size);
        int r = upload_group_table(w); if (r != RP_OK) return r;
        // the dataflow solver's tickets are laid out per (substep, iteration): rebuilt, and re-checked against FLOW_TICKET_BITS, by the next step
        int one = 1; HIPCHK(w, hipMemcpy(w->dw.flags + FL_FLOW_DIRTY, &one, sizeof(int), hipMemcpyHostToDevice));
        destroy_graphs(w);
    }
    return RP_OK;
}
extern "C" int32_t rp_num_bodies(const rp_world *w) { return w ? (int32_t)w->bodies.size() : 0; }

// parry Shape::mass_properties for cuboid / ball / capsule; frame = the shape's principal inertia local frame
// (identity except for capsules along X / Z: MassProperties::from_capsule rotates Y onto the segment direction)
// the inner shape of a round one (parry RoundShape<S>::inner_shape) and its border radius
static int core_shape(int shape) { return (shape >= RP_SHAPE_ROUND_CUBOID && shape <= RP_SHAPE_ROUND_CONVEX_POLYHEDRON) ? (shape == RP_SHAPE_ROUND_CUBOID ? RP_SHAPE_CUBOID : shape - RP_SHAPE_ROUND_CYLINDER + RP_SHAPE_CYLINDER) : shape; }
static float shape_border(const rp_collider_desc &c) { return (c.shape >= RP_SHAPE_ROUND_CUBOID && c.shape <= RP_SHAPE_ROUND_CONVEX_POLYHEDRON) ? c.border_radius : 0.0f; }
static bool shape_composite(int shape) { return shape == RP_SHAPE_COMPOUND || shape == RP_SHAPE_TRIMESH; }
// glam Quat::mul_vec3 (scalar path), the form the device and the checker use
static void h_qrot(const float q[4], const float v[3], float out[3]) {
    const float bx = q[0], by = q[1], bz = q[2], w = q[3];
    const float b2 = bx * bx + by * by + bz * bz, vb = v[0] * bx + v[1] * by + v[2] * bz;
    const float cx = by * v[2] - bz * v[1], cy = bz * v[0] - bx * v[2], cz = bx * v[1] - by * v[0];
    const float k0 = w * w - b2, k1 = vb * 2.0f, k2 = w * 2.0f;
    out[0] = v[0] * k0 + bx * k1 + cx * k2; out[1] = v[1] * k0 + by * k1 + cy * k2; out[2] = v[2] * k0 + bz * k1 + cz * k2;
}
static float shape_bounding_radius_core(const rp_world *w, int ci);
static float shape_bounding_radius(const rp_world *w, int ci) { // RoundShape: the inner sphere + the border
    const float r = shape_bounding_radius_core(w, ci), b = shape_border(w->colliders[ci]);
    return b > 0.0f ? r + b : r;
}
static float shape_bounding_radius_core(const rp_world *w, int ci) { // Shape::compute_local_bounding_sphere (about the collider origin)
    rp_collider_desc c = w->colliders[ci]; c.shape = core_shape(c.shape);
    if (c.shape == RP_SHAPE_CONVEX_POLYHEDRON) return w->polys[w->collider_poly[ci]].origin_radius; // (the CCD pre-filter and the grid's cell size; max_extent uses the point cloud's own sphere)
    if (c.shape == RP_SHAPE_CUBOID || shape_composite(c.shape)) // (a composite: the sphere about its local box)
        return std::sqrt(c.half_extents[0] * c.half_extents[0] + c.half_extents[1] * c.half_extents[1] + c.half_extents[2] * c.half_extents[2]);
    if (false) return std::sqrt(c.half_extents[0] * c.half_extents[0] + c.half_extents[1] * c.half_extents[1] + c.half_extents[2] * c.half_extents[2]);
    if (c.shape == RP_SHAPE_CAPSULE) return c.half_extents[0] + c.half_extents[1];
    if (c.shape == RP_SHAPE_HALFSPACE) return 3.402823466e+38f;
    if (c.shape == RP_SHAPE_CYLINDER || c.shape == RP_SHAPE_CONE) { volatile float rr = c.half_extents[1] * c.half_extents[1], hh2 = c.half_extents[0] * c.half_extents[0]; return std::sqrt(rr + hh2); }
    return c.half_extents[0];
}
static void hmp_diagonalise(float a[3][3], float pi[3], float frame[4]);
static void shape_mass_props_desc(const rp_world *w, rp_collider_desc c, int poly_id, float density, float &mass, float pi[3], float frame[4], float com[3]);
static void shape_mass_props(const rp_world *w, int ci, float density, float &mass, float pi[3], float frame[4], float com[3]) {
    shape_mass_props_desc(w, w->colliders[ci], w->collider_poly[ci], density, mass, pi, frame, com);
}
// (by descriptor: a collider, or a part of a compound shape)
static void shape_mass_props_desc(const rp_world *w, rp_collider_desc c, int poly_id, float density, float &mass, float pi[3], float frame[4], float com[3]) {
    c.shape = core_shape(c.shape); // (RoundShape::mass_properties = the inner shape's)
    frame[0] = 0.0f; frame[1] = 0.0f; frame[2] = 0.0f; frame[3] = 1.0f;
    com[0] = com[1] = com[2] = 0.0f;
    if (c.shape == RP_SHAPE_CONVEX_POLYHEDRON) { // MassProperties::from_convex_polyhedron -> with_inertia_matrix(com, volume * density, tensor * density)
        const HostPolyhedron &P = w->polys[(size_t)poly_id];
        float a[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = P.inertia[i][j] * density;
        hmp_diagonalise(a, pi, frame);
        mass = P.volume * density;
        com[0] = P.com[0] - P.centre[0]; com[1] = P.com[1] - P.centre[1]; com[2] = P.com[2] - P.centre[2]; // (in the recentred collider frame)
        return;
    }
    if (c.shape == RP_SHAPE_CYLINDER) { // MassProperties::from_cylinder (cylinder_y_volume_unit_inertia)
        float hh = c.half_extents[0], r = c.half_extents[1];
        volatile float vol = hh * r * r * 3.14159265358979323846f * 2.0f;
        volatile float sq_radius = r * r, sq_height = hh * hh * 4.0f;
        volatile float off_principal = (sq_radius * 3.0f + sq_height) / 12.0f;
        volatile float m = vol * density;
        volatile float iy = sq_radius / 2.0f * m, ixz = off_principal * m;
        mass = m; pi[0] = ixz; pi[1] = iy; pi[2] = ixz;
    } else if (c.shape == RP_SHAPE_CONE) { // MassProperties::from_cone (cone_y_volume_unit_inertia): the centre of mass a quarter of the height above the base
        float hh = c.half_extents[0], r = c.half_extents[1];
        volatile float vol = r * r * 3.14159265358979323846f * hh * 2.0f / 3.0f;
        volatile float sq_radius = r * r, sq_height = hh * hh * 4.0f;
        volatile float t0 = sq_radius * 3.0f / 20.0f, t1 = sq_height * 3.0f / 80.0f;
        volatile float off_principal = t0 + t1;
        volatile float principal = sq_radius * 3.0f / 10.0f;
        volatile float m = vol * density;
        volatile float iy = principal * m, ixz = off_principal * m;
        mass = m; pi[0] = ixz; pi[1] = iy; pi[2] = ixz;
        com[1] = -hh / 2.0f;
    } else if (c.shape == RP_SHAPE_CUBOID) {
        const float *he = c.half_extents;
        volatile float vol = he[0] * he[1] * he[2] * 8.0f;
        volatile float m = vol * density;
        volatile float ix = (he[1] * he[1] + he[2] * he[2]) / 3.0f;
        volatile float iy = (he[0] * he[0] + he[2] * he[2]) / 3.0f;
        volatile float iz = (he[0] * he[0] + he[1] * he[1]) / 3.0f;
        mass = m; pi[0] = ix * m; pi[1] = iy * m; pi[2] = iz * m;
    } else if (c.shape == RP_SHAPE_CAPSULE) { // MassProperties::from_capsule: a Y cylinder + a ball split in two caps
        float hh = c.half_extents[0], r = c.half_extents[1];
        volatile float cyl_vol = hh * r * r * 3.14159265358979323846f * 2.0f;
        volatile float sq_radius = r * r, sq_height = hh * hh * 4.0f;
        volatile float off_principal = (sq_radius * 3.0f + sq_height) / 12.0f;
        volatile float ball_vol = 3.14159265358979323846f * r * r * r * 4.0f / 3.0f;
        volatile float ball_i = r * r * 0.4f;
        volatile float cap_mass = (cyl_vol + ball_vol) * density;
        volatile float ix = (off_principal * cyl_vol + ball_i * ball_vol) * density;
        volatile float iy = (sq_radius / 2.0f * cyl_vol + ball_i * ball_vol) * density;
        volatile float h = hh * 2.0f;
        volatile float extra = (h * h * 0.25f + h * r * 3.0f / 8.0f) * ball_vol * density;
        mass = cap_mass; pi[0] = ix + extra; pi[1] = iy; pi[2] = ix + extra;
        int axis = (int)c.half_extents[2];
        if (axis == 0) { frame[2] = -0.70710678118654752f; frame[3] = 0.70710678118654752f; }
        else if (axis == 2) { frame[0] = 0.70710678118654752f; frame[3] = 0.70710678118654752f; }
    } else if (c.shape == RP_SHAPE_HALFSPACE) { // MassProperties::zero(): an unbounded shape weighs nothing
        mass = 0.0f; pi[0] = pi[1] = pi[2] = 0.0f;
    } else {
        float r = c.half_extents[0];
        volatile float vol = 3.14159265358979323846f * r * r * r * 4.0f / 3.0f;
        volatile float m = vol * density;
        volatile float i = r * r * 0.4f;
        mass = m; pi[0] = pi[1] = pi[2] = i * m;
    }
}
static float h_inv(float x) { return (x > -1.0e-20f && x < 1.0e-20f) ? 0.0f : 1.0f / x; }

/* ---- parry MassProperties algebra (not in /root/reference; restated from its public definition) ----------------
 * A MassProperties value = (mass, local_com, principal inertia, principal frame).  `transform_by(pos)` moves the
 * centre and rotates the frame; `a + b` = total mass, mass-weighted centre, sum of the two inertia tensors shifted
 * to the common centre (parallel-axis theorem), re-diagonalised.  parry diagonalises with nalgebra's
 * symmetric_eigen (Householder + QR); a cyclic Jacobi iteration is used here instead — same eigen-system, rounding
 * differs (unpinned like every other parry quantity).  Arithmetic is plain f32, no contraction. */
typedef struct { float mass; float com[3]; float pi[3]; float frame[4]; } hmp_mp;   /* frame: quaternion x,y,z,w */

static void hmp_quat_to_rot(const float q[4], float r[3][3]) {
    float x2 = q[0] + q[0], y2 = q[1] + q[1], z2 = q[2] + q[2];
    float xx = q[0] * x2, xy = q[0] * y2, xz = q[0] * z2;
    float yy = q[1] * y2, yz = q[1] * z2, zz = q[2] * z2;
    float wx = q[3] * x2, wy = q[3] * y2, wz = q[3] * z2;
    r[0][0] = 1.0f - (yy + zz); r[0][1] = xy - wz; r[0][2] = xz + wy;
    r[1][0] = xy + wz; r[1][1] = 1.0f - (xx + zz); r[1][2] = yz - wx;
    r[2][0] = xz - wy; r[2][1] = yz + wx; r[2][2] = 1.0f - (xx + yy);
}
/* reconstruct_inertia_matrix: R diag(pi) R^T */
static void hmp_inertia_matrix(const hmp_mp *m, float out[3][3]) {
    float r[3][3]; hmp_quat_to_rot(m->frame, r);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            out[i][j] = r[i][0] * m->pi[0] * r[j][0] + r[i][1] * m->pi[1] * r[j][1] + r[i][2] * m->pi[2] * r[j][2];
}
/* construct_shifted_inertia_matrix: I + (|s|^2 Id - s s^T) * mass */
static void hmp_shifted_inertia(const hmp_mp *m, const float s[3], float out[3][3]) {
    hmp_inertia_matrix(m, out);
    float d = s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            out[i][j] = out[i][j] + ((i == j ? d : 0.0f) - s[i] * s[j]) * m->mass;
}
/* rotation matrix (columns = axes) -> unit quaternion (Shepperd's method) */
static void hmp_rot_to_quat(float v[3][3], float q[4]) {
    float tr = v[0][0] + v[1][1] + v[2][2];
    if (tr > 0.0f) {
        float s = sqrtf(tr + 1.0f) * 2.0f;
        q[3] = 0.25f * s; q[0] = (v[2][1] - v[1][2]) / s; q[1] = (v[0][2] - v[2][0]) / s; q[2] = (v[1][0] - v[0][1]) / s;
    } else if (v[0][0] > v[1][1] && v[0][0] > v[2][2]) {
        float s = sqrtf(1.0f + v[0][0] - v[1][1] - v[2][2]) * 2.0f;
        q[3] = (v[2][1] - v[1][2]) / s; q[0] = 0.25f * s; q[1] = (v[0][1] + v[1][0]) / s; q[2] = (v[0][2] + v[2][0]) / s;
    } else if (v[1][1] > v[2][2]) {
        float s = sqrtf(1.0f + v[1][1] - v[0][0] - v[2][2]) * 2.0f;
        q[3] = (v[0][2] - v[2][0]) / s; q[0] = (v[0][1] + v[1][0]) / s; q[1] = 0.25f * s; q[2] = (v[1][2] + v[2][1]) / s;
    } else {
        float s = sqrtf(1.0f + v[2][2] - v[0][0] - v[1][1]) * 2.0f;
        q[3] = (v[1][0] - v[0][1]) / s; q[0] = (v[0][2] + v[2][0]) / s; q[1] = (v[1][2] + v[2][1]) / s; q[2] = 0.25f * s;
    }
    float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    float inv = 1.0f / n;
    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
/* with_inertia_matrix: principal inertia + frame of a symmetric 3x3 tensor (cyclic Jacobi, 12 sweeps) */
static void hmp_diagonalise(float a[3][3], float pi[3], float frame[4]) {
    float v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        float off = fabsf(a[0][1]) + fabsf(a[0][2]) + fabsf(a[1][2]);
        if (off == 0.0f) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0f) continue;
                float theta = (a[q][q] - a[p][p]) / (2.0f * a[p][q]);
                float t = (theta >= 0.0f ? 1.0f : -1.0f) / (fabsf(theta) + sqrtf(theta * theta + 1.0f));
                float c = 1.0f / sqrtf(t * t + 1.0f), s = t * c;
                for (int k = 0; k < 3; ++k) { float akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 3; ++k) { float apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 3; ++k) { float vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq; }
            }
    }
    /* a proper rotation: flip the last axis when the determinant is negative */
    float det = v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1]) - v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0]) +
                v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
    if (det < 0.0f) { v[0][2] = -v[0][2]; v[1][2] = -v[1][2]; v[2][2] = -v[2][2]; }
    for (int i = 0; i < 3; ++i) pi[i] = a[i][i] > 0.0f ? a[i][i] : 0.0f;
    hmp_rot_to_quat(v, frame);
}
/* MassProperties::transform_by(pose): centre moved, frame rotated */
static void hmp_mp_transform(hmp_mp *m, const float t[3], const float q[4]) {
    /* rotate com by q (glam Quat::mul_vec3), then translate */
    float bx = q[0], by = q[1], bz = q[2], w = q[3];
    float b2 = bx * bx + by * by + bz * bz, vb = m->com[0] * bx + m->com[1] * by + m->com[2] * bz;
    float cx = by * m->com[2] - bz * m->com[1], cy = bz * m->com[0] - bx * m->com[2], cz = bx * m->com[1] - by * m->com[0];
    float k0 = w * w - b2, k1 = vb * 2.0f, k2 = w * 2.0f;
    float rx = m->com[0] * k0 + bx * k1 + cx * k2, ry = m->com[1] * k0 + by * k1 + cy * k2, rz = m->com[2] * k0 + bz * k1 + cz * k2;
    m->com[0] = rx + t[0]; m->com[1] = ry + t[1]; m->com[2] = rz + t[2];
    /* frame = q * frame */
    float a[4] = {q[0], q[1], q[2], q[3]}, f[4] = {m->frame[0], m->frame[1], m->frame[2], m->frame[3]};
    m->frame[0] = a[3] * f[0] + a[0] * f[3] + a[1] * f[2] - a[2] * f[1];
    m->frame[1] = a[3] * f[1] - a[0] * f[2] + a[1] * f[3] + a[2] * f[0];
    m->frame[2] = a[3] * f[2] + a[0] * f[1] - a[1] * f[0] + a[2] * f[3];
    m->frame[3] = a[3] * f[3] - a[0] * f[0] - a[1] * f[1] - a[2] * f[2];
}
/* MassProperties + MassProperties */
static void hmp_mp_add(hmp_mp *acc, const hmp_mp *o) {
    if (acc->mass == 0.0f) { *acc = *o; return; }
    if (o->mass == 0.0f) return;
    float m1 = acc->mass, m2 = o->mass, total = m1 + m2, inv = 1.0f / total;
    float com[3], s1[3], s2[3];
    for (int k = 0; k < 3; ++k) com[k] = (acc->com[k] * m1 + o->com[k] * m2) * inv;
    for (int k = 0; k < 3; ++k) { s1[k] = com[k] - acc->com[k]; s2[k] = com[k] - o->com[k]; }
    float i1[3][3], i2[3][3], sum[3][3];
    hmp_shifted_inertia(acc, s1, i1); hmp_shifted_inertia(o, s2, i2);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) sum[i][j] = i1[i][j] + i2[i][j];
    /* symmetrise exactly (the shifted tensors are symmetric up to rounding) */
    sum[1][0] = sum[0][1]; sum[2][0] = sum[0][2]; sum[2][1] = sum[1][2];
    acc->mass = total; acc->com[0] = com[0]; acc->com[1] = com[1]; acc->com[2] = com[2];
    hmp_diagonalise(sum, acc->pi, acc->frame);
}

// sum of the attached colliders' mass properties at `density_override` (< 0: each collider's own density)
static void sum_collider_mass_props(const rp_world *w, int body, float density_override, hmp_mp *acc) {
    memset(acc, 0, sizeof(*acc)); acc->frame[3] = 1.0f;
    for (int i : w->bodies[body].cols) { // (not a walk over every collider of the world: a million in b3d_large_world, per inserted body)
        if (w->collider_removed[i]) continue;
        const rp_collider_desc &c = w->colliders[i];
        hmp_mp m; memset(&m, 0, sizeof(m)); m.frame[3] = 1.0f;
        if (shape_composite(c.shape)) { // MassProperties::from_compound: the sum of the parts' (a triangle mesh on a fixed body weighs nothing)
            if (c.shape == RP_SHAPE_COMPOUND) {
                const rp_world::HostComposite &C = w->comps[(size_t)w->collider_comp[i]];
                for (size_t k = 0; k < C.parts.size(); ++k) {
                    const rp_collider_desc &pd = C.parts[k];
                    hmp_mp pm; memset(&pm, 0, sizeof(pm)); pm.frame[3] = 1.0f;
                    shape_mass_props_desc(w, pd, C.part_poly[k], density_override < 0.0f ? c.density : density_override, pm.mass, pm.pi, pm.frame, pm.com);
                    hmp_mp_transform(&pm, pd.translation, pd.rotation);
                    if (k == 0) m = pm; else hmp_mp_add(&m, &pm);
                }
            }
        } else
        shape_mass_props(w, i, density_override < 0.0f ? c.density : density_override, m.mass, m.pi, m.frame, m.com);
        float qn = std::sqrt(c.rotation[0] * c.rotation[0] + c.rotation[1] * c.rotation[1] + c.rotation[2] * c.rotation[2] + c.rotation[3] * c.rotation[3]);
        float qi = qn > 0.0f ? 1.0f / qn : 1.0f;
        float q[4] = {c.rotation[0] * qi, c.rotation[1] * qi, c.rotation[2] * qi, qn > 0.0f ? c.rotation[3] * qi : 1.0f};
        hmp_mp_transform(&m, c.translation, q);
        hmp_mp_add(acc, &m);
    }
}
// RigidBodyMassProps::recompute_mass_properties_from_colliders — rigid_body_components.rs:421-489: the attached colliders'
// MassProperties (transformed by pos_wrt_parent) are summed in attachment order, then the additional mass.
static void recompute_mass(rp_world *w, int body) {
    HostBody &b = w->bodies[body];
    hmp_mp acc; sum_collider_mass_props(w, body, -1.0f, &acc);
    float add = b.d.additional_mass;
    if (add != 0.0f) {
        if (acc.mass > 0.0f) { // MassProperties::set_mass(prev + add, adjust_angular_inertia = true)
            float nm = acc.mass + add;
            float k = nm / acc.mass;
            acc.pi[0] = acc.pi[0] * k; acc.pi[1] = acc.pi[1] * k; acc.pi[2] = acc.pi[2] * k; acc.mass = nm;
        } else {
            hmp_mp unit; sum_collider_mass_props(w, body, 1.0f, &unit);
            if (unit.mass > 0.0f) {
                float k = add / unit.mass;
                unit.pi[0] = unit.pi[0] * k; unit.pi[1] = unit.pi[1] * k; unit.pi[2] = unit.pi[2] * k; unit.mass = add;
                acc = unit;
            } else acc.mass = add;
        }
    }
    b.inv_mass = h_inv(acc.mass);
    for (int q = 0; q < 3; ++q) { b.inv_pi[q] = h_inv(acc.pi[q]); b.lcom[q] = acc.com[q]; }
    for (int q = 0; q < 4; ++q) b.pframe[q] = acc.frame[q];
    // recompute_max_extent (rigid_body_components.rs:491-515): bounding spheres of the attached shapes about the local CoM
    b.max_extent = 0.0f;
    for (int i : b.cols) {
        if (w->collider_removed[i]) continue;
        const rp_collider_desc &c = w->colliders[i];
        float radius = shape_bounding_radius(w, i);
        float ctr[3] = {c.translation[0], c.translation[1], c.translation[2]};
        if (core_shape(c.shape) == RP_SHAPE_CONVEX_POLYHEDRON) { // point_cloud_bounding_sphere: centred on the mean of the points
            const HostPolyhedron &P = w->polys[w->collider_poly[i]];
            const float off[3] = {P.sphere_centre[0] - P.centre[0], P.sphere_centre[1] - P.centre[1], P.sphere_centre[2] - P.centre[2]};
            float qn = std::sqrt(c.rotation[0] * c.rotation[0] + c.rotation[1] * c.rotation[1] + c.rotation[2] * c.rotation[2] + c.rotation[3] * c.rotation[3]);
            float qi = qn > 0.0f ? 1.0f / qn : 1.0f;
            const float q[4] = {c.rotation[0] * qi, c.rotation[1] * qi, c.rotation[2] * qi, qn > 0.0f ? c.rotation[3] * qi : 1.0f};
            float r[3]; h_qrot(q, off, r);
            ctr[0] = r[0] + c.translation[0]; ctr[1] = r[1] + c.translation[1]; ctr[2] = r[2] + c.translation[2];
            radius = shape_border(c) > 0.0f ? P.sphere_radius + shape_border(c) : P.sphere_radius;
        }
        float dx = ctr[0] - b.lcom[0], dy = ctr[1] - b.lcom[1], dz = ctr[2] - b.lcom[2];
        float extent = std::sqrt(dx * dx + dy * dy + dz * dz) + radius;
        if (extent > b.max_extent) b.max_extent = extent;
    }
    // RigidBodyCcd::ccd_thickness (rigid_body_components.rs:1227): min over the attached shapes of Shape::ccd_thickness
    // (ball: radius, cuboid: smallest half extent, capsule: radius)
    b.ccd_thickness = 3.402823466e+38f;
    for (int i : b.cols) {
        if (w->collider_removed[i]) continue;
        const rp_collider_desc &c = w->colliders[i];
        if (c.shape == RP_SHAPE_HALFSPACE) continue; // Shape::ccd_thickness of a half-space is f32::MAX
        const int ck = core_shape(c.shape);
        float th = ck == RP_SHAPE_BALL ? c.half_extents[0] : ck == RP_SHAPE_CAPSULE ? c.half_extents[1] : (ck == RP_SHAPE_CYLINDER || ck == RP_SHAPE_CONE) ? std::min(c.half_extents[0], c.half_extents[1]) : std::min(c.half_extents[0], std::min(c.half_extents[1], c.half_extents[2]));
        if (shape_border(c) > 0.0f) th = th + shape_border(c); // RoundShape::ccd_thickness = inner + border
        b.ccd_thickness = std::min(b.ccd_thickness, th);
    }
}
// dynamic bodies with several colliders, or with a collider away from the body origin: the fused fast step validates ONE
// collider per body (b_collider), so such worlds keep to the fast graph / full graph
static bool world_has_compound_bodies(const rp_world *w) {
    for (size_t i = 0; i < w->colliders.size(); ++i) {
        int p = w->collider_parent[i];
        if (p < 0 || w->collider_removed[i] || w->bodies[p].d.body_type != RP_BODY_DYNAMIC) continue;
        const float *t = w->colliders[i].translation, *r = w->colliders[i].rotation;
        bool at_origin = t[0] == 0.0f && t[1] == 0.0f && t[2] == 0.0f && r[0] == 0.0f && r[1] == 0.0f && r[2] == 0.0f;
        if (!at_origin || w->bodies[p].ncolliders > 1) return true;
    }
    return false;
}

// Bring the host mirrors up to date with the device (poses, velocities) before the device world is
// rebuilt from them: inserting into a world that has been stepped continues from the current state
// (contact warm-start data is not carried over a rebuild).
static int download_rows(rp_world *w);
static int download_state(rp_world *w) {
    if (!w->finalized) return RP_OK;
    { int r = settle(w); if (r != RP_OK) return r; }
    return download_rows(w);
}
// (the copies alone: the caller knows the stream to be idle and the device world to hold every requested step)
static int download_rows(rp_world *w) {
    int nb = w->dw.n_bodies;
    std::vector<float4> pos(nb), rot(nb), lv(nb), av(nb);
    HIPCHK(w, hipMemcpy(pos.data(), w->dw.b_pos, nb * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(w, hipMemcpy(rot.data(), w->dw.b_rot, nb * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(w, hipMemcpy(lv.data(), w->dw.b_linvel, nb * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(w, hipMemcpy(av.data(), w->dw.b_angvel, nb * sizeof(float4), hipMemcpyDeviceToHost));
    std::vector<float4> slp(nb), spt(nb), spr(nb); std::vector<int> bfl(nb), slab(nb);
    HIPCHK(w, hipMemcpy(slp.data(), w->dw.b_sleep, nb * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(w, hipMemcpy(spt.data(), w->dw.b_sprev_t, nb * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(w, hipMemcpy(spr.data(), w->dw.b_sprev_r, nb * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(w, hipMemcpy(bfl.data(), w->dw.b_flags, nb * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(w, hipMemcpy(slab.data(), w->dw.b_slabel, nb * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<float4> npos(nb), nrot(nb);
    HIPCHK(w, hipMemcpy(npos.data(), w->dw.b_next_pos, nb * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(w, hipMemcpy(nrot.data(), w->dw.b_next_rot, nb * sizeof(float4), hipMemcpyDeviceToHost));
    if (w->dw.sleep_enabled) { // persistent islands: ids per body + the island table
        std::vector<int> isl(nb);
        HIPCHK(w, hipMemcpy(isl.data(), w->dw.b_isl, nb * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < nb; ++i) w->bodies[i].isl = isl[i];
        auto &ps = w->pi_saved;
        int fl[FL_COUNT];
        HIPCHK(w, hipMemcpy(fl, w->dw.flags, sizeof(fl), hipMemcpyDeviceToHost));
        ps.next = fl[FL_PI_NEXT]; ps.nfree = fl[FL_PI_NFREE]; ps.pending = fl[FL_PI_PENDING];
        const int n = std::max(ps.next, 1);
        ps.used.resize(n); ps.nb.resize(n); ps.dirty.resize(n); ps.denied.resize(n); ps.sleeping.resize(n); ps.freel.resize(std::max(ps.nfree, 1)); ps.stats.resize(16);
        HIPCHK(w, hipMemcpy(ps.used.data(), w->dw.pi_used, n * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(ps.nb.data(), w->dw.pi_nb, n * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(ps.dirty.data(), w->dw.pi_dirty, n * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(ps.denied.data(), w->dw.pi_denied, n * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(ps.sleeping.data(), w->dw.pi_sleeping, n * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(ps.freel.data(), w->dw.pi_free, ps.freel.size() * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(ps.stats.data(), w->dw.pi_stats, 16 * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(ps.w64, w->dw.pi_w64, sizeof(ps.w64), hipMemcpyDeviceToHost));
        ps.valid = true;
    } else w->pi_saved.valid = false;
    for (int i = 0; i < nb; ++i) {
        HostBody &hb = w->bodies[i];
        hb.has_next = true; hb.next[0] = npos[i].x; hb.next[1] = npos[i].y; hb.next[2] = npos[i].z;
        hb.next[3] = nrot[i].x; hb.next[4] = nrot[i].y; hb.next[5] = nrot[i].z; hb.next[6] = nrot[i].w;
        hb.sleep_timer = slp[i].x; hb.sleeping = (bfl[i] & RP_BF_SLEEPING) ? 1 : 0; hb.slabel = slab[i];
        hb.sprev[0] = spt[i].x; hb.sprev[1] = spt[i].y; hb.sprev[2] = spt[i].z;
        hb.sprev[3] = spr[i].x; hb.sprev[4] = spr[i].y; hb.sprev[5] = spr[i].z; hb.sprev[6] = spr[i].w;
    }
    for (int i = 0; i < nb; ++i) {
        rp_body_desc &d = w->bodies[i].d;
        d.translation[0] = pos[i].x; d.translation[1] = pos[i].y; d.translation[2] = pos[i].z;
        d.rotation[0] = rot[i].x; d.rotation[1] = rot[i].y; d.rotation[2] = rot[i].z; d.rotation[3] = rot[i].w;
        d.linvel[0] = lv[i].x; d.linvel[1] = lv[i].y; d.linvel[2] = lv[i].z;
        d.angvel[0] = av[i].x; d.angvel[1] = av[i].y; d.angvel[2] = av[i].z;
    }
    return RP_OK;
}
// pose of a body descriptor / local frame of a joint descriptor (GenericJoint::local_frame1/2)
static Pose host_body_pose(const HostBody &b) {
    Pose p; const float *r = b.d.rotation;
    float qn = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    float qi = qn > 0.0f ? 1.0f / qn : 1.0f;
    p.r = q4(r[0] * qi, r[1] * qi, r[2] * qi, qn > 0.0f ? r[3] * qi : 1.0f);
    p.t = v3(b.d.translation[0], b.d.translation[1], b.d.translation[2]);
    return p;
}
static Pose joint_local_frame(const float *anchor, const float *basis) {
    Pose p; p.r = qnormalize(q4(basis[0], basis[1], basis[2], basis[3])); p.t = v3(anchor[0], anchor[1], anchor[2]); return p;
}
static int rebuild_begin(rp_world *w) { // called before the host mirrors grow
    if (!w->finalized) return RP_OK;
    HIPCHK(w, hipSetDevice(w->device));
    int r = download_state(w);
    if (r != RP_OK) return r;
    free_device(w);
    return RP_OK;
}

extern "C" int32_t rp_bodies_insert(rp_world *w, int32_t n, const rp_body_desc *descs, uint64_t *handles_out) {
    if (!w || n < 0 || (n > 0 && !descs)) return RP_ERR_INVALID;
    // RigidBodySet::insert into a live world: rows are appended in place while the device arrays have
    // room (every pair keeps its warm-start data); otherwise the device world is rebuilt from the
    // current body states
    // Arena::insert (arena.rs:260-290): removed slots first (LIFO), then fresh indices.  A batch that needs both is split: the reused
    // slots are rewritten in place, the rest follows the append path (in place, or through the growth carry-over)
    static const bool no_reuse = getenv("RP_NO_ARENA_REUSE") != nullptr; // (debug: append only, like rounds 1-3)
    const int n_reuse = no_reuse ? 0 : std::min<int>(n, (int)w->body_free.size());
    // the WHOLE batch is validated before anything changes (and before the split below: a batch is inserted entirely or not at all)
    if ((long long)w->bodies.size() + (n - n_reuse) >= 0xfffff) { w->err = "rp_bodies_insert: more than 2^20 - 1 bodies"; return RP_ERR_CAPACITY; }
    for (int i = 0; i < n; ++i) if (descs[i].body_type < RP_BODY_DYNAMIC || descs[i].body_type > RP_BODY_KINEMATIC_VELOCITY) { w->err = "rp_bodies_insert: unknown body_type"; return RP_ERR_INVALID; }
    for (int i = 0; i < n; ++i) if (descs[i].additional_solver_iterations < 0 || descs[i].additional_solver_iterations > 4096) { w->err = "rp_bodies_insert: additional_solver_iterations must be in [0, 4096]"; return RP_ERR_INVALID; }
    { // the distinct-count limit of the solve groups is checked on the prospective values, before host or device state changes
        std::vector<int> extras; group_table(w, extras);
        for (int i = 0; i < n; ++i) if (descs[i].additional_solver_iterations > 0 && std::find(extras.begin(), extras.end(), descs[i].additional_solver_iterations) == extras.end()) extras.push_back(descs[i].additional_solver_iterations);
        if ((int)extras.size() > RP_MAX_GROUPS) { w->err = "more than 15 distinct positive additional_solver_iterations values in one world"; return RP_ERR_CAPACITY; }
    }
    if (n_reuse > 0 && n_reuse < n) {
        int r = rp_bodies_insert(w, n_reuse, descs, handles_out);
        return r != RP_OK ? r : rp_bodies_insert(w, n - n_reuse, descs + n_reuse, handles_out ? handles_out + n_reuse : nullptr);
    }
    const bool reuse = n_reuse > 0;
    const bool in_place = w->finalized && (reuse || (int)w->bodies.size() + n <= w->cap_bodies);
    if (n > 0 && w->finalized) {
        HIPCHK(w, hipSetDevice(w->device));
        int r = in_place ? settle(w) : grow_begin(w);
        if (r != RP_OK) return r;
    }
    if (reuse && w->finalized) { int r = purge_dead_pairs(w); if (r != RP_OK) return r; } // no pair may still name the slots' previous occupants
    int first_slot = -1;
    for (int i = 0; i < n; ++i) {
        HostBody b; b.d = descs[i]; b.ncolliders = 0; b.removed = false; b.inv_mass = 0; b.inv_pi[0] = b.inv_pi[1] = b.inv_pi[2] = 0; b.lcom[0] = b.lcom[1] = b.lcom[2] = 0;
        int slot;
        if (reuse) {
            slot = w->body_free.back(); w->body_free.pop_back();
            // (the removed colliders of the slot's previous occupant no longer name it)
            for (size_t c = 0; c < w->colliders.size(); ++c) if (w->collider_parent[c] == slot && w->collider_removed[c]) w->collider_parent[c] = -1;
            b.slabel = slot;
            w->bodies[(size_t)slot] = b;
        } else {
            slot = (int)w->bodies.size();
            b.slabel = slot;
            w->bodies.push_back(b); w->body_gen.push_back(0);
        }
        w->body_gen[(size_t)slot] = w->body_arena_gen;
        if (first_slot < 0) first_slot = slot;
        recompute_mass(w, slot);
        if (handles_out) handles_out[i] = ((uint64_t)w->body_arena_gen << 32) | (uint64_t)(uint32_t)slot;
        if (in_place) {
            int r;
            if (reuse && (r = reset_row(w, DOM_BODY, slot)) != RP_OK) return r;
            if ((r = upload_body_row(w, slot)) != RP_OK) return r;
            if (reuse && w->dw.sleep_enabled) rp_launch_pi_ensure(w->dw, w->stream, slot, 1, 0);
        }
    }
    if (in_place && n > 0) {
        const int first_new = w->dw.n_bodies, was_sleep_enabled = w->dw.sleep_enabled;
        w->dw.n_bodies = (int)w->bodies.size();
        { int r = check_sleep_scope(w); if (r != RP_OK) return r; }
        w->dw.sleep_enabled = world_sleep_enabled(w) ? 1 : 0;
        // persistent islands: ensure_body for the new rows; a world that becomes sleep-enabled now bootstraps its islands
        if (w->dw.sleep_enabled && (!reuse || !was_sleep_enabled)) rp_launch_pi_ensure(w->dw, w->stream, was_sleep_enabled ? first_new : 0, was_sleep_enabled ? n : w->dw.n_bodies, was_sleep_enabled ? 0 : 1);
        w->dw.has_kinematic_pos = world_has_kinematic_pos(w) ? 1 : 0; refresh_ccd_facts(w);
        bool extras = false; for (int i = 0; i < n; ++i) extras |= descs[i].additional_solver_iterations > 0;
        if (extras) { int r = upload_group_table(w); if (r != RP_OK) return r; } // (a body without additional solver iterations leaves the group table alone)
        HIPCHK(w, hipStreamSynchronize(w->stream)); // replays of the graphs destroyed below may still be in flight
        destroy_graphs(w); // kernel arguments (DevWorld by value) hold the body count
        return after_topology_edit(w, true);
    }
    if (w->carry) return finalize(w); // the larger device world takes over the rows of the one it replaces
    return RP_OK;
}
// the cv_* tables of the device world (rp_world.h): every registered polyhedron, flattened.  Own allocations (not rows of a growth domain):
// replaced as a whole when a polyhedron is registered, referenced by pointer from DevWorld (the step graphs are captured again)
static float4 mk4(float x, float y, float z, float w_);
#define RP_CM_WS_F4_HOST (4 * 32 * 4 + 64 / 2) // = RP_CM_WS_F4 of rp_composite.h (clusters x points x 4 planes + the candidate list)
static int upload_polyhedra(rp_world *w) {
    std::vector<int4> hdr; std::vector<float4> pts, fn; std::vector<int2> fl, loop;
    for (const HostPolyhedron &P : w->polys) {
        int4 h; h.x = (int)pts.size(); h.y = P.nv(); h.z = (int)fn.size(); h.w = P.nf();
        hdr.push_back(h);
        const int loop0 = (int)loop.size();
        for (int i = 0; i < P.nv(); ++i) pts.push_back(mk4(P.pts[3 * i], P.pts[3 * i + 1], P.pts[3 * i + 2], P.origin_radius));
        for (int f = 0; f < P.nf(); ++f) { fn.push_back(mk4(P.fnormal[3 * f], P.fnormal[3 * f + 1], P.fnormal[3 * f + 2], 0.0f)); int2 r; r.x = loop0 + P.ffirst[f]; r.y = P.fcount[f]; fl.push_back(r); }
        for (size_t k = 0; k < P.loop_v.size(); ++k) { int2 r; r.x = P.loop_v[k]; r.y = P.loop_e[k]; loop.push_back(r); }
    }
    HIPCHK(w, hipStreamSynchronize(w->stream));
    for (void *&b : w->cv_dev) if (b) { HIPCHK(w, hipFree(b)); b = nullptr; }
    const void *src[5] = {hdr.data(), pts.data(), fn.data(), fl.data(), loop.data()};
    const size_t bytes[5] = {hdr.size() * sizeof(int4), pts.size() * sizeof(float4), fn.size() * sizeof(float4), fl.size() * sizeof(int2), loop.size() * sizeof(int2)};
    for (int k = 0; k < 5; ++k) {
        if (bytes[k] == 0) continue;
        HIPCHK(w, hipMalloc(&w->cv_dev[k], bytes[k]));
        HIPCHK(w, hipMemcpyAsync(w->cv_dev[k], src[k], bytes[k], hipMemcpyHostToDevice, w->stream));
    }
    HIPCHK(w, hipStreamSynchronize(w->stream));
    w->dw.cv_hdr = (int4 *)w->cv_dev[0]; w->dw.cv_pts = (float4 *)w->cv_dev[1]; w->dw.cv_fn = (float4 *)w->cv_dev[2]; w->dw.cv_fl = (int2 *)w->cv_dev[3]; w->dw.cv_loop = (int2 *)w->cv_dev[4];
    w->polys_uploaded = true;
    return RP_OK;
}
extern "C" int32_t rp_convex_polyhedron_create(rp_world *w, int32_t n_points, const float *points_xyz, int32_t n_triangles, const uint32_t *indices, int32_t *id_out) {
    if (!w || !points_xyz || !id_out || n_points < 4 || (indices && n_triangles < 4)) { if (w) w->err = "rp_convex_polyhedron_create: at least four points (and four triangles)"; return RP_ERR_INVALID; }
    std::vector<uint32_t> hull;
    if (!indices) { // SharedShape::convex_hull
        if (!rp_poly::convex_hull(n_points, points_xyz, hull)) { w->err = "rp_convex_polyhedron_create: the points have no volume (convex_hull returns None)"; return RP_ERR_INVALID; }
        indices = hull.data(); n_triangles = (int32_t)(hull.size() / 3);
    }
    HostPolyhedron P;
    if (!rp_poly::build(P, n_points, points_xyz, n_triangles, indices)) { w->err = "rp_convex_polyhedron_create: not a closed, outward-wound convex triangle mesh of 4..256 vertices"; return RP_ERR_INVALID; }
    w->polys.push_back(std::move(P));
    *id_out = (int32_t)w->polys.size() - 1;
    if (w->finalized) { // the tables are replaced: no launch may still read the old ones, the graphs hold the old pointers
        HIPCHK(w, hipSetDevice(w->device));
        int r = settle(w); if (r != RP_OK) return r;
        r = upload_polyhedra(w); if (r != RP_OK) return r;
        destroy_graphs(w);
    }
    return RP_OK;
}
extern "C" int32_t rp_convex_polyhedron_read(const rp_world *w, int32_t id, int32_t counts[4], float *points_xyz, float *face_normals, int32_t *face_first, int32_t *face_count,
                                             int32_t *loop_vertex, int32_t *loop_edge, float props[20]) {
    if (!w || !counts || id < 0 || id >= (int32_t)w->polys.size()) return RP_ERR_INVALID;
    const HostPolyhedron &P = w->polys[(size_t)id];
    counts[0] = P.nv(); counts[1] = P.nf(); counts[2] = (int32_t)P.loop_v.size(); counts[3] = P.ne;
    if (points_xyz) memcpy(points_xyz, P.pts.data(), P.pts.size() * sizeof(float));
    if (face_normals) memcpy(face_normals, P.fnormal.data(), P.fnormal.size() * sizeof(float));
    if (face_first) memcpy(face_first, P.ffirst.data(), P.ffirst.size() * sizeof(int));
    if (face_count) memcpy(face_count, P.fcount.data(), P.fcount.size() * sizeof(int));
    if (loop_vertex) memcpy(loop_vertex, P.loop_v.data(), P.loop_v.size() * sizeof(int));
    if (loop_edge) memcpy(loop_edge, P.loop_e.data(), P.loop_e.size() * sizeof(int));
    if (props) {
        float *o = props;
        o[0] = P.centre[0]; o[1] = P.centre[1]; o[2] = P.centre[2]; o[3] = P.half[0]; o[4] = P.half[1]; o[5] = P.half[2]; o[6] = P.origin_radius;
        o[7] = P.sphere_centre[0]; o[8] = P.sphere_centre[1]; o[9] = P.sphere_centre[2]; o[10] = P.sphere_radius;
        o[11] = P.volume; o[12] = P.com[0]; o[13] = P.com[1]; o[14] = P.com[2];
        o[15] = P.inertia[0][0]; o[16] = P.inertia[1][1]; o[17] = P.inertia[2][2]; o[18] = P.inertia[0][1]; o[19] = P.inertia[0][2];
    }
    return RP_OK;
}
// ---- composite shapes (include/rapier_hip.h: rp_compound_create / rp_trimesh_create / rp_heightfield_create) -------------------------
// Shape::compute_aabb(pos) of a part, in the words of the device's prim_aabb_at (rp_composite.h) and the oracle's (ro_composite.h):
// he in the c_he layout (cylinder / cone: radius, half height, radius; capsule: half height, radius, axis; polyhedron: its box)
static void h_prim_aabb_at(int core, const float he[3], float border, Pose at, V3 &mn, V3 &mx) {
    if (core == RP_SHAPE_CUBOID || core >= RP_SHAPE_CYLINDER) {
        float m[3][3]; quat_to_mat(at.r, m);
        V3 h = v3(fabsf(m[0][0]) * he[0] + fabsf(m[0][1]) * he[1] + fabsf(m[0][2]) * he[2],
                  fabsf(m[1][0]) * he[0] + fabsf(m[1][1]) * he[1] + fabsf(m[1][2]) * he[2],
                  fabsf(m[2][0]) * he[0] + fabsf(m[2][1]) * he[1] + fabsf(m[2][2]) * he[2]);
        mn = at.t - h; mx = at.t + h;
    } else if (core == RP_SHAPE_CAPSULE) {
        const int axis = (int)he[2];
        V3 e = v3(axis == 0 ? 1.0f : 0.0f, axis == 1 ? 1.0f : 0.0f, axis == 2 ? 1.0f : 0.0f);
        V3 pa = pose_tp(at, e * -he[0]), pb = pose_tp(at, e * he[0]);
        V3 r = v3(he[1], he[1], he[1]);
        mn = v3(rp_min(pa.x, pb.x), rp_min(pa.y, pb.y), rp_min(pa.z, pb.z)) - r;
        mx = v3(rp_max(pa.x, pb.x), rp_max(pa.y, pb.y), rp_max(pa.z, pb.z)) + r;
    } else {
        V3 h = v3(he[0], he[0], he[0]);
        mn = at.t - h; mx = at.t + h;
    }
    if (border > 0.0f) { V3 b = v3(border, border, border); mn = mn - b; mx = mx + b; }
}
// the part's half extents in the c_he layout (pack_collider's rule)
static void part_che(const rp_world *w, const rp_collider_desc &c, int poly, float he[3]) {
    const int core = core_shape(c.shape);
    he[0] = c.half_extents[0]; he[1] = c.half_extents[1]; he[2] = c.half_extents[2];
    if (core == RP_SHAPE_CYLINDER || core == RP_SHAPE_CONE) { he[0] = c.half_extents[1]; he[1] = c.half_extents[0]; he[2] = c.half_extents[1]; }
    if (core == RP_SHAPE_CONVEX_POLYHEDRON) { const HostPolyhedron &P = w->polys[(size_t)poly]; he[0] = P.half[0]; he[1] = P.half[1]; he[2] = P.half[2]; }
}
static Pose part_pose(const rp_collider_desc &c) {
    Pose p; p.t = v3(c.translation[0], c.translation[1], c.translation[2]); p.r = q4(c.rotation[0], c.rotation[1], c.rotation[2], c.rotation[3]);
    return p;
}
static int upload_composites(rp_world *w) {
    std::vector<int4> hdr; std::vector<float4> mn, mx, a, b, c; std::vector<float> border;
    for (const rp_world::HostComposite &C : w->comps) {
        int4 h; h.x = C.kind; h.y = (int)mn.size(); h.z = C.count(); h.w = 0;
        hdr.push_back(h);
        for (int i = 0; i < C.count(); ++i) {
            mn.push_back(mk4(C.smin[3 * i], C.smin[3 * i + 1], C.smin[3 * i + 2], 0.0f)); mx.push_back(mk4(C.smax[3 * i], C.smax[3 * i + 1], C.smax[3 * i + 2], 0.0f));
            if (C.kind == RP_SHAPE_COMPOUND) {
                const rp_collider_desc &d = C.parts[(size_t)i];
                float he[3]; part_che(w, d, C.part_poly[(size_t)i], he);
                float4 ra = mk4(he[0], he[1], he[2], 0.0f);
                if (core_shape(d.shape) == RP_SHAPE_CONVEX_POLYHEDRON) { int id = C.part_poly[(size_t)i]; memcpy(&ra.w, &id, sizeof(int)); }
                float4 rb = mk4(d.translation[0], d.translation[1], d.translation[2], 0.0f); int sh = d.shape; memcpy(&rb.w, &sh, sizeof(int));
                a.push_back(ra); b.push_back(rb); c.push_back(mk4(d.rotation[0], d.rotation[1], d.rotation[2], d.rotation[3])); border.push_back(shape_border(d));
            } else {
                const float *t = &C.tri[9 * (size_t)i];
                a.push_back(mk4(t[0], t[1], t[2], 0.0f)); b.push_back(mk4(t[3], t[4], t[5], 0.0f)); c.push_back(mk4(t[6], t[7], t[8], 0.0f)); border.push_back(0.0f);
            }
        }
    }
    HIPCHK(w, hipStreamSynchronize(w->stream));
    for (int k = 0; k < 7; ++k) if (w->cm_dev[k]) { HIPCHK(w, hipFree(w->cm_dev[k])); w->cm_dev[k] = nullptr; }
    const void *src[7] = {hdr.data(), mn.data(), mx.data(), a.data(), b.data(), c.data(), border.data()};
    const size_t bytes[7] = {hdr.size() * sizeof(int4), mn.size() * sizeof(float4), mx.size() * sizeof(float4), a.size() * sizeof(float4), b.size() * sizeof(float4), c.size() * sizeof(float4), border.size() * sizeof(float)};
    for (int k = 0; k < 7; ++k) {
        if (bytes[k] == 0) continue;
        HIPCHK(w, hipMalloc(&w->cm_dev[k], bytes[k]));
        HIPCHK(w, hipMemcpyAsync(w->cm_dev[k], src[k], bytes[k], hipMemcpyHostToDevice, w->stream));
    }
    if (!w->cm_dev[7]) { // the cluster workspace of k_np_composite: RP_CM_WS_F4 float4 per thread
        w->dw.cm_ws_threads = 32 * 128;
        HIPCHK(w, hipMalloc(&w->cm_dev[7], (size_t)w->dw.cm_ws_threads * RP_CM_WS_F4_HOST * sizeof(float4)));
    }
    HIPCHK(w, hipStreamSynchronize(w->stream));
    w->dw.cm_hdr = (int4 *)w->cm_dev[0]; w->dw.cm_min = (float4 *)w->cm_dev[1]; w->dw.cm_max = (float4 *)w->cm_dev[2];
    w->dw.cm_a = (float4 *)w->cm_dev[3]; w->dw.cm_b = (float4 *)w->cm_dev[4]; w->dw.cm_c = (float4 *)w->cm_dev[5]; w->dw.cm_border = (float *)w->cm_dev[6];
    w->dw.cm_ws = (float4 *)w->cm_dev[7]; w->dw.cm_ws_threads = 32 * 128;
    return RP_OK;
}
static int composite_registered(rp_world *w, rp_world::HostComposite &&C, int32_t *id_out) {
    w->comps.push_back(std::move(C));
    *id_out = (int32_t)w->comps.size() - 1;
    if (w->finalized) { // the tables are replaced: no launch may still read the old ones, the graphs hold the old pointers
        HIPCHK(w, hipSetDevice(w->device));
        int r = settle(w); if (r != RP_OK) return r;
        r = upload_composites(w); if (r != RP_OK) return r;
        destroy_graphs(w);
    }
    return RP_OK;
}
extern "C" int32_t rp_compound_create(rp_world *w, int32_t n_parts, const rp_collider_desc *parts, int32_t *id_out) {
    if (!w || !parts || !id_out || n_parts < 1) { if (w) w->err = "rp_compound_create: at least one part"; return RP_ERR_INVALID; }
    rp_world::HostComposite C; C.kind = RP_SHAPE_COMPOUND;
    for (int i = 0; i < n_parts; ++i) {
        rp_collider_desc d = parts[i];
        if (d.shape < RP_SHAPE_BALL || d.shape > RP_SHAPE_ROUND_CONVEX_POLYHEDRON || d.shape == RP_SHAPE_HALFSPACE) { w->err = "rp_compound_create: a part is a primitive or round primitive (no half-space, no composite)"; return RP_ERR_INVALID; }
        int poly = -1;
        if (core_shape(d.shape) == RP_SHAPE_CONVEX_POLYHEDRON) {
            if (!(d.half_extents[0] >= 0.0f && d.half_extents[0] < (float)w->polys.size() && d.half_extents[0] == std::floor(d.half_extents[0]))) { w->err = "rp_compound_create: a polyhedron part's half_extents[0] holds the id rp_convex_polyhedron_create returned"; return RP_ERR_INVALID; }
            poly = (int)d.half_extents[0];
        }
        const float qn = std::sqrt(d.rotation[0] * d.rotation[0] + d.rotation[1] * d.rotation[1] + d.rotation[2] * d.rotation[2] + d.rotation[3] * d.rotation[3]);
        const float qi = qn > 0.0f ? 1.0f / qn : 1.0f;
        d.rotation[0] *= qi; d.rotation[1] *= qi; d.rotation[2] *= qi; d.rotation[3] = qn > 0.0f ? d.rotation[3] * qi : 1.0f;
        if (poly >= 0) { // a polyhedron is stored recentred: the offset rides in the part's pose
            float r[3]; h_qrot(d.rotation, w->polys[(size_t)poly].centre, r);
            d.translation[0] = r[0] + d.translation[0]; d.translation[1] = r[1] + d.translation[1]; d.translation[2] = r[2] + d.translation[2];
        }
        C.parts.push_back(d); C.part_poly.push_back(poly);
    }
    V3 bmn = v3(0, 0, 0), bmx = bmn;
    for (int i = 0; i < n_parts; ++i) {
        float he[3]; part_che(w, C.parts[(size_t)i], C.part_poly[(size_t)i], he);
        V3 mn, mx; h_prim_aabb_at(core_shape(C.parts[(size_t)i].shape), he, shape_border(C.parts[(size_t)i]), part_pose(C.parts[(size_t)i]), mn, mx);
        if (i == 0) { bmn = mn; bmx = mx; }
        else { bmn = v3(rp_min(bmn.x, mn.x), rp_min(bmn.y, mn.y), rp_min(bmn.z, mn.z)); bmx = v3(rp_max(bmx.x, mx.x), rp_max(bmx.y, mx.y), rp_max(bmx.z, mx.z)); }
    }
    const V3 centre = (bmn + bmx) * 0.5f, half = (bmx - bmn) * 0.5f;
    C.centre[0] = centre.x; C.centre[1] = centre.y; C.centre[2] = centre.z; C.half[0] = half.x; C.half[1] = half.y; C.half[2] = half.z;
    for (int i = 0; i < n_parts; ++i) { // recentred on the local AABB (the centre is folded into the collider's pose, like a polyhedron's)
        rp_collider_desc &d = C.parts[(size_t)i];
        const V3 t = v3(d.translation[0], d.translation[1], d.translation[2]) - centre;
        d.translation[0] = t.x; d.translation[1] = t.y; d.translation[2] = t.z;
        float he[3]; part_che(w, d, C.part_poly[(size_t)i], he);
        V3 mn, mx; h_prim_aabb_at(core_shape(d.shape), he, shape_border(d), part_pose(d), mn, mx);
        C.smin.push_back(mn.x); C.smin.push_back(mn.y); C.smin.push_back(mn.z); C.smax.push_back(mx.x); C.smax.push_back(mx.y); C.smax.push_back(mx.z);
    }
    return composite_registered(w, std::move(C), id_out);
}
extern "C" int32_t rp_trimesh_create(rp_world *w, int32_t nv, const float *xyz, int32_t nt, const uint32_t *idx, int32_t *id_out) {
    if (!w || !xyz || !idx || !id_out || nv < 3 || nt < 1) { if (w) w->err = "rp_trimesh_create: at least three vertices and one triangle"; return RP_ERR_INVALID; }
    for (int i = 0; i < 3 * nt; ++i) if (idx[i] >= (uint32_t)nv) { w->err = "rp_trimesh_create: a triangle names a vertex that does not exist"; return RP_ERR_INVALID; }
    rp_world::HostComposite C; C.kind = RP_SHAPE_TRIMESH;
    V3 bmn = v3(xyz[0], xyz[1], xyz[2]), bmx = bmn;
    for (int i = 1; i < nv; ++i) {
        const V3 p = v3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        bmn = v3(rp_min(bmn.x, p.x), rp_min(bmn.y, p.y), rp_min(bmn.z, p.z)); bmx = v3(rp_max(bmx.x, p.x), rp_max(bmx.y, p.y), rp_max(bmx.z, p.z));
    }
    const V3 centre = (bmn + bmx) * 0.5f, half = (bmx - bmn) * 0.5f;
    C.centre[0] = centre.x; C.centre[1] = centre.y; C.centre[2] = centre.z; C.half[0] = half.x; C.half[1] = half.y; C.half[2] = half.z;
    for (int t = 0; t < nt; ++t) {
        V3 q[3];
        for (int k = 0; k < 3; ++k) { const uint32_t vi = idx[3 * t + k]; q[k] = v3(xyz[3 * vi], xyz[3 * vi + 1], xyz[3 * vi + 2]) - centre; C.tri.push_back(q[k].x); C.tri.push_back(q[k].y); C.tri.push_back(q[k].z); }
        V3 mn = q[0], mx = q[0];
        for (int k = 1; k < 3; ++k) { mn = v3(rp_min(mn.x, q[k].x), rp_min(mn.y, q[k].y), rp_min(mn.z, q[k].z)); mx = v3(rp_max(mx.x, q[k].x), rp_max(mx.y, q[k].y), rp_max(mx.z, q[k].z)); }
        C.smin.push_back(mn.x); C.smin.push_back(mn.y); C.smin.push_back(mn.z); C.smax.push_back(mx.x); C.smax.push_back(mx.y); C.smax.push_back(mx.z);
    }
    return composite_registered(w, std::move(C), id_out);
}
extern "C" int32_t rp_heightfield_create(rp_world *w, int32_t nrows, int32_t ncols, const float *heights, const float scale[3], int32_t *id_out) {
    if (!w || !heights || !scale || !id_out || nrows < 2 || ncols < 2) { if (w) w->err = "rp_heightfield_create: at least 2 x 2 heights"; return RP_ERR_INVALID; }
    const int nv = nrows * ncols, nt = 2 * (nrows - 1) * (ncols - 1);
    std::vector<float> xyz((size_t)3 * nv); std::vector<uint32_t> idx((size_t)3 * nt);
    for (int r = 0; r < nrows; ++r) for (int c = 0; c < ncols; ++c) {
        const int i = r * ncols + c;
        xyz[3 * (size_t)i] = ((float)c / (float)(ncols - 1) - 0.5f) * scale[0]; xyz[3 * (size_t)i + 1] = heights[i] * scale[1]; xyz[3 * (size_t)i + 2] = ((float)r / (float)(nrows - 1) - 0.5f) * scale[2];
    }
    size_t t = 0;
    for (int r = 0; r + 1 < nrows; ++r) for (int c = 0; c + 1 < ncols; ++c) { // heightfield3.rs: two triangles per cell, cut along (r, c) -> (r + 1, c + 1)
        const uint32_t p00 = (uint32_t)(r * ncols + c), p01 = p00 + 1, p10 = p00 + (uint32_t)ncols, p11 = p10 + 1;
        idx[3 * t] = p00; idx[3 * t + 1] = p10; idx[3 * t + 2] = p11; ++t;
        idx[3 * t] = p00; idx[3 * t + 1] = p11; idx[3 * t + 2] = p01; ++t;
    }
    return rp_trimesh_create(w, nv, xyz.data(), nt, idx.data(), id_out);
}
extern "C" int32_t rp_colliders_insert(rp_world *w, int32_t n, const rp_collider_desc *descs, const uint64_t *parents, uint64_t *handles_out) {
    if (!w || n < 0 || (n > 0 && !descs)) return RP_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        const rp_collider_desc &cd = descs[i];
        if (shape_composite(cd.shape)) {
            if (!(cd.half_extents[0] >= 0.0f && cd.half_extents[0] < (float)w->comps.size() && cd.half_extents[0] == std::floor(cd.half_extents[0])) ||
                w->comps[(size_t)cd.half_extents[0]].kind != cd.shape) { w->err = "rp_colliders_insert: a composite collider's half_extents[0] holds the id rp_compound_create / rp_trimesh_create / rp_heightfield_create returned (for that shape)"; return RP_ERR_INVALID; }
            if (cd.shape == RP_SHAPE_TRIMESH && parents && parents[i] != RP_INVALID_HANDLE) {
                const int pb = body_of(w, parents[i]);
                if (pb >= 0 && pb < (int)w->bodies.size() && w->bodies[pb].d.body_type == RP_BODY_DYNAMIC) { w->err = "rp_colliders_insert: a triangle mesh / height field needs a fixed or kinematic parent (or none)"; return RP_ERR_INVALID; }
            }
            continue;
        }
        if (cd.shape < RP_SHAPE_BALL || cd.shape > RP_SHAPE_ROUND_CONVEX_POLYHEDRON) { w->err = "rp_colliders_insert: unknown shape (ball, cuboid, capsule, half-space, cylinder, cone, convex polyhedron and their round variants are implemented)"; return RP_ERR_INVALID; }
        if (cd.shape >= RP_SHAPE_ROUND_CUBOID && !(cd.border_radius > 0.0f)) { w->err = "rp_colliders_insert: a round shape needs a positive border_radius"; return RP_ERR_INVALID; }
        if (core_shape(cd.shape) == RP_SHAPE_CONVEX_POLYHEDRON && !(cd.half_extents[0] >= 0.0f && cd.half_extents[0] < (float)w->polys.size() && cd.half_extents[0] == std::floor(cd.half_extents[0]))) { w->err = "rp_colliders_insert: a convex polyhedron's half_extents[0] holds the id rp_convex_polyhedron_create returned"; return RP_ERR_INVALID; }
        if ((core_shape(cd.shape) == RP_SHAPE_CYLINDER || core_shape(cd.shape) == RP_SHAPE_CONE) && !(cd.half_extents[0] > 0.0f && cd.half_extents[1] > 0.0f)) { w->err = "rp_colliders_insert: cylinder / cone half_extents = (half_height, radius, -), both positive"; return RP_ERR_INVALID; }
        if (cd.shape == RP_SHAPE_HALFSPACE) {
            const float *nn = cd.half_extents; const float l2 = nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2];
            if (!(std::fabs(l2 - 1.0f) <= 1.0e-3f)) { w->err = "rp_colliders_insert: a half-space's half_extents hold its unit outward normal"; return RP_ERR_INVALID; }
            if (parents && parents[i] != RP_INVALID_HANDLE) {
                const int pb = body_of(w, parents[i]);
                if (pb >= 0 && pb < (int)w->bodies.size() && w->bodies[pb].d.body_type == RP_BODY_DYNAMIC) { w->err = "rp_colliders_insert: a half-space needs a fixed or kinematic parent (or none)"; return RP_ERR_INVALID; }
            }
        }
        if (cd.shape == RP_SHAPE_CAPSULE && (cd.half_extents[2] != 0.0f && cd.half_extents[2] != 1.0f && cd.half_extents[2] != 2.0f)) { w->err = "rp_colliders_insert: capsule half_extents = (half_height, radius, axis) with axis 0, 1 or 2"; return RP_ERR_INVALID; }
    }
    if ((long long)w->colliders.size() + n >= (1ll << 24)) { w->err = "rp_colliders_insert: more than 2^24 - 1 colliders (the broad-phase grid's one-word entries)"; return RP_ERR_CAPACITY; }
    static const bool no_reuse = getenv("RP_NO_ARENA_REUSE") != nullptr;
    const int n_reuse = no_reuse ? 0 : std::min<int>(n, (int)w->coll_free.size()); // Arena::insert: removed slots first (see rp_bodies_insert)
    { // parents and per-body collider counts of the WHOLE batch, before anything changes and before the split below
        std::vector<std::pair<int, int>> added; int added_free = 0; // (parent, colliders this batch gives it)
        for (int i = 0; i < n; ++i) {
            int parent = -1;
            if (parents && parents[i] != RP_INVALID_HANDLE) {
                parent = body_of(w, parents[i]);
                if (parent < 0 || w->bodies[parent].quarantined) { w->err = "rp_colliders_insert: invalid parent handle (unknown, stale, removed or quarantined body)"; return RP_ERR_INVALID; }
            }
            int count = 0;
            if (parent < 0) count = ++added_free;
            else { auto it = std::find_if(added.begin(), added.end(), [&](const std::pair<int, int> &e) { return e.first == parent; }); if (it == added.end()) { added.push_back({parent, 1}); count = 1; } else count = ++it->second; }
            if ((parent >= 0 ? w->bodies[parent].next_ord : w->next_free_ord) + count > (parent >= 0 ? 4096 : (1 << 20))) { w->err = "rp_colliders_insert: more than 4,096 colliders on one body (or 2^20 without a parent)"; return RP_ERR_CAPACITY; }
        }
    }
    if (n_reuse > 0 && n_reuse < n) {
        int r = rp_colliders_insert(w, n_reuse, descs, parents, handles_out);
        return r != RP_OK ? r : rp_colliders_insert(w, n - n_reuse, descs + n_reuse, parents ? parents + n_reuse : nullptr, handles_out ? handles_out + n_reuse : nullptr);
    }
    const bool reuse = n_reuse > 0;
    const bool in_place = w->finalized && (reuse || (int)w->colliders.size() + n <= w->cap_colliders); // see rp_bodies_insert
    if (n > 0 && w->finalized) {
        HIPCHK(w, hipSetDevice(w->device));
        int r = in_place ? settle(w) : grow_begin(w);
        if (r != RP_OK) return r;
    }
    for (int i = 0; i < n; ++i) if (parents && parents[i] != RP_INVALID_HANDLE) {
        const int parent = body_of(w, parents[i]);
        if (parent < 0 || w->bodies[parent].quarantined) { w->err = "rp_colliders_insert: invalid parent handle (unknown, stale, removed or quarantined body)"; return RP_ERR_INVALID; }
    }
    if (reuse && w->finalized) { int r = purge_dead_pairs(w); if (r != RP_OK) return r; } // no pair may still name the slots' previous occupants
    std::vector<int> new_slots;
    for (int i = 0; i < n; ++i) {
        const int parent = (parents && parents[i] != RP_INVALID_HANDLE) ? body_of(w, parents[i]) : -1;
        int &ord_counter = parent >= 0 ? w->bodies[parent].next_ord : w->next_free_ord;
        if (ord_counter >= (parent >= 0 ? 4096 : (1 << 20))) { w->err = "rp_colliders_insert: more than 4,096 colliders on one body (or 2^20 without a parent)"; return RP_ERR_CAPACITY; }
        int ci;
        rp_collider_desc cd = descs[i];
        int poly = -1, comp = -1;
        if (shape_composite(cd.shape)) { // stored recentred on its local AABB, like a polyhedron: the centre rides in the collider's pose, half_extents = the box
            comp = (int)cd.half_extents[0];
            const rp_world::HostComposite &C = w->comps[(size_t)comp];
            float qn = std::sqrt(cd.rotation[0] * cd.rotation[0] + cd.rotation[1] * cd.rotation[1] + cd.rotation[2] * cd.rotation[2] + cd.rotation[3] * cd.rotation[3]);
            float qi = qn > 0.0f ? 1.0f / qn : 1.0f;
            const float q[4] = {cd.rotation[0] * qi, cd.rotation[1] * qi, cd.rotation[2] * qi, qn > 0.0f ? cd.rotation[3] * qi : 1.0f};
            float r[3]; h_qrot(q, C.centre, r);
            cd.translation[0] = r[0] + cd.translation[0]; cd.translation[1] = r[1] + cd.translation[1]; cd.translation[2] = r[2] + cd.translation[2];
            cd.half_extents[0] = C.half[0]; cd.half_extents[1] = C.half[1]; cd.half_extents[2] = C.half[2];
            cd.border_radius = 0.0f;
        }
        if (core_shape(cd.shape) == RP_SHAPE_CONVEX_POLYHEDRON) { // stored recentred on its local AABB: the centre rides in the collider's pose, half_extents = the box (rp_polyhedron.h)
            poly = (int)cd.half_extents[0];
            const HostPolyhedron &P = w->polys[(size_t)poly];
            float qn = std::sqrt(cd.rotation[0] * cd.rotation[0] + cd.rotation[1] * cd.rotation[1] + cd.rotation[2] * cd.rotation[2] + cd.rotation[3] * cd.rotation[3]);
            float qi = qn > 0.0f ? 1.0f / qn : 1.0f;
            const float q[4] = {cd.rotation[0] * qi, cd.rotation[1] * qi, cd.rotation[2] * qi, qn > 0.0f ? cd.rotation[3] * qi : 1.0f};
            float r[3]; h_qrot(q, P.centre, r);
            cd.translation[0] = r[0] + cd.translation[0]; cd.translation[1] = r[1] + cd.translation[1]; cd.translation[2] = r[2] + cd.translation[2];
            cd.half_extents[0] = P.half[0]; cd.half_extents[1] = P.half[1]; cd.half_extents[2] = P.half[2];
        }
        if (reuse) {
            ci = w->coll_free.back(); w->coll_free.pop_back();
            w->collider_ord[(size_t)ci] = ord_counter++; w->colliders[(size_t)ci] = cd; w->collider_parent[(size_t)ci] = parent; w->collider_removed[(size_t)ci] = 0; w->collider_poly[(size_t)ci] = poly; w->collider_comp[(size_t)ci] = comp; w->collider_sub[(size_t)ci] = w->cur_sub;
        } else {
            ci = (int)w->colliders.size();
            w->collider_ord.push_back(ord_counter++); w->colliders.push_back(cd); w->collider_parent.push_back(parent); w->collider_removed.push_back(0); w->coll_gen.push_back(0); w->collider_poly.push_back(poly); w->collider_comp.push_back(comp); w->collider_sub.push_back(w->cur_sub);
        }
        w->coll_gen[(size_t)ci] = w->coll_arena_gen;
        new_slots.push_back(ci);
        if (parent >= 0) w->bodies[parent].cols.push_back(ci);
        if (parent >= 0) { w->bodies[parent].ncolliders++; recompute_mass(w, parent); }
        if (descs[i].restitution > 0.0f) w->has_restitution = true;
        if (handles_out) handles_out[i] = ((uint64_t)w->coll_arena_gen << 32) | (uint64_t)(uint32_t)ci;
        if (in_place) {
            int r = reuse ? reset_row(w, DOM_COLL, ci) : RP_OK;
            if (r == RP_OK) r = upload_collider_row(w, ci);
            if (r == RP_OK && parent >= 0) r = upload_body_row_mass(w, parent);
            if (r == RP_OK && parent >= 0) r = refresh_joint_frames(w, parent); // the local centre of mass moved
            if (r == RP_OK && parent >= 0 && w->bodies[parent].d.body_type == RP_BODY_DYNAMIC) {
                r = upload_collider_chain(w, parent);
            }
            if (r != RP_OK) return r;
        }
    }
    if (in_place && n > 0) {
        w->dw.n_colliders = (int)w->colliders.size();
        // the world-wide facts can only be switched ON by an insertion: looked up on the new rows alone (the full scans of
        // world_has_* walk every collider — a million in b3d_large_world, per dropped sphere)
        for (size_t ci : new_slots) {
            const rp_collider_desc &c = w->colliders[ci];
            const int p = w->collider_parent[ci];
            if (c.active_events & RP_EVENTS_CONTACT_FORCE) w->dw.has_force_events = 1;
            if (c.sensor) w->dw.has_sensors = 1;
            if (c.shape >= RP_SHAPE_CYLINDER) w->dw.has_convex = 1;
            if (shape_composite(c.shape)) w->dw.has_composite = 1;
            if (p >= 0 && w->bodies[p].d.body_type == RP_BODY_DYNAMIC) {
                const float *t = c.translation, *r = c.rotation;
                const bool at_origin = t[0] == 0.0f && t[1] == 0.0f && t[2] == 0.0f && r[0] == 0.0f && r[1] == 0.0f && r[2] == 0.0f;
                if (!at_origin || w->bodies[p].ncolliders > 1) w->compound = true; // (world_has_compound_bodies)
            }
        }
        refresh_ccd_facts(w);
        HIPCHK(w, hipStreamSynchronize(w->stream));
        destroy_graphs(w);
        return after_topology_edit(w, true);
    }
    if (w->carry) return finalize(w);
    return RP_OK;
}
extern "C" int32_t rp_impulse_joints_insert(rp_world *w, int32_t n, const rp_joint_desc *descs, uint64_t *handles_out) {
    if (!w || n < 0 || (n > 0 && !descs)) return RP_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        const rp_joint_desc &j = descs[i];
        if (j.body1 < 0 || j.body2 < 0 || j.body1 >= (int)w->bodies.size() || j.body2 >= (int)w->bodies.size()) { w->err = "rp_impulse_joints_insert: invalid body index"; return RP_ERR_INVALID; }
        if (w->bodies[(size_t)j.body1].removed || w->bodies[(size_t)j.body2].removed) { w->err = "rp_impulse_joints_insert: a joint names a removed body (a free arena slot)"; return RP_ERR_INVALID; }
        if ((j.locked_axes & ~0x3fu) != 0 || (j.limit_axes & ~0x3fu) != 0 || (j.motor_axes & ~0x3fu) != 0) { w->err = "rp_impulse_joints_insert: locked_axes / limit_axes / motor_axes must be JointAxesMasks (coupled axes are not implemented on the device path)"; return RP_ERR_INVALID; }
        for (int a = 0; a < 6; ++a) if (j.motors[a].model != RP_MOTOR_ACCELERATION_BASED && j.motors[a].model != RP_MOTOR_FORCE_BASED) { w->err = "rp_impulse_joints_insert: unknown motor model"; return RP_ERR_INVALID; }
    }
    if (n > 0) { int r = grow_begin(w); if (r != RP_OK) return r; } // the joint arrays have no spare rows: the world moves to larger ones
    for (int i = 0; i < n; ++i) {
        w->joints.push_back(descs[i]);
        w->pending_wake.push_back(descs[i].body1); w->pending_wake.push_back(descs[i].body2);
        w->joint_removed.push_back(0);
        if (handles_out) handles_out[i] = (uint64_t)(w->joints.size() - 1);
    }
    if (w->carry) return finalize(w);
    return RP_OK;
}

template <typename T>
static int dalloc(rp_world *w, T *&p, size_t count, int fill_byte = 0, int dom = DOM_NONE, int planes = 1, int per = 1) {
    void *q = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (locked_hipMallocBytes((void **)&q, bytes) != hipSuccess) { w->err = "hipMalloc failed"; return RP_ERR_DEVICE; }
    if (hipMemsetAsync(q, fill_byte, bytes, w->stream) != hipSuccess) { w->err = "hipMemset failed"; return RP_ERR_DEVICE; }
    AllocRec a; a.ptr = q; a.off = (size_t)((char *)&p - (char *)&w->dw); a.elem = sizeof(T); a.per = (size_t)per; a.planes = planes; a.dom = dom; a.fill = fill_byte;
    a.stride = count / ((size_t)planes * (size_t)per);
    w->allocs.push_back(a);
    p = (T *)q;
    return RP_OK;
}
// DA / DAF: scratch or host-authoritative arrays; DAC / DAFC: persistent rows carried over when the world grows (domain, planes[, per])
#define DA(ptr, count) do { int r_ = dalloc(w, ptr, (size_t)(count)); if (r_ != RP_OK) return r_; } while (0)
// DAS: scratch that every reader finds written by an earlier kernel of the same step (no rest state).  RP_TEST_POISON=1 fills it with
// 0xFF bytes (NaN / -1) instead of zeros: a kernel that reads such an array before it was written shows up in the parity tests
#ifdef RP_TESTING // (the testing build, `make testing` -> librapier_hip_testing.so: the product library carries no test hook)
#define DAS(ptr, count) do { static const int poison_ = (getenv("RP_TEST_POISON") && atoi(getenv("RP_TEST_POISON"))) ? 0xff : 0; int r_ = dalloc(w, ptr, (size_t)(count), poison_); if (r_ != RP_OK) return r_; } while (0)
#else
#define DAS(ptr, count) DA(ptr, count)
#endif
#define DAF(ptr, count, fill) do { int r_ = dalloc(w, ptr, (size_t)(count), fill); if (r_ != RP_OK) return r_; } while (0)
#define DAC(ptr, count, ...) do { int r_ = dalloc(w, ptr, (size_t)(count), 0, __VA_ARGS__); if (r_ != RP_OK) return r_; } while (0)
#define DAFC(ptr, count, fill, ...) do { int r_ = dalloc(w, ptr, (size_t)(count), fill, __VA_ARGS__); if (r_ != RP_OK) return r_; } while (0)

template <typename T>
static int upload(rp_world *w, T *dst, const std::vector<T> &src) {
    if (src.empty()) return RP_OK;
    if (hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, w->stream) != hipSuccess) { w->err = "upload failed"; return RP_ERR_DEVICE; }
    return RP_OK;
}
#define UP(dst, vec) do { int r_ = upload(w, dst, vec); if (r_ != RP_OK) return r_; } while (0)

static int next_pow2(long long x) { long long p = 1; while (p < x) p <<= 1; return (int)p; }
static float4 mk4(float x, float y, float z, float w_) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w_; return r; }

// One body / collider row of the SoA device world from the host mirrors (finalize and incremental inserts).
struct BodyRow { float4 pos, rot, lv, av, lci, ipi, pfr, damp, slp, spt, spr, npos, nrot; int fl, slabel; };
static BodyRow pack_body(const HostBody &b) {
    const rp_body_desc &bd = b.d;
    BodyRow o;
    float qn = std::sqrt(bd.rotation[0] * bd.rotation[0] + bd.rotation[1] * bd.rotation[1] + bd.rotation[2] * bd.rotation[2] + bd.rotation[3] * bd.rotation[3]);
    float qi = qn > 0.0f ? 1.0f / qn : 1.0f;
    o.pos = mk4(bd.translation[0], bd.translation[1], bd.translation[2], 0);
    o.rot = mk4(bd.rotation[0] * qi, bd.rotation[1] * qi, bd.rotation[2] * qi, qn > 0.0f ? bd.rotation[3] * qi : 1.0f);
    o.lv = mk4(bd.linvel[0], bd.linvel[1], bd.linvel[2], 0); o.av = mk4(bd.angvel[0], bd.angvel[1], bd.angvel[2], 0);
    o.lci = mk4(b.lcom[0], b.lcom[1], b.lcom[2], b.inv_mass);
    o.ipi = mk4(b.inv_pi[0], b.inv_pi[1], b.inv_pi[2], b.max_extent); // w: max_extent again (next to what body_writeback loads anyway)
    o.pfr = mk4(b.pframe[0], b.pframe[1], b.pframe[2], b.pframe[3]);
    o.damp = mk4(bd.linear_damping, bd.angular_damping, bd.gravity_scale, b.ccd_thickness);
    int fl = ((b.removed ? RP_BODY_FIXED : bd.body_type) & RP_BF_TYPE_MASK);
    if (bd.gyroscopic && bd.body_type == RP_BODY_DYNAMIC) fl |= RP_BF_GYRO; // gyroscopic forces: dynamic bodies only (worker.rs:86)
    if (bd.allow_fast_rotation) fl |= RP_BF_FASTROT;
    fl |= ((int)(bd.dominance & 0xff)) << RP_BF_DOM_SHIFT;
    fl |= ((int)(bd.locked_axes & 0x3fu)) << RP_BF_LOCK_SHIFT;
    if (b.sleeping && !b.removed && bd.body_type != RP_BODY_FIXED) fl |= RP_BF_SLEEPING;
    if (bd.ccd_enabled && bd.body_type == RP_BODY_DYNAMIC && !b.removed) fl |= RP_BF_CCD_ENABLED; // a bullet (sweeps.rs:29-31)
    o.fl = fl;
    // RigidBodyActivation::active() / cannot_sleep() — rigid_body_components.rs:1354-1385
    o.slp = mk4(b.sleep_timer, bd.can_sleep ? 0.05f : -1.0f, bd.can_sleep ? 0.5f : -1.0f, 0.5f);
    o.spt = mk4(b.sprev[0], b.sprev[1], b.sprev[2], b.max_extent);
    o.spr = mk4(b.sprev[3], b.sprev[4], b.sprev[5], b.sprev[6]);
    o.slabel = b.slabel;
    o.npos = b.has_next ? mk4(b.next[0], b.next[1], b.next[2], 0) : o.pos;
    o.nrot = b.has_next ? mk4(b.next[3], b.next[4], b.next[5], b.next[6]) : o.rot;
    return o;
}
#define PUT(arr, idx, val) HIPCHK(w, hipMemcpyAsync((arr) + (idx), &(val), sizeof(val), hipMemcpyHostToDevice, w->stream))
static int upload_body_row(rp_world *w, int i) {
    // hipMemcpyAsync from pageable stack memory is staged by the runtime before it returns
    const DevWorld &d = w->dw;
    BodyRow r = pack_body(w->bodies[i]);
    PUT(d.b_pos, i, r.pos); PUT(d.b_rot, i, r.rot); PUT(d.b_linvel, i, r.lv); PUT(d.b_angvel, i, r.av); PUT(d.b_lcom_invm, i, r.lci);
    PUT(d.b_invpi, i, r.ipi); PUT(d.b_pframe, i, r.pfr); PUT(d.b_damp, i, r.damp); PUT(d.b_flags, i, r.fl);
    PUT(d.b_sleep, i, r.slp); PUT(d.b_sprev_t, i, r.spt); PUT(d.b_sprev_r, i, r.spr); PUT(d.b_slabel, i, r.slabel);
    PUT(d.b_next_pos, i, r.npos); PUT(d.b_next_rot, i, r.nrot);
    int extra = w->bodies[i].d.additional_solver_iterations; PUT(d.b_extra, i, extra);
    return RP_OK;
}
struct ColliderRow { int ord; int parent, shape; float4 lp, lr, he, mat, fmn, fmx; int2 rules; uint2 groups; float2 events; };
static ColliderRow pack_collider(const rp_world *w, int i) {
    const rp_collider_desc &c = w->colliders[i];
    ColliderRow o;
    o.parent = w->collider_parent[i]; o.shape = c.shape; o.ord = w->collider_ord[i];
    float qn = std::sqrt(c.rotation[0] * c.rotation[0] + c.rotation[1] * c.rotation[1] + c.rotation[2] * c.rotation[2] + c.rotation[3] * c.rotation[3]);
    float qi = qn > 0.0f ? 1.0f / qn : 1.0f;
    o.lp = mk4(c.translation[0], c.translation[1], c.translation[2], 0);
    o.lr = mk4(c.rotation[0] * qi, c.rotation[1] * qi, c.rotation[2] * qi, qn > 0.0f ? c.rotation[3] * qi : 1.0f);
    o.he = mk4(c.half_extents[0], c.half_extents[1], c.half_extents[2], 0);
    if (core_shape(c.shape) == RP_SHAPE_CYLINDER || core_shape(c.shape) == RP_SHAPE_CONE) o.he = mk4(c.half_extents[1], c.half_extents[0], c.half_extents[1], 0); // (radius, half_height, radius): the local AABB's half extents
    if (core_shape(c.shape) == RP_SHAPE_CONVEX_POLYHEDRON) { int id = w->collider_poly[i]; memcpy(&o.he.w, &id, sizeof(int)); } // (half extents of the local box; w = the polyhedron's row in the cv_* tables, as bits)
    if (shape_composite(c.shape)) { int id = w->collider_comp[i]; memcpy(&o.he.w, &id, sizeof(int)); } // (likewise: the composite's row in cm_hdr)
    o.mat = mk4(c.friction, c.restitution, c.density, shape_border(c)); // (w: a round shape's border radius)
    o.rules.x = c.friction_rule; o.rules.y = c.restitution_rule;
    o.groups.x = w->collider_removed[i] ? 0u : c.collision_memberships; o.groups.y = w->collider_removed[i] ? 0u : c.collision_filter;
    // an "inverted" AABB: the first k_collider_update always rewrites it (and flags the broad phase)
    o.fmn = mk4(1.0f, 1.0f, 1.0f, 0); o.fmx = mk4(-1.0f, -1.0f, -1.0f, 0);
    int ev = (int)(c.active_events & 3u) | (c.sensor ? RP_EVENTS_SENSOR_BIT : 0); memcpy(&o.events.x, &ev, sizeof(int)); o.events.y = c.contact_force_event_threshold;
    return o;
}
// GenericJoint::transform_to_solver_body_space for the joints of body b after its local centre of mass changed (a collider was
// attached or removed): the frame of a non-fixed side lives in CoM space, local_frame.t - local_com
static int refresh_joint_frames(rp_world *w, int b) {
    if (!w->finalized || w->bodies[b].d.body_type == RP_BODY_FIXED) return RP_OK;
    const HostBody &hb = w->bodies[b];
    for (int k = 0; k < (int)w->active_joint_ids.size(); ++k) {
        int ji = w->active_joint_ids[k];
        const rp_joint_desc &jd = w->joints[ji];
        if (w->joint_removed[ji] || (jd.body1 != b && jd.body2 != b)) continue;
        int r;
        if (jd.body1 == b) {
            Pose f = joint_local_frame(jd.local_anchor1, jd.local_basis1); f.t = f.t - v3(hb.lcom[0], hb.lcom[1], hb.lcom[2]);
            if ((r = poke(w, w->dw.j_f1t + k, mk4(f.t.x, f.t.y, f.t.z, 0))) != RP_OK) return r;
        }
        if (jd.body2 == b) {
            Pose f = joint_local_frame(jd.local_anchor2, jd.local_basis2); f.t = f.t - v3(hb.lcom[0], hb.lcom[1], hb.lcom[2]);
            if ((r = poke(w, w->dw.j_f2t + k, mk4(f.t.x, f.t.y, f.t.z, 0))) != RP_OK) return r;
        }
    }
    return RP_OK;
}
static int upload_body_row_mass(rp_world *w, int i) { // mass properties only (a collider was attached / removed)
    const DevWorld &d = w->dw;
    BodyRow r = pack_body(w->bodies[i]);
    PUT(d.b_lcom_invm, i, r.lci); PUT(d.b_invpi, i, r.ipi); PUT(d.b_pframe, i, r.pfr);
    PUT((float *)(d.b_sprev_t + i) + 3, 0, r.spt.w); // max_extent follows the attached shapes
    PUT(d.b_damp, i, r.damp);                         // ... and so does ccd_thickness (damp.w)
    return RP_OK;
}
// b_collider / c_sibling of one body: its live colliders as a chain from the last one down (the fused step's validators walk it)
static int upload_collider_chain(rp_world *w, int parent) {
    int last = -1;
    if (w->bodies[(size_t)parent].d.body_type == RP_BODY_DYNAMIC)
        for (int q = 0; q < (int)w->colliders.size(); ++q)
            if (w->collider_parent[(size_t)q] == parent && !w->collider_removed[(size_t)q]) { PUT(w->dw.c_sibling, q, last); last = q; }
    PUT(w->dw.b_collider, parent, last);
    return RP_OK;
}
static int upload_collider_row(rp_world *w, int i) {
    const DevWorld &d = w->dw;
    ColliderRow r = pack_collider(w, i);
    PUT(d.c_sub, i, w->collider_sub[(size_t)i]);
    PUT(d.c_parent, i, r.parent); PUT(d.c_ord, i, r.ord); PUT(d.c_shape, i, r.shape); PUT(d.c_lpos, i, r.lp); PUT(d.c_lrot, i, r.lr); PUT(d.c_he, i, r.he); PUT(d.c_mat, i, r.mat);
    PUT(d.c_rules, i, r.rules); PUT(d.c_groups, i, r.groups); PUT(d.c_fatmin, i, r.fmn); PUT(d.c_fatmax, i, r.fmx); PUT(d.c_events, i, r.events);
    return RP_OK;
}

static bool world_sleep_enabled(const rp_world *w) {
    for (const HostBody &b : w->bodies) {
        if (b.removed) continue;
        if (b.d.body_type == RP_BODY_DYNAMIC && b.d.can_sleep) return true;
        // a kinematic body is sleep-eligible whenever its velocity is exactly zero, whatever can_sleep says
        if (b.d.body_type == RP_BODY_KINEMATIC_POSITION || b.d.body_type == RP_BODY_KINEMATIC_VELOCITY) return true;
    }
    return false;
}
// continuous-collision facts of the world: any bullet (a dynamic body with ccd_enabled), the thinnest dynamic body
static void refresh_ccd_facts(rp_world *w) {
    w->has_bullets = false; w->min_ccd_thickness = 3.402823466e+38f;
    for (const HostBody &b : w->bodies) {
        if (b.removed || b.quarantined || b.d.body_type != RP_BODY_DYNAMIC) continue;
        if (b.d.ccd_enabled) w->has_bullets = true;
        w->min_ccd_thickness = std::min(w->min_ccd_thickness, b.ccd_thickness);
    }
}
static bool world_has_sensors(const rp_world *w) {
    for (size_t i = 0; i < w->colliders.size(); ++i) if (!w->collider_removed[i] && w->colliders[i].sensor) return true;
    return false;
}
static bool world_has_convex(const rp_world *w) {
    for (size_t i = 0; i < w->colliders.size(); ++i) if (!w->collider_removed[i] && w->colliders[i].shape >= RP_SHAPE_CYLINDER) return true;
    return false;
}
static bool world_has_force_events(const rp_world *w) {
    for (size_t i = 0; i < w->colliders.size(); ++i) if (!w->collider_removed[i] && (w->colliders[i].active_events & RP_EVENTS_CONTACT_FORCE)) return true;
    return false;
}
static std::vector<unsigned long long> no_contact_keys(const rp_world *w) {
    std::vector<unsigned long long> k;
    for (size_t j = 0; j < w->joints.size(); ++j) {
        if (w->joint_removed[j] || w->joints[j].contacts_enabled) continue;
        unsigned lo = (unsigned)std::min(w->joints[j].body1, w->joints[j].body2), hi = (unsigned)std::max(w->joints[j].body1, w->joints[j].body2);
        k.push_back(((unsigned long long)lo << 32) | hi);
    }
    std::sort(k.begin(), k.end());
    return k;
}
static bool world_has_kinematic_pos(const rp_world *w) {
    for (const HostBody &b : w->bodies) if (!b.removed && b.d.body_type == RP_BODY_KINEMATIC_POSITION) return true;
    return false;
}
static int check_sleep_scope(rp_world *w) { (void)w; return RP_OK; } // impulse joints, sleeping and kinematic bodies mix freely

// Growth with state carry-over.  A live world that outgrows its row capacities — or receives a joint, whose arrays have no spare
// rows — moves to larger device arrays: grow_begin() parks the current device world, finalize() builds the larger one from the host
// mirrors (new rows included), and carry_over() copies every persistent row of the old world into it (bodies, colliders, pair
// slots with their manifolds / warm-start impulses / colours / recycle state, joints by device index, the step flags), rebuilds
// the pair hash for the new table size and marks the pair set / solver layout / joint layout dirty — exactly the state an
// in-place insertion leaves behind, so the next step matches the oracle bit for bit.  Host-authoritative arrays (mass
// properties, body -> collider / joint counts, joint limits and motors) come from the fresh upload.
static int grow_begin(rp_world *w) {
    if (!w->finalized) return RP_OK;
    HIPCHK(w, hipSetDevice(w->device));
    int r = download_state(w); // settles; the host mirrors of the body states are refreshed too
    if (r != RP_OK) return r;
    destroy_graphs(w);
    w->old_allocs.swap(w->allocs); w->allocs.clear();
    w->old_dw = w->dw;
    w->old_pinned = w->pinned_flags; w->pinned_flags = nullptr;
    w->old_active_joint_ids = w->active_joint_ids;
    w->carry = true;
    w->finalized = false;
    return RP_OK;
}
static int carry_over(rp_world *w) {
    const DevWorld &o = w->old_dw, &d = w->dw;
    for (const AllocRec &a : w->allocs) {
        if (a.dom == DOM_NONE) continue;
        const AllocRec *b = nullptr;
        for (const AllocRec &q : w->old_allocs) if (q.off == a.off) { b = &q; break; }
        if (!b || b->planes != a.planes || b->per != a.per || b->elem != a.elem) { w->err = "carry_over: allocation layout changed"; return RP_ERR_DEVICE; }
        size_t n = a.dom == DOM_BODY ? (size_t)o.n_bodies : a.dom == DOM_COLL ? (size_t)o.n_colliders : a.dom == DOM_PAIR ? (size_t)o.pool_cap : a.dom == DOM_JOINT ? (size_t)o.n_joints : a.stride;
        n = std::min(n, std::min(a.stride, b->stride));
        if (n == 0) continue;
        for (int p = 0; p < a.planes; ++p)
            HIPCHK(w, hipMemcpyAsync((char *)a.ptr + (size_t)p * a.stride * a.per * a.elem, (const char *)b->ptr + (size_t)p * b->stride * b->per * b->elem,
                                     n * a.per * a.elem, hipMemcpyDeviceToDevice, w->stream));
    }
    // b_sprev_t: xyz = the sleep reference translation (device state), w = max_extent of the attached shapes (host-authoritative)
    if (o.n_bodies > 0) HIPCHK(w, hipMemcpy2DAsync(d.b_sprev_t, sizeof(float4), o.b_sprev_t, sizeof(float4), 3 * sizeof(float), (size_t)o.n_bodies, hipMemcpyDeviceToDevice, w->stream));
    HIPCHK(w, hipMemcpyAsync(d.flags, o.flags, FL_COUNT * sizeof(int), hipMemcpyDeviceToDevice, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    memcpy(w->pinned_flags, w->old_pinned, FL_COUNT * sizeof(int));
    rp_launch_bp_rehash(d, w->stream); // the live pairs enter the (larger) current-epoch table
    int one = 1;
    for (int f : {FL_BP_DIRTY, FL_LAYOUT_DIRTY, FL_JOINT_DIRTY, FL_FLOW_DIRTY}) HIPCHK(w, hipMemcpyAsync(d.flags + f, &one, sizeof(int), hipMemcpyHostToDevice, w->stream));
    { int zero = 0; for (int f : {FL_BP_GRID_OK, FL_BP_NCHG, FL_BP_NFREED, FL_BP_TOMBS, FL_BP_FORCE_FULL}) HIPCHK(w, hipMemcpyAsync(d.flags + f, &zero, sizeof(int), hipMemcpyHostToDevice, w->stream)); } // the grid and the change lists are scratch of the old world
    HIPCHK(w, hipStreamSynchronize(w->stream));
    w->pinned_flags[FL_LAYOUT_DIRTY] = 1; // next steps stay on the full graph until the device reports a clean state
    w->full_until = w->steps_requested + 3;
    for (const AllocRec &a : w->old_allocs) hipFree(a.ptr);
    w->old_allocs.clear();
    hipHostFree(w->old_pinned); w->old_pinned = nullptr;
    w->carry = false;
    return RP_OK;
}

// the shard guard (rp_world_set_shard_guard) of the current device world: boxes + the coarse grid over them
static int upload_shard_guard(rp_world *w) {
    DevWorld &d = w->dw;
    d.sg_bmin = nullptr; d.sg_bmax = nullptr; d.sg_cell_start = nullptr; d.sg_cell_items = nullptr;
    if (w->guard_min.empty()) return RP_OK;
    DA(d.sg_bmin, w->guard_min.size()); DA(d.sg_bmax, w->guard_max.size()); DA(d.sg_cell_start, w->guard_start.size()); DA(d.sg_cell_items, std::max<size_t>(w->guard_items.size(), 1));
    UP(d.sg_bmin, w->guard_min); UP(d.sg_bmax, w->guard_max); UP(d.sg_cell_start, w->guard_start); UP(d.sg_cell_items, w->guard_items);
    for (int k = 0; k < 3; ++k) { d.sg_origin[k] = w->guard_origin[k]; d.sg_dims[k] = w->guard_dims[k]; }
    d.sg_inv_cell = 1.0f / w->guard_cell;
    d.sg_horizon = w->guard_horizon;
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return RP_OK;
}

// Upload the host mirrors into the SoA device world (the "upload = resume" path of SURVEY §5).
static int finalize(rp_world *w) {
    HIPCHK(w, hipSetDevice(w->device));
    { int r = check_sleep_scope(w); if (r != RP_OK) return r; }
    DevWorld &d = w->dw;
    memset(&d, 0, sizeof(d));
    d.sleep_enabled = world_sleep_enabled(w) ? 1 : 0;
    d.has_kinematic_pos = world_has_kinematic_pos(w) ? 1 : 0;
    d.has_force_events = world_has_force_events(w) ? 1 : 0;
    d.has_sensors = world_has_sensors(w) ? 1 : 0;
    d.has_convex = world_has_convex(w) ? 1 : 0;
    if (!w->polys.empty()) { int r = upload_polyhedra(w); if (r != RP_OK) return r; } // (after the memset above: the cv_* pointers)
    if (!w->comps.empty()) { int r = upload_composites(w); if (r != RP_OK) return r; }
    d.has_composite = 0; for (size_t i = 0; i < w->colliders.size(); ++i) if (!w->collider_removed[i] && shape_composite(w->colliders[i].shape)) d.has_composite = 1;
    d.gbar_blocks = gbar_grid_for_device(w->device);
    { const char *ni = getenv("RP_NO_BP_INCR"); d.bp_incremental = (ni && ni[0] == '1') ? 0 : 1; }
    { const char *dv = getenv("RP_BP_INCR_DIV"); d.bp_incr_div = dv ? std::max(1, atoi(dv)) : 1; }
    { const char *ab = getenv("RP_BP_ALWAYS_BUILD"); d.bp_always_build = (ab && ab[0] == '1') ? 1 : 0; }
    { const char *nt = getenv("RP_NO_TINY_ROUTING"); d.isl_route_tiny = (nt && nt[0] == '1') ? 0 : 1; }
    { const char *tn = getenv("RP_ISL_TINY_NC"); d.isl_tiny_nc = tn ? std::max(0, atoi(tn)) : 8; }
    { const char *im = getenv("RP_ISL_MANY"); d.isl_many = im ? std::max(1, atoi(im)) : (w->fused_grid > 0 ? w->fused_grid : 240); } // (more candidates than ONE resident pass of k_island_solve: round 5, a batch of 64 capsule worlds — 320 islands of 4 manifolds — 414 -> 230 us per step; round 4 waited for 960)
    { const char *ig = getenv("RP_ISL_GENERIC"); d.isl_generic = (ig && ig[0] == '1') ? 1 : 0; }
    w->compound = world_has_compound_bodies(w); refresh_ccd_facts(w);
    int nb = (int)w->bodies.size(), nc = (int)w->colliders.size();
    d.n_bodies = nb; d.n_colliders = nc;
    // capacities leave room for bodies / colliders inserted later without rebuilding the device world
    // (RP_SPARE_ROWS: test hook — a tiny spare makes live worlds outgrow their arrays, i.e. exercises the carry-over path)
    const char *env_spare = getenv("RP_SPARE_ROWS");
    const int spare = env_spare ? std::max(0, atoi(env_spare)) : 256, quarter = env_spare ? 0 : 1;
    const int capb = nb + quarter * (nb / 4) + spare, capc = nc + quarter * (nc / 4) + spare;
    w->cap_bodies = capb; w->cap_colliders = capc;
    const char *env_pool = getenv("RP_PAIRS_PER_COLLIDER");
    int ppc = env_pool ? atoi(env_pool) : 8;
    d.pool_cap = (int)std::min<long long>((long long)ppc * w->pairs_scale * capc + 1024, 1ll << 28);
    d.hash_cap = next_pow2(4LL * d.pool_cap);
    d.grid_cap = std::min(next_pow2(8LL * std::max(capc, 1)), 1 << 23); // (a million colliders fill a million cells: with fewer buckets than cells the 32-slot buckets of colliding cells overflow)
    d.grid_cap = std::max(d.grid_cap, 1024);
    d.large_cap = std::max(capc, 1); // (the brute-force list can hold every collider: a world of wildly mixed sizes gets slow, it does not fail)
    d.cons_cap = d.pool_cap;
    // grid cell size: 90th percentile of the collider bounding extents (+ fat margins)
    float pred = w->params.normalized_prediction_distance * w->params.length_unit;
    float margin = 2.0f * (pred * 0.5f + 4.0e-2f * w->params.length_unit);
    std::vector<float> ext;
    for (int ci = 0; ci < (int)w->colliders.size(); ++ci) {
        const rp_collider_desc &c = w->colliders[ci];
        if (c.shape == RP_SHAPE_HALFSPACE) continue; // unbounded: always on the broad phase's large list
        float r = shape_bounding_radius(w, ci);
        ext.push_back(2.0f * r + margin);
    }
    float cell = 1.0f;
    if (!ext.empty()) { std::sort(ext.begin(), ext.end()); cell = ext[(size_t)((ext.size() - 1) * 0.9)]; }
    if (!(cell > 1.0e-6f)) cell = 1.0f;
    fill_sim_params(w, d.prm, cell);

    DA(d.flags, FL_COUNT); DA(d.dbg, 1024); DA(d.bar, 16);
    DAC(d.b_pos, capb, DOM_BODY, 1, 1); DAC(d.b_rot, capb, DOM_BODY, 1, 1); DAC(d.b_linvel, capb, DOM_BODY, 1, 1); DAC(d.b_angvel, capb, DOM_BODY, 1, 1); DA(d.b_lcom_invm, capb); DA(d.b_invpi, capb);
    DA(d.b_pframe, capb); DAC(d.b_wcom, capb, DOM_BODY, 1, 1); DAC(d.b_eim, capb, DOM_BODY, 1, 1); DAC(d.b_eii0, capb, DOM_BODY, 1, 1); DAC(d.b_eii1, capb, DOM_BODY, 1, 1); DA(d.b_damp, capb); /* host-authoritative (damping, gravity scale, ccd_thickness): NOT carried over a growth — the carried copy of a row whose body got a collider in the same call held the old thickness (found by the growth fuzz, round 4) */
    DAC(d.b_uforce, capb, DOM_BODY, 1, 1); DAC(d.b_utorque, capb, DOM_BODY, 1, 1); DAC(d.b_flags, capb, DOM_BODY, 1, 1); DAC(d.b_quar, capb, DOM_BODY, 1, 1); DAF(d.b_collider, capb, 0xff); DAF(d.c_sibling, capc, 0xff);
    DA(d.b_ccd0_pos, capb); DA(d.b_ccd0_rot, capb); DA(d.ccd_list, capb); // continuous-collision pass: scratch of one step
    DAC(d.b_sleep, capb, DOM_BODY, 1, 1); DA(d.b_sprev_t, capb); DAC(d.b_sprev_r, capb, DOM_BODY, 1, 1); DAC(d.b_slabel, capb, DOM_BODY, 1, 1); DAC(d.b_slept_at, capb, DOM_BODY, 1, 1); DA(d.b_sleep_stamp, capb); DAC(d.b_wake_req, capb, DOM_BODY, 1, 1);
    DAC(d.lab_wake, capb, DOM_BODY, 1, 1); DAC(d.lab_awake, capb, DOM_BODY, 1, 1); DAC(d.b_next_pos, capb, DOM_BODY, 1, 1); DAC(d.b_next_rot, capb, DOM_BODY, 1, 1);
    // persistent islands (rp_sleep.hip): ids per body, the island table (index = island id < bodies), scratch of a maintenance pass
    DAFC(d.b_isl, capb, 0xff, DOM_BODY, 1, 1); DAC(d.pi_used, capb, DOM_BODY, 1, 1); DAC(d.pi_nb, capb, DOM_BODY, 1, 1); DAC(d.pi_dirty, capb, DOM_BODY, 1, 1); DAC(d.pi_denied, capb, DOM_BODY, 1, 1);
    DAC(d.pi_sleeping, capb, DOM_BODY, 1, 1); DAC(d.pi_free, capb, DOM_BODY, 1, 1); DA(d.pi_uf, capb); DA(d.pi_new, capb); DA(d.pi_best, capb); DA(d.pi_csize, capb); DA(d.pi_cisl, capb); DA(d.pi_list, capb);
    DAC(d.pi_w64, 4, DOM_FIXED, 1, 1); DAC(d.pi_stats, 16, DOM_FIXED, 1, 1);
    DA(d.sg_hit, capb); DA(d.lay_state, 16); DA(d.ov_owner, d.cons_cap);
    DAC(d.s_lin, capb, DOM_BODY, 1, 1); DAC(d.s_ang, capb, DOM_BODY, 1, 1); DAC(d.s_rot, capb, DOM_BODY, 1, 1); DAC(d.s_trans, capb, DOM_BODY, 1, 1); DAC(d.s_incl, capb, DOM_BODY, 1, 1); DAC(d.s_inca, capb, DOM_BODY, 1, 1);
    DAC(d.b_cmask, 4 * (size_t)capb, DOM_BODY, 1, 4); DAFC(d.b_min, capb, 0xff, DOM_BODY, 1, 1);
    DAC(d.c_parent, capc, DOM_COLL, 1, 1); DAC(d.c_sub, capc, DOM_COLL, 1, 1); DAC(d.c_ord, capc, DOM_COLL, 1, 1); DAC(d.c_shape, capc, DOM_COLL, 1, 1); DAC(d.c_lpos, capc, DOM_COLL, 1, 1); DAC(d.c_lrot, capc, DOM_COLL, 1, 1); DAC(d.c_pos, capc, DOM_COLL, 1, 1); DAC(d.c_rot, capc, DOM_COLL, 1, 1); DAC(d.c_he, capc, DOM_COLL, 1, 1);
    DAC(d.c_mat, capc, DOM_COLL, 1, 1); DAC(d.c_rules, capc, DOM_COLL, 1, 1); DAC(d.c_groups, capc, DOM_COLL, 1, 1); DAC(d.c_fatmin, capc, DOM_COLL, 1, 1); DAC(d.c_fatmax, capc, DOM_COLL, 1, 1); DAC(d.c_events, capc, DOM_COLL, 1, 1);
    d.ev_cap = std::max(65536, d.pool_cap); // a step raises at most one collision event and one force event per pair slot: a queue that is read every step cannot overflow
    DAC(d.ev_col, d.ev_cap, DOM_FIXED, 1, 1); DAC(d.ev_force_meta, d.ev_cap, DOM_FIXED, 1, 1); DAC(d.ev_force_a, d.ev_cap, DOM_FIXED, 1, 1); DAC(d.ev_force_b, d.ev_cap, DOM_FIXED, 1, 1);
    for (int k = 0; k < 2; ++k) { DA(d.bk_cnt[k], d.grid_cap); DA(d.bk_items[k], (size_t)d.grid_cap * RP_BP_BUCKET); } // the broad-phase grid: fixed-slot hash buckets, two copies (rp_broadphase.hip)
    DA(d.scan_block, 1024 + 8); // the scratch counters of a running broad-phase rebuild
    DA(d.large_list, d.large_cap);
    d.sub_cap = std::max(1024, 2 * w->n_sub);
    DA(d.large_sub_begin, (size_t)d.sub_cap + 2); DA(d.large_sub_cur, (size_t)d.sub_cap + 2); DA(d.large_tmp, d.large_cap);
    DA(d.c_fatold_min, capc); DA(d.c_fatold_max, capc);
    DA(d.c_chgstamp, capc); DA(d.c_stale, capc); DA(d.c_inlarge, capc); DA(d.c_rver, capc); DA(d.bp_chg_list, capc); DA(d.free_pending, d.pool_cap); // incremental broad phase (scratch: rebuilt by the next full pass)
    DAF(d.h_key[0], d.hash_cap, 0xff); DAF(d.h_key[1], d.hash_cap, 0xff); DA(d.h_slot[0], d.hash_cap); DA(d.h_slot[1], d.hash_cap);
    DAC(d.free_stack, d.pool_cap, DOM_PAIR, 1, 1);
    size_t P = (size_t)d.pool_cap;
    DAFC(d.p_c1, P, 0xff, DOM_PAIR, 1, 1); DAFC(d.p_c2, P, 0xff, DOM_PAIR, 1, 1); DAC(d.p_stamp, P, DOM_PAIR, 1, 1); DAC(d.p_color, P, DOM_PAIR, 1, 1); DAC(d.p_nsc, P, DOM_PAIR, 1, 1); DAC(d.p_npts, P, DOM_PAIR, 1, 1); DAC(d.p_pflags, P, DOM_PAIR, 1, 1); DAC(d.p_reldom, P, DOM_PAIR, 1, 1);
    DAFC(d.p_aux, P, 0xff, DOM_PAIR, 1, 1); DAFC(d.p_sub, P, 0xff, DOM_PAIR, 1, 1); // (composite pairs: no aux slot, no sub-shape yet = -1; the cluster count is set by bp_insert_pair)
    DAC(d.p_hint_seq, P, DOM_PAIR, 1, 1); DAC(d.p_colorb, P, DOM_PAIR, 1, 1); DAC(d.p_rb, P, DOM_PAIR, 1, 1); DAC(d.p_ln1, P, DOM_PAIR, 1, 1); DAC(d.p_ln2, P, DOM_PAIR, 1, 1); DAC(d.p_normal, P, DOM_PAIR, 1, 1); DAC(d.p_misc, P, DOM_PAIR, 1, 1);
    DAC(d.r_t, P, DOM_PAIR, 1, 1); DAC(d.r_r, P, DOM_PAIR, 1, 1); DAC(d.r_rot1, P, DOM_PAIR, 1, 1); DAC(d.r_rot2, P, DOM_PAIR, 1, 1);
    DAC(d.pt_lp1d, RP_MAX_PTS * P, DOM_PAIR, RP_MAX_PTS, 1); DAC(d.pt_lp2f, RP_MAX_PTS * P, DOM_PAIR, RP_MAX_PTS, 1); DAC(d.pt_imp, RP_MAX_PTS * P, DOM_PAIR, RP_MAX_PTS, 1); DAC(d.pt_wst, RP_MAX_PTS * P, DOM_PAIR, RP_MAX_PTS, 1);
    DAC(d.pt_dp1, RP_MAX_PTS * P, DOM_PAIR, RP_MAX_PTS, 1); DAC(d.pt_dp2, RP_MAX_PTS * P, DOM_PAIR, RP_MAX_PTS, 1);
    DAC(d.sc_a1, 4 * P, DOM_PAIR, 4, 1); DAC(d.sc_a2, 4 * P, DOM_PAIR, 4, 1);
    DAC(d.todo_slot, P, DOM_PAIR, 1, 1); DAC(d.todo_key, P, DOM_PAIR, 1, 1); DAC(d.todo_tmp, P, DOM_PAIR, 1, 1); DAC(d.np_list, P, DOM_PAIR, 1, 1);
    DA(d.grp_sub, RP_MAX_GROUPS); DA(d.grp_extra, RP_MAX_GROUPS); DA(d.b_extra, capb); DA(d.b_group, capb); DA(d.g_parent, capb); DA(d.g_key, capb); DA(d.k_group, d.cons_cap);
    DA(d.col_cnt, capb); DA(d.col_fill, capb); DA(d.col_begin, capb); DA(d.col_list, 2 * P); DA(d.col_sorted, 2 * P);
    DA(d.col_rec, P); DA(d.col_rank, P); DA(d.col_succ, P); DA(d.col_deps, P); DA(d.col_q, 2 * P);
    DA(d.color_count, RP_NUM_COLORS + 1); DA(d.color_begin, RP_NUM_COLORS + 1); DA(d.color_cursor, RP_NUM_COLORS + 1);
    DA(d.stage_color, RP_NUM_COLORS + 1); DA(d.stage_begin, RP_NUM_COLORS + 1); DA(d.stage_count, RP_NUM_COLORS + 1);
    DA(d.cons_pair, d.cons_cap); DAFC(d.p_conspos, P, 0xff, DOM_PAIR, 1, 1);
    DA(d.color_count_glob, RP_NUM_COLORS + 1); DA(d.color_rank, RP_NUM_COLORS + 1);
    d.cb_words = (capb + 31) / 32;
    DA(d.cb_bits, (size_t)128 * d.cb_words); DA(d.cb_prefix, (size_t)128 * d.cb_words);
    DAC(d.b_label, capb, DOM_BODY, 1, 1); DAFC(d.b_island, capb, 0xff, DOM_BODY, 1, 1); DAFC(d.b_local, capb, 0xff, DOM_BODY, 1, 1); DAC(d.r_nb, capb, DOM_BODY, 1, 1); DAC(d.r_nc, capb, DOM_BODY, 1, 1); DAFC(d.r_island, capb, 0xff, DOM_BODY, 1, 1);
    DAFC(d.p_island, P, 0xff, DOM_PAIR, 1, 1); DA(d.uf_pairs, P);
    DAC(d.isl_body_begin, capb, DOM_BODY, 1, 1); DAC(d.isl_nb, capb, DOM_BODY, 1, 1); DAC(d.isl_cons_begin, capb, DOM_BODY, 1, 1); DAC(d.isl_nc, capb, DOM_BODY, 1, 1); DAC(d.isl_fill_b, capb, DOM_BODY, 1, 1); DAC(d.isl_fill_c, capb, DOM_BODY, 1, 1);
    DAC(d.isl_bodies, capb, DOM_BODY, 1, 1); DAC(d.isl_cons, P, DOM_PAIR, 1, 1); DAC(d.isl_cstage, P, DOM_PAIR, 1, 1); DAC(d.isl_sorted, capb, DOM_BODY, 1, 1); DAC(d.isl_nstages, capb, DOM_BODY, 1, 1);
    DAC(d.isl_cg1, P, DOM_PAIR, 1, 1); DAC(d.isl_cg2, P, DOM_PAIR, 1, 1); DAC(d.isl_cl1, P, DOM_PAIR, 1, 1); DAC(d.isl_cl2, P, DOM_PAIR, 1, 1); DA(d.isl_inc_pos, 2 * P); DAC(d.r_ni, capb, DOM_BODY, 1, 1); DAC(d.isl_ni, capb, DOM_BODY, 1, 1); DAC(d.isl_icons_begin, capb, DOM_BODY, 1, 1); DAC(d.isl_fill_i, capb, DOM_BODY, 1, 1); DAC(d.isl_icons, P, DOM_PAIR, 1, 1); DAC(d.isl_inc_begin, capb, DOM_BODY, 1, 1); DAC(d.isl_inc_cnt, capb, DOM_BODY, 1, 1);
    // impulse joints: only joints with a dynamic side are active (select_active_interactions,
    // impulse_joint_set.rs:504-572), kept in edge order; frames go to solver-body space once
    // (GenericJoint::transform_to_solver_body_space, generic_joint.rs:624-636)
    std::vector<int> jb1, jb2, jlocked, jlimited, jmotor, jcolor, bnj(nb, 0);
    std::vector<float4> jlim[6], jmot[12];
    std::vector<float4> jf1t, jf1r, jf2t, jf2r;
    w->active_joint_ids.clear();
    for (size_t ji = 0; ji < w->joints.size(); ++ji) {
        // a world that is growing keeps the device index of every joint it already held: removed ones stay as tombstones
        const bool held = w->carry && std::binary_search(w->old_active_joint_ids.begin(), w->old_active_joint_ids.end(), (int)ji);
        const rp_joint_desc &j = w->joints[ji];
        const HostBody &rb1 = w->bodies[j.body1], &rb2 = w->bodies[j.body2];
        bool d1 = rb1.d.body_type != RP_BODY_FIXED && !rb1.removed, d2 = rb2.d.body_type != RP_BODY_FIXED && !rb2.removed; // is_dynamic_or_kinematic
        if (w->joint_removed[ji] || (!d1 && !d2)) {
            if (!held) continue;
            jb1.push_back(-1); jb2.push_back(-1); jf1t.push_back(mk4(0, 0, 0, 0)); jf1r.push_back(mk4(0, 0, 0, 1)); jf2t.push_back(mk4(0, 0, 0, 0)); jf2r.push_back(mk4(0, 0, 0, 1));
            jlocked.push_back(0); jlimited.push_back(0); jmotor.push_back(0); jcolor.push_back(RP_COLOR_UNCOLORED);
            for (int a = 0; a < 12; ++a) jmot[a].push_back(mk4(0, 0, 0, 0));
            for (int a = 0; a < 6; ++a) jlim[a].push_back(mk4(0, 0, 0, 0));
            w->active_joint_ids.push_back((int)ji);
            continue;
        }
        Pose f1 = joint_local_frame(j.local_anchor1, j.local_basis1), f2 = joint_local_frame(j.local_anchor2, j.local_basis2);
        if (!d1) f1 = pose_mul(host_body_pose(rb1), f1); else f1.t = f1.t - v3(rb1.lcom[0], rb1.lcom[1], rb1.lcom[2]);
        if (!d2) f2 = pose_mul(host_body_pose(rb2), f2); else f2.t = f2.t - v3(rb2.lcom[0], rb2.lcom[1], rb2.lcom[2]);
        jb1.push_back(d1 ? j.body1 : -1); jb2.push_back(d2 ? j.body2 : -1);
        jf1t.push_back(mk4(f1.t.x, f1.t.y, f1.t.z, 0)); jf1r.push_back(mk4(f1.r.x, f1.r.y, f1.r.z, f1.r.w));
        jf2t.push_back(mk4(f2.t.x, f2.t.y, f2.t.z, 0)); jf2r.push_back(mk4(f2.r.x, f2.r.y, f2.r.z, f2.r.w));
        jlocked.push_back((int)j.locked_axes); jlimited.push_back((int)(j.limit_axes & 0x3fu)); jmotor.push_back((int)(j.motor_axes & 0x3fu)); jcolor.push_back(RP_COLOR_UNCOLORED);
        for (int a = 0; a < 6; ++a) { const rp_joint_motor &m = j.motors[a]; jmot[2 * a].push_back(mk4(m.target_vel, m.target_pos, m.stiffness, m.damping)); jmot[2 * a + 1].push_back(mk4(m.max_force, (float)m.model, 0, 0)); }
        for (int a = 0; a < 3; ++a) jlim[a].push_back(mk4(j.limits[a][0], j.limits[a][1], 0, 0));
        for (int a = 0; a < 3; ++a) { // AngularLimitParams::new(min, max) — joint_constraint_helper.rs:44-72
            float mn = j.limits[3 + a][0], mx = j.limits[3 + a][1];
            float half_range = (mx - mn) * 0.5f;
            if (half_range >= 3.14159265358979323846f || half_range != half_range) jlim[3 + a].push_back(mk4(1.0f, 0.0f, 10.0f, 0));
            else { float center = (mn + mx) * 0.5f; jlim[3 + a].push_back(mk4(cosf(center * 0.5f), sinf(center * 0.5f), half_range, 0)); }
        }
        if (d1) bnj[j.body1]++;
        if (d2) bnj[j.body2]++;
        w->active_joint_ids.push_back((int)ji);
    }
    int nj = (int)jb1.size();
    d.n_joints = nj;
    { // joints that disable the contacts between their two bodies (GenericJoint::contacts_enabled = false)
        std::vector<unsigned long long> nck = no_contact_keys(w);
        d.n_nc = (int)nck.size();
        DA(d.nc_keys, w->joints.size() + 1);
        UP(d.nc_keys, nck);
        HIPCHK(w, hipStreamSynchronize(w->stream));
    }
    // the joint descriptors (bodies, CoM-space frames, axis masks, limits, motors) are host-authoritative: the fresh upload holds the
    // current local centres of mass and fixed-body poses; only the solver's own state (colours, impulses) is carried over
    DA(d.j_b1, nj); DA(d.j_b2, nj); DA(d.j_f1t, nj); DA(d.j_f1r, nj); DA(d.j_f2t, nj); DA(d.j_f2r, nj);
    DA(d.j_locked, nj); DA(d.j_limited, nj); DAC(d.j_color, nj, DOM_JOINT, 1, 1); DAC(d.j_tmp, nj, DOM_JOINT, 1, 1); DAC(d.j_order, nj, DOM_JOINT, 1, 1); DAC(d.j_imp, nj, DOM_JOINT, 1, 1); DAC(d.j_imp_ang, nj, DOM_JOINT, 1, 1);
    DA(d.j_lim, (size_t)6 * std::max(nj, 1)); DAC(d.j_imp_lim, nj, DOM_JOINT, 1, 1); DAC(d.j_imp_lim_ang, nj, DOM_JOINT, 1, 1);
    DA(d.j_motor, nj); DA(d.j_mot, (size_t)12 * std::max(nj, 1)); DAC(d.j_imp_mot, nj, DOM_JOINT, 1, 1); DAC(d.j_imp_mot_ang, nj, DOM_JOINT, 1, 1);
    d.pj_cap = next_pow2((long long)d.pool_cap + (long long)w->joints.size() + 16); // removal journal: every pair and joint at most once between two sleep passes
    DAC(d.pj_key, d.pj_cap, DOM_FIXED, 1, 1); DAC(d.pj_b, d.pj_cap, DOM_FIXED, 1, 1);
    DA(d.j_stage_begin, RP_NUM_COLORS + 1); DA(d.j_stage_count, RP_NUM_COLORS + 1); DA(d.j_group, std::max(nj, 1));
    DA(d.jc_first, std::max(nj, 1)); DA(d.jc_list, 2 * (size_t)std::max(nj, 1)); DA(d.jc_sorted, 2 * (size_t)std::max(nj, 1)); DA(d.jc_deps, std::max(nj, 1)); DA(d.jc_q, 2 * (size_t)std::max(nj, 1)); DA(d.jc_rank, std::max(nj, 1)); DA(d.jc_succ, std::max(nj, 1));
    DAC(d.bj_cmask, 4 * (size_t)capb, DOM_BODY, 1, 4); DAFC(d.bj_min, capb, 0xff, DOM_BODY, 1, 1); DA(d.b_njoints, capb);
    DAS(d.JR, (size_t)RP_JR_COUNT * std::max(nj, 1)); // im1, im2 + 12 rows x 6 planes (rp_joints.h); planes of unused rows are never touched
    UP(d.j_b1, jb1); UP(d.j_b2, jb2); UP(d.j_f1t, jf1t); UP(d.j_f1r, jf1r); UP(d.j_f2t, jf2t); UP(d.j_f2r, jf2r);
    d.joints_spherical = nj > 0 ? 1 : 0;
    for (int k = 0; k < nj; ++k) if (!(jlocked[k] == 0x7 && (jlimited[k] & ~jlocked[k]) == 0 && (jmotor[k] & ~jlocked[k]) == 0)) d.joints_spherical = 0; // (joint_update_one_t's test)
    UP(d.j_locked, jlocked); UP(d.j_limited, jlimited); UP(d.j_motor, jmotor); UP(d.j_color, jcolor); UP(d.b_njoints, bnj);
    for (int a = 0; a < 12; ++a) if (nj > 0 && hipMemcpyAsync(d.j_mot + (size_t)a * nj, jmot[a].data(), (size_t)nj * sizeof(float4), hipMemcpyHostToDevice, w->stream) != hipSuccess) { w->err = "upload failed"; return RP_ERR_DEVICE; }
    for (int a = 0; a < 6; ++a) if (nj > 0 && hipMemcpyAsync(d.j_lim + (size_t)a * nj, jlim[a].data(), (size_t)nj * sizeof(float4), hipMemcpyHostToDevice, w->stream) != hipSuccess) { w->err = "upload failed"; return RP_ERR_DEVICE; }
    {
        // b_collider: the LAST live collider of a dynamic body; c_sibling: the one before it (the chain the fused step's validators walk)
        std::vector<int> bcol(nb, -1), csib(nc, -1);
        for (int c = 0; c < nc; ++c) { int pb = w->collider_parent[c]; if (pb >= 0 && !w->collider_removed[c] && w->bodies[pb].d.body_type == RP_BODY_DYNAMIC && !w->bodies[pb].removed) { csib[c] = bcol[pb]; bcol[pb] = c; } }
        UP(d.b_collider, bcol); UP(d.c_sibling, csib);
        HIPCHK(w, hipStreamSynchronize(w->stream));
    }
    DAS(d.k_b1, d.cons_cap); DAS(d.k_b2, d.cons_cap); DAS(d.k_n, d.cons_cap); DAS(d.k_cid, d.cons_cap);
    // dataflow solver: toucher lists, rebuilt on the device whenever the layout changes (no carry-over needed)
    DA(d.f_rec, 2 * (size_t)capb); DA(d.fk_rank, d.cons_cap); DA(d.fj_rank, std::max(nj, 1)); DA(d.fb_deg, capb); DA(d.fb_begin, capb); DA(d.fb_fill, capb);
    DAS(d.f_adj, 2 * (size_t)d.cons_cap); DAS(d.f_jadj, 2 * (size_t)std::max(nj, 1)); DAS(d.f_sorted, 2 * (size_t)d.cons_cap); DAS(d.f_other, 2 * (size_t)d.cons_cap);
    if (w->params.friction_model != RP_FRICTION_COULOMB) DAS(d.ws_terms, (size_t)11 * 2 * d.cons_cap);
    // LDS tiles of the global path (rp_tiles.hip): contact-only worlds under the twist model that are large enough to leave the single
    // workgroup; RP_NO_TILES=1 keeps the per-stage launches, RP_TILE_TARGET=<n> sets the number of tiles aimed at (default: one per CU)
    {
        const char *nt = getenv("RP_NO_TILES"), *tt = getenv("RP_TILE_TARGET"), *tm = getenv("RP_TILE_MIN");
        d.tile_min = tm && atoi(tm) > 0 ? atoi(tm) : RP_TILE_MIN_BODIES;
        const bool eligible = !(nt && nt[0] == '1') && w->params.friction_model != RP_FRICTION_COULOMB && capb >= d.tile_min;
        d.tile_cap = eligible ? capb / 64 + 2 : 0;
        d.tile_target = tt && atoi(tt) > 0 ? atoi(tt) : 240;
        // constraint planes: Coulomb: + 9 tangent planes per point (rp_coulomb.h); worlds that may tile: + the shadow copy of the mutable planes
        DAS(d.C, (size_t)(w->params.friction_model == RP_FRICTION_COULOMB ? CQ_COUNT : CP_COUNT + (d.tile_cap ? CP_SHADOW_COUNT : 0)) * d.cons_cap);
        if (d.tile_cap) {
            DAS(d.t_lin, capb); DAS(d.t_ang, capb); DAS(d.t_rot, capb); DAS(d.t_trans, capb); DAS(d.fk_ids, d.cons_cap); DAS(d.tl_body_tile, capb); DAS(d.tl_owned, capb);
            DA(d.tl_hist, 2 * RP_TILE_CELLS); DAS(d.tl_cellofs, 2 * RP_TILE_CELLS); DA(d.tl_bbox, 16); DAS(d.tl_hdr, d.tile_cap);
            DAS(d.tl_cell, capb); DAS(d.tl_sorted, capb); DA(d.b_order, capb);
            if (nj > 0) { DAS(d.jm, (size_t)2 * 12 * nj); DAS(d.f_jsorted, 2 * (size_t)nj); DAS(d.f_jother, 2 * (size_t)nj); } // joint stages on tiles: the sweeps' mutable row words (two copies), sorted joint toucher lists
            // (Both uploads below ride the world's stream behind the zero fills of DA: the stream is non-blocking, so a synchronous
            // hipMemcpy — legacy stream — is NOT ordered against a fill that is still queued.  Round 3 used hipMemcpy here: a fill that
            // ran late wiped b_order to all zeros — every manifold of a colour then ranked to the same position — or tl_bbox's rest
            // state.  That was the intermittent gross mismatch of DESIGN.md section 4.10.)
            { std::vector<int> iota(capb); for (int i = 0; i < capb; ++i) iota[i] = i; UP(d.b_order, iota); HIPCHK(w, hipStreamSynchronize(w->stream)); }
            DAS(d.tl_soff, (size_t)d.tile_cap * (RP_TILE_STAGES + 1)); DAS(d.tl_bodies, (size_t)d.tile_cap * RP_TILE_BCAP); DAS(d.tl_cons, (size_t)d.tile_cap * RP_TILE_CCAP);
            const unsigned rest[16] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            HIPCHK(w, hipMemcpyAsync(d.tl_bbox, rest, sizeof(rest), hipMemcpyHostToDevice, w->stream));
            HIPCHK(w, hipStreamSynchronize(w->stream));
#ifdef RP_TESTING
            // test hook (tools/tile_race_stress.py --replay): what the round-3 race left behind when the fill lost it — 1: b_order wiped, 2: tl_bbox wiped
            if (const char *lf = getenv("RP_TEST_LATE_FILL")) {
                if (atoi(lf) & 1) HIPCHK(w, hipMemsetAsync(d.b_order, 0, (size_t)capb * sizeof(int), w->stream));
                if (atoi(lf) & 2) HIPCHK(w, hipMemsetAsync(d.tl_bbox, 0, sizeof(rest), w->stream));
            }
            // test hook (RP_TILE_STALE_PLAN=1): the device never finds a tiling worth having while the host plans tile sweeps all the same —
            // every sweep then takes the branch that normally only a stale hint reaches (k_tile_sweep with FL_N_TILES == 0)
            if (getenv("RP_TILE_STALE_PLAN")) d.tile_min = 0x7fffffff;
#endif
        }
    }

    { int r = upload_shard_guard(w); if (r != RP_OK) return r; } // the shard guard follows the device world
    // host SoA staging (one batched copy per attribute)
    {
        std::vector<float4> pos(nb), rot(nb), lv(nb), av(nb), lci(nb), ipi(nb), pfr(nb), damp(nb);
        std::vector<float4> slp(nb), spt(nb), spr(nb), npos(nb), nrot(nb);
        std::vector<int> bfl(nb), slab(nb);
        for (int i = 0; i < nb; ++i) {
            BodyRow r = pack_body(w->bodies[i]);
            pos[i] = r.pos; rot[i] = r.rot; lv[i] = r.lv; av[i] = r.av; lci[i] = r.lci; ipi[i] = r.ipi; pfr[i] = r.pfr; damp[i] = r.damp; bfl[i] = r.fl;
            slp[i] = r.slp; spt[i] = r.spt; spr[i] = r.spr; slab[i] = r.slabel; npos[i] = r.npos; nrot[i] = r.nrot;
        }
        UP(d.b_next_pos, npos); UP(d.b_next_rot, nrot);
        UP(d.b_sleep, slp); UP(d.b_sprev_t, spt); UP(d.b_sprev_r, spr); UP(d.b_slabel, slab);
        UP(d.b_pos, pos); UP(d.b_rot, rot); UP(d.b_linvel, lv); UP(d.b_angvel, av); UP(d.b_lcom_invm, lci); UP(d.b_invpi, ipi);
        UP(d.b_pframe, pfr); UP(d.b_damp, damp); UP(d.b_flags, bfl);
        std::vector<int> bext(nb); for (int i = 0; i < nb; ++i) bext[i] = w->bodies[i].d.additional_solver_iterations;
        UP(d.b_extra, bext);
        std::vector<int> cpar(nc), csh(nc), cord(nc);
        std::vector<float4> clp(nc), clr(nc), che(nc), cmat(nc), fmn(nc), fmx(nc);
        std::vector<int2> crul(nc); std::vector<uint2> cgrp(nc); std::vector<float2> cev(nc);
        for (int i = 0; i < nc; ++i) {
            ColliderRow r = pack_collider(w, i);
            cpar[i] = r.parent; cord[i] = r.ord; csh[i] = r.shape; clp[i] = r.lp; clr[i] = r.lr; che[i] = r.he; cmat[i] = r.mat; crul[i] = r.rules; cgrp[i] = r.groups; fmn[i] = r.fmn; fmx[i] = r.fmx; cev[i] = r.events;
        }
        UP(d.c_sub, w->collider_sub); d.n_sub = w->n_sub;
        UP(d.c_parent, cpar); UP(d.c_ord, cord); UP(d.c_shape, csh); UP(d.c_lpos, clp); UP(d.c_lrot, clr); UP(d.c_he, che); UP(d.c_mat, cmat);
        UP(d.c_rules, crul); UP(d.c_groups, cgrp); UP(d.c_fatmin, fmn); UP(d.c_fatmax, fmx); UP(d.c_events, cev);
        HIPCHK(w, hipStreamSynchronize(w->stream)); // the staging vectors die here
    }
    { int r = upload_group_table(w); if (r != RP_OK) return r; }
    std::vector<int> fl(FL_COUNT, 0);
    fl[FL_BP_DIRTY] = 1; fl[FL_LAYOUT_DIRTY] = 1; fl[FL_JOINT_DIRTY] = 1; fl[FL_FLOW_DIRTY] = 1;
    UP(d.flags, fl);
    HIPCHK(w, locked_hipHostMalloc((void **)&w->pinned_flags, FL_COUNT * sizeof(int), hipHostMallocMapped));
    memset(w->pinned_flags, 0, FL_COUNT * sizeof(int));
    HIPCHK(w, hipHostGetDevicePointer((void **)&d.host_flags, w->pinned_flags, 0));
    const bool had_islands = w->carry && w->old_dw.sleep_enabled;
    const int old_nb = w->carry ? w->old_dw.n_bodies : 0, old_nj = w->carry ? w->old_dw.n_joints : 0;
    if (w->carry) { int r = carry_over(w); if (r != RP_OK) return r; } // the rows of the previous device world move in; step counters keep running
    else { w->steps_requested = 0; w->seq_enqueued = 0; w->full_until = 0; }
    if (d.sleep_enabled) {
        // persistent islands: a world that grew keeps its table (carried rows) and gives the new bodies singleton islands, its new
        // joints are linked by the next sleep pass; a rebuilt world takes the saved table; everything else starts from singletons
        if (had_islands) {
            rp_launch_pi_ensure(d, w->stream, old_nb, nb - old_nb, 0);
            if (nj > old_nj) { int cur = 0; HIPCHK(w, hipMemcpy(&cur, d.flags + FL_PI_JLINK, sizeof(int), hipMemcpyDeviceToHost)); if (cur == 0) { cur = old_nj + 1; HIPCHK(w, hipMemcpy(d.flags + FL_PI_JLINK, &cur, sizeof(int), hipMemcpyHostToDevice)); } }
        } else if (!w->carry && w->pi_saved.valid) {
            auto &ps = w->pi_saved;
            std::vector<int> isl(nb); for (int i = 0; i < nb; ++i) isl[i] = w->bodies[i].isl;
            UP(d.b_isl, isl); UP(d.pi_used, ps.used); UP(d.pi_nb, ps.nb); UP(d.pi_dirty, ps.dirty); UP(d.pi_denied, ps.denied); UP(d.pi_sleeping, ps.sleeping); UP(d.pi_free, ps.freel); UP(d.pi_stats, ps.stats);
            HIPCHK(w, hipMemcpyAsync(d.pi_w64, ps.w64, sizeof(ps.w64), hipMemcpyHostToDevice, w->stream));
            int v[3] = {ps.next, ps.nfree, ps.pending};
            HIPCHK(w, hipMemcpyAsync(d.flags + FL_PI_NEXT, &v[0], sizeof(int), hipMemcpyHostToDevice, w->stream));
            HIPCHK(w, hipMemcpyAsync(d.flags + FL_PI_NFREE, &v[1], sizeof(int), hipMemcpyHostToDevice, w->stream));
            HIPCHK(w, hipMemcpyAsync(d.flags + FL_PI_PENDING, &v[2], sizeof(int), hipMemcpyHostToDevice, w->stream));
            HIPCHK(w, hipStreamSynchronize(w->stream));
        } else rp_launch_pi_ensure(d, w->stream, 0, nb, 1); // bootstrap (persistent.rs:600-625): singletons; joints and touching pairs link in the first sleep pass
    }
    w->pi_saved.valid = false;
    rp_launch_init_bodies(d, w->stream);
    rp_launch_collider_update(d, w->stream);
    HIPCHK(w, hipStreamSynchronize(w->stream));
    HIPCHK(w, hipGetLastError());
    w->finalized = true; w->hints_valid = false;
    if (d.sleep_enabled) for (int b : w->pending_wake) if (b >= 0 && b < nb) { int r = queue_wake(w, b, 2); if (r != RP_OK) return r; }
    w->pending_wake.clear();
    return RP_OK;
}

static void enqueue_collision(rp_world *w) {
    if (w->cur_fast && w->plan_fused) { // the fused k_island_solve validates the step itself (fat AABBs, recycle tests, sleep observation) ...
        rp_launch_sensor_check(w->dw, w->stream); // ... except that a sensor's intersection may start or stop (worlds with sensors only)
        return;
    }
    if (w->cur_fast) {
        rp_launch_fast_front(w->dw, w->stream, w->plan_no_global);
        rp_launch_sleep_fast(w->dw, w->stream); // sleep-enabled worlds: the per-step observation + "would an island fall asleep?" (then: abort)
        rp_launch_sensor_fast(w->dw, w->stream); // worlds with sensors: their pairs' intersection tests (after the last kernel that can abort)
        return;
    }
    if (w->cur_lean) { // lean graph: collision detection only (the world has no sleeping, no sensors: lean_world_ok)
        rp_launch_collider_update(w->dw_lean, w->stream);
        rp_launch_broadphase(w->dw_lean, w->stream);
        rp_launch_narrowphase_part(w->dw_lean, w->stream, 2);
        return;
    }
    rp_launch_collider_update(w->dw, w->stream);
    rp_launch_broadphase(w->dw, w->stream);
    rp_launch_wake(w->dw, w->stream, 0); // user wake-ups and pair deletions take effect before the narrow phase reads the awake set
    rp_launch_narrowphase(w->dw, w->stream);
}
// build_islands_and_solve_velocity_constraints: LDS island megakernel + the global path
static void enqueue_island_solver(rp_world *w) {
    // SINGLE mode: workgroup 0 of this launch retires the step (FL_SEQ / FL_STEP, hint publication)
    const int fused = (w->cur_fast && w->plan_fused) ? 1 : 0;
    if (w->cur_lean && (w->dw_lean.lean & 2)) return; // a bare lean graph: no island exists (verified by lean_dead in every kernel of the graph)
    // every workgroup of the fused step must be resident at once: the grid is capped by what the device can hold (rp_fused_grid)
    // the register-lean form of the kernel (rp_islands_lean.h: one 640-thread workgroup = TWO islands per CU) is the planner's choice
    // when the world has more islands than one pass of the classic form holds (plan_dense); a workgroup takes two islands per round
    const int dense = w->plan_dense;
    const int cap = dense ? w->fused_grid_dense : w->fused_grid;
    const int want = dense ? (w->plan_island_grid + 1) / 2 : w->plan_island_grid;
    rp_launch_island_solve(w->cur_lean ? w->dw_lean : w->dw, w->stream, fused ? std::min(want, cap) : want,
                           w->has_restitution ? 1 : 0, w->cur_fast, w->plan_single, fused, dense, w->plan_wide);
}
// MULTI mode of the global path, measured on MI355X: contact-only worlds under the twist model are fastest
// with one launch per colour stage + the body-centric warm start (b3d_large_pyramid 0.81 ms against 0.94 ms); worlds with impulse
// joints (b3d_joint_grid 0.37 against 0.45 ms) and the Coulomb model (no body-centric warm start) with the dataflow launch.
static bool flow_now(const rp_world *w) {
    if (!w->use_flow) return false;
    // with no internal PGS iteration nothing on a body's hand-off chain separates the joint-row update of a substep from that substep's
    // integrate, so the dataflow launch could rebuild joint rows from poses one substep ahead: such worlds take the per-stage launches
    if (w->dw.n_joints > 0 && w->params.num_internal_pgs_iterations == 0) return false;
    return w->force_flow || w->dw.n_joints > 0 || w->params.friction_model == RP_FRICTION_COULOMB;
}
static void enqueue_global_solver(rp_world *w) {
    int hr = w->has_restitution ? 1 : 0;
    if (w->cur_fast && w->plan_no_global) return; // k_fast_front verified on the device that the global path is empty
    if (w->plan_single) rp_launch_global_single(w->dw, w->stream, hr, w->cur_fast);
    else if (w->plan_tile_grid == 0 && flow_now(w)) rp_launch_global_flow(w->dw, w->stream, w->flow_grid, hr); // one dataflow launch (rp_flow.hip)
    else {
        const DevWorld &dw = w->cur_lean ? w->dw_lean : w->dw;
        rp_launch_solver_assembly(dw, w->stream, w->cur_lean);
        const int parity = rp_launch_solver_loop(dw, w->stream, w->plan_stages, w->plan_blocks, hr, w->plan_joint_stages, w->plan_tile_grid, w->plan_no_contacts);
        rp_launch_solver_writeback(dw, w->stream, parity, (!w->cur_fast && rp_ccd_launches(w->dw)) ? 0 : 1); // (a full step's k_ccd publishes the hints)
    }
}
static void enqueue_solver(rp_world *w) { enqueue_island_solver(w); enqueue_global_solver(w); }
static void enqueue_finish(rp_world *w) {
    // the scalars reach the mapped hint buffer from the device: k_island_solve (SINGLE) / k_publish (MULTI)
    if (!w->cur_fast) rp_launch_ccd(w->cur_lean ? w->dw_lean : w->dw, w->stream, w->has_bullets ? 1 : 0, (!w->plan_single && !(w->plan_tile_grid == 0 && flow_now(w))) ? 1 : 0); // (publishes for the per-stage / tile path) // run_ccd_motion_clamping (substep.rs:496-519): fast bodies are swept on full steps
    rp_launch_force_events(w->dw, w->stream, w->cur_fast); // contact force events of the step that just retired
}

static int pow2_ceil(int x) { int b = 1; while (b < x) b <<= 1; return b; }
static void plan_from_hints(rp_world *w, const int *fl) {
    // global path: one workgroup is enough while it holds little work, else one launch per colour stage
    w->plan_single = (fl[FL_N_CONS] <= 1024 && fl[FL_N_GLOB_BODIES] <= 4096 && w->dw.n_joints <= 1024) ? 1 : 0;
    const char *force = getenv("RP_FORCE_MULTI");
    if (force && force[0] == '1') w->plan_single = 0;
    if (w->dw.n_groups > 1) w->plan_single = 1; // substep solve-groups: the one-workgroup group solver (rp_groups.h)
    w->plan_stages = fl[FL_N_PARALLEL];
    w->plan_joint_stages = fl[FL_NJ_STAGES];
    if (flow_now(w)) { w->plan_stages = 0; w->plan_joint_stages = 0; } // the dataflow launch does not depend on the stage layout
    w->plan_no_global = (fl[FL_N_CONS] == 0 && fl[FL_N_GLOB_BODIES] == 0 && w->dw.n_joints == 0) ? 1 : 0;
    // round up to a power of two so small changes of the stage size do not force a re-capture
    w->plan_blocks = flow_now(w) ? 1 : std::min(std::max(pow2_ceil((fl[FL_MAX_STAGE] + 255) / 256), 1), 4096);
    w->plan_island_grid = std::min(std::max(pow2_ceil(fl[FL_N_ISLANDS]), 1), 8192);
    // more islands than one resident pass of k_island_solve holds: the lean form puts two on a CU.  Cost model from the island-count sweep
    // on MI355X (profiles/r05_island_count_sweep.txt): a pass of the classic form (<= fused_grid islands) takes ~71 us, a pass of the
    // lean form (<= 2 x fused_grid_dense islands) ~118 us — the lean form wins when it saves enough passes (361 islands: 118 against
    // 141 us; 484: 235 against 209, so the classic form keeps those; 2,916: 830 against 936).
    {
        const int n_isl = fl[FL_N_ISLANDS];
        bool lean_wins = false;
        // (the model is the pyramids' — 145 manifolds per island; islands of a few bodies finish a pass of the classic form long before
        // 71 us and gain nothing from sharing a CU: a batch of 64 capsule worlds, 320 islands of 4 manifolds, took 136 us per lean launch)
        const bool sizable = (long long)fl[FL_N_CONS_ALL] - fl[FL_N_CONS] >= 48ll * n_isl;
        if (w->fused_grid_dense > 0 && w->fused_grid > 0 && n_isl > w->fused_grid && sizable) {
            const long long classic = (long long)((n_isl + w->fused_grid - 1) / w->fused_grid) * 71;
            const long long lean = (long long)((n_isl + 2 * w->fused_grid_dense - 1) / (2 * w->fused_grid_dense)) * 118;
            lean_wins = lean < classic;
        }
        w->plan_dense = (w->fused_grid_dense > 0 && (w->force_dense || (w->auto_dense && lean_wins))) ? 1 : 0;
    }
    // LDS tiles (rp_tiles.hip): once the device has published a valid tiling of the global component, a sweep is one launch over the
    // tiles (grid rounded up to 16 so small changes of the tile count do not force a re-capture; the kernel loops over tiles beyond it)
    w->plan_no_contacts = (fl[FL_N_CONS] == 0 && w->dw.tile_cap > 0) ? 1 : 0; // (tile sweeps: the increment folds into the sweep while no manifold exists)
    // a world without a single manifold and without an LDS island (b3d_joint_grid): its LEAN graphs also leave out the launches that
    // only contacts and islands give work to — k_island_solve and the four k_ws_prepare of a step — and validate that on the device
    // too (DevWorld::lean bit 1, lean_dead): ~25 us of launches that found nothing to do in a 0.27 ms step
    w->plan_bare = (fl[FL_N_CONS] == 0 && fl[FL_N_ISLANDS] == 0 && fl[FL_N_CONS_ALL] == 0 && w->dw.tile_cap > 0) ? 1 : 0;
    // (a valid tiling also replaces the dataflow launch of jointed worlds; forced flow — RP_FLOW=1 — keeps it)
    w->plan_tile_grid = (w->dw.tile_cap > 0 && !w->plan_single && !w->force_flow && fl[FL_N_TILES] > 0) ? ((fl[FL_N_TILES] + 15) / 16) * 16 : 0;
    if (w->dw.tile_cap > 0 && w->dw.tile_min == 0x7fffffff && !w->plan_single && !w->force_flow) w->plan_tile_grid = 16; // (RP_TILE_STALE_PLAN: see finalize)
    if (w->plan_tile_grid > 0) { w->plan_stages = fl[FL_N_PARALLEL]; w->plan_joint_stages = fl[FL_NJ_STAGES]; w->plan_blocks = std::min(std::max(pow2_ceil((fl[FL_MAX_STAGE] + 255) / 256), 1), 4096); } // (the per-stage launches of the restitution sweep)
    // fused single-kernel fast step: every workgroup must be resident at once (in-launch arrival barrier)
    // (a grid of at most fused_grid workgroups; workgroups loop over islands beyond that)
    // Round 5: compound bodies (the validators walk every collider of a body), sleep-enabled worlds (the validators run the sleep
    // observation of their islands' bodies and abort when an island could fall asleep), worlds with sensors (k_sensor_check in front)
    // and contact-force events (k_force_events behind, as in every graph) keep it.
    // (the worlds whose islands run on k_island_generic — FrictionModel::Coulomb — do not: that kernel has no fused form)
    // (a body thinner than ~3.5 x the fat-AABB margin could move half its thickness — the CCD criterion — without leaving its fat AABB,
    // i.e. without the fused step noticing: such worlds take the fast graph, whose front kernel predicts the criterion)
    const bool ccd_safe = w->params.max_ccd_substeps == 0 || w->min_ccd_thickness >= 0.14f * w->params.length_unit;
    static const bool narrow_fused = getenv("RP_FUSED_NARROW") && getenv("RP_FUSED_NARROW")[0] == '1'; // A/B: round 4's rule (no compound bodies, sleeping, sensors, force events)
    w->plan_wide = (w->compound || w->dw.sleep_enabled) ? 1 : 0; // the island kernel with the WIDE validators (every collider of a body, sleep observation)
    const bool cliffs = w->compound || w->dw.has_force_events || w->dw.sleep_enabled || w->dw.has_sensors;
    w->plan_fused = (ccd_safe && w->use_fused && w->fused_grid > 0 && !(narrow_fused && cliffs) && !(w->dw.has_sensors && w->dw.has_composite) && w->params.friction_model != RP_FRICTION_COULOMB && !w->dw.isl_generic && w->plan_no_global && w->plan_single && fl[FL_N_ISLANDS] > 0) ? 1 : 0;
}

static int capture(rp_world *w, hipGraph_t *g, hipGraphExec_t *ge, void (*fn)(rp_world *)) {
    // Relaxed: another host thread stepping another world on this device may issue synchronous HIP calls (hipMemcpy in settle())
    // while this thread captures; only kernel launches on this world's own stream happen between Begin and End
    {   // the lock covers the capture only (fn launches kernels on this world's stream); instantiation (~10 ms) runs outside it, so a
        // re-capture in one world no longer stalls every other host thread's copies
        std::lock_guard<std::mutex> guard(g_hip_unsafe_api);
        HIPCHK(w, hipStreamBeginCapture(w->stream, hipStreamCaptureModeRelaxed));
        fn(w);
        HIPCHK(w, hipStreamEndCapture(w->stream, g));
    }
    HIPCHK(w, hipGraphInstantiate(ge, *g, nullptr, nullptr, 0));
    return RP_OK;
}
static void enqueue_global_and_finish(rp_world *w) { enqueue_global_solver(w); enqueue_finish(w); }
static void enqueue_whole(rp_world *w) { enqueue_collision(w); enqueue_solver(w); enqueue_finish(w); }

static int check_overflow(rp_world *w, const int *fl) {
    if (fl[FL_OVERFLOW] & RP_OVF_SHARD) {
        w->err = "shard guard: a body of this shard moved into a cell that holds another shard's bodies (the shards are no longer independent)";
        return RP_ERR_INVALID;
    }
    if (fl[FL_OVERFLOW]) {
        char buf[256];
        snprintf(buf, sizeof(buf), "device error (flags 0x%x: 1=pair pool 2=pair hash 4=grid cells 8=large list 16=constraints: raise RP_PAIRS_PER_COLLIDER; "
                 "32=a rebuild kernel's grid barrier timed out: the GPU is shared 64=dataflow solver stalled: the GPU is shared, set RP_NO_FLOW=1)", fl[FL_OVERFLOW]);
        w->err = buf;
        return RP_ERR_CAPACITY;
    }
    return RP_OK;
}

// Enqueue one step graph.  `fast`: 1 selects the steady-state graph (see the file header), 2 the lean graph (rp_world.h "lean step
// graphs": a full step without the launches that rebuild colouring / layout / toucher ranks / tiling, self-validating on the device).
static int launch_step(rp_world *w, int fast) {
    w->cur_lean = fast == 2 ? 1 : 0;
    if (w->cur_lean) { w->dw_lean = w->dw; w->dw_lean.lean = 1 | (w->plan_bare ? 2 : 0); }
    w->cur_fast = fast == 1 ? 1 : 0;
    w->seq_enqueued++;
    if (fast == 1) { w->fast_steps++; if (w->plan_fused) w->fused_steps++; } else if (fast == 2) w->lean_steps++; else w->full_steps++;
    if (w->timers) {
        // three sub-graphs with events in between (Counters from hipEvents)
        if (!w->timed_ready[fast]) {
            int r;
            if (fast && (!w->plan_fused || w->dw.has_sensors) && (r = capture(w, &w->g_col[fast], &w->ge_col[fast], enqueue_collision)) != RP_OK) return r; // (fused: only k_sensor_check lives there)
            w->timed_ready[fast] = true;
            if (!(fast && w->plan_no_global && !w->dw.has_force_events) && (r = capture(w, &w->g_fin[fast], &w->ge_fin[fast], enqueue_global_and_finish)) != RP_OK) return r;
        }
        HIPCHK(w, hipEventRecord(w->ev[0], w->stream));
        if (!fast) { // full step: the collision stage is launched piecewise so CollisionDetectionCounters / island_construction_time get their own events
            rp_launch_collider_update(w->dw, w->stream);
            rp_launch_broadphase(w->dw, w->stream);
            rp_launch_wake(w->dw, w->stream, 0);
            HIPCHK(w, hipEventRecord(w->ev[4], w->stream));
            rp_launch_narrowphase_part(w->dw, w->stream, 0);
            HIPCHK(w, hipEventRecord(w->ev[5], w->stream));
            rp_launch_narrowphase_part(w->dw, w->stream, 1);
        } else if (w->ge_col[fast]) HIPCHK(w, hipGraphLaunch(w->ge_col[fast], w->stream));
        HIPCHK(w, hipEventRecord(w->ev[1], w->stream));
        enqueue_island_solver(w); // launched directly so the two events bracket the kernel alone (no graph-launch gap)
        HIPCHK(w, hipEventRecord(w->ev[2], w->stream));
        if (w->ge_fin[fast]) HIPCHK(w, hipGraphLaunch(w->ge_fin[fast], w->stream));
        HIPCHK(w, hipEventRecord(w->ev[3], w->stream));
        HIPCHK(w, hipEventSynchronize(w->ev[3]));
        float a = 0, c = 0, d = 0;
        hipEventElapsedTime(&a, w->ev[0], w->ev[1]); hipEventElapsedTime(&c, w->ev[1], w->ev[2]); hipEventElapsedTime(&d, w->ev[2], w->ev[3]);
        if (!fast || !w->pinned_flags[FL_FAST_ABORT]) { // aborted fast steps did no work: keep them out of the averages
            // SINGLE mode: c = k_island_solve alone (the TGS loop of every LDS-resident island), d = the
            // global single-workgroup solve; MULTI mode: the per-colour launch sequence is in d.
            w->acc_col_ms += a; w->acc_isl_ms += c; w->acc_glob_ms += d; w->acc_step_ms += a + c + d; w->acc_steps++;
            if (!fast) {
                float bp = 0, np = 0, ic = 0;
                hipEventElapsedTime(&bp, w->ev[0], w->ev[4]); hipEventElapsedTime(&np, w->ev[4], w->ev[5]); hipEventElapsedTime(&ic, w->ev[5], w->ev[1]);
                w->acc_bp_ms += bp; w->acc_np_ms += np; w->acc_islc_ms += ic; w->acc_full_steps++;
            }
            w->loop_ms_since_read += c; w->loop_steps_since_read++;
        }
        return RP_OK;
    }
    // a fused fast step is ONE kernel: launched directly (a one-node graph replay costs more than the launch)
    static const bool fused_eager = getenv("RP_FUSED_GRAPH") == nullptr;
    if (!w->use_graph || w->steps_requested <= w->eager_until || (fast == 1 && w->plan_fused && fused_eager)) { enqueue_whole(w); HIPCHK(w, hipGetLastError()); return RP_OK; }
    static const bool dbg = getenv("RP_DEBUG") != nullptr;
    if (!w->ge_whole[fast]) {
        if (dbg) fprintf(stderr, "RPDBG capture fast=%d seq=%lld stages=%d blocks=%d single=%d grid=%d jst=%d\n", fast, w->seq_enqueued, w->plan_stages, w->plan_blocks, w->plan_single, w->plan_island_grid, w->plan_joint_stages);
        int r = capture(w, &w->g_whole[fast], &w->ge_whole[fast], enqueue_whole);
        if (r != RP_OK) return r;
        if (dbg) fprintf(stderr, "RPDBG captured\n");
    }
    if (dbg) fprintf(stderr, "RPDBG launch seq=%lld\n", w->seq_enqueued);
    HIPCHK(w, hipGraphLaunch(w->ge_whole[fast], w->stream));
    return RP_OK;
}

static int step_once(rp_world *w, bool allow_fast) {
    if (!w->hints_valid) {
        // First step after (re)building the world: run collision detection eagerly and read the
        // colour layout once so the solver launch plan is right from the start.
        w->cur_fast = 0; w->cur_lean = 0;
        int fl[FL_COUNT];
        for (int attempt = 0; ; ++attempt) {
            enqueue_collision(w);
            HIPCHK(w, hipMemcpyAsync(fl, w->dw.flags, sizeof(fl), hipMemcpyDeviceToHost, w->stream));
            HIPCHK(w, hipStreamSynchronize(w->stream));
            int r = check_overflow(w, fl);
            if (r == RP_ERR_CAPACITY && fl[FL_OVERFLOW] == RP_OVF_POOL && w->never_stepped && w->seq_enqueued == 0 && attempt < 10) {
                // the very first broad-phase pass of a freshly built world found more pairs than the pool holds (a dense pile: the pool
                // starts at RP_PAIRS_PER_COLLIDER = 8 slots per collider row).  No step has run: build the device world again with twice
                // the slots.  (A pool that fills up LATER grows ahead of time: rp_step.)
                // What the host entry points wrote into device rows only since the world was built (rp_bodies_write, add_force,
                // apply_impulse, wake_up, set_next_kinematic_position after a step(0) / an auto-finalize) comes along: the body rows
                // return to the host mirrors, user forces and wake requests are put back after the rebuild.  finalize() starts the
                // step counters afresh — but rp_step has already counted this step: they are put back too (a host one step behind the
                // device's FL_STEP would take the first aborted fast step for a retired one and never replay it).
                w->pairs_scale *= 2; w->err.clear();
                const long long req = w->steps_requested, full_until = w->full_until;
                const int nb0 = w->dw.n_bodies;
                std::vector<float4> uf(nb0), ut(nb0); std::vector<int> wr(nb0);
                {   // only the rows the user wrote: a row that still holds what finalize() uploaded keeps its host mirror untouched (taking
                    // it back would run the mirror's quaternion through pack_body's normalisation a second time: not the oracle's bits)
                    std::vector<float4> pos(nb0), rot(nb0), lv(nb0), av(nb0), npos(nb0), nrot(nb0);
                    if (nb0 > 0) {
                        HIPCHK(w, hipMemcpy(pos.data(), w->dw.b_pos, (size_t)nb0 * sizeof(float4), hipMemcpyDeviceToHost));
                        HIPCHK(w, hipMemcpy(rot.data(), w->dw.b_rot, (size_t)nb0 * sizeof(float4), hipMemcpyDeviceToHost));
                        HIPCHK(w, hipMemcpy(lv.data(), w->dw.b_linvel, (size_t)nb0 * sizeof(float4), hipMemcpyDeviceToHost));
                        HIPCHK(w, hipMemcpy(av.data(), w->dw.b_angvel, (size_t)nb0 * sizeof(float4), hipMemcpyDeviceToHost));
                        HIPCHK(w, hipMemcpy(npos.data(), w->dw.b_next_pos, (size_t)nb0 * sizeof(float4), hipMemcpyDeviceToHost));
                        HIPCHK(w, hipMemcpy(nrot.data(), w->dw.b_next_rot, (size_t)nb0 * sizeof(float4), hipMemcpyDeviceToHost));
                    }
                    auto same = [](const float4 &a, const float4 &b) { return memcmp(&a, &b, sizeof(float4)) == 0; };
                    for (int i = 0; i < nb0; ++i) {
                        HostBody &hb = w->bodies[i];
                        const BodyRow up = pack_body(hb);
                        rp_body_desc &d = hb.d;
                        if (!same(pos[i], up.pos)) { d.translation[0] = pos[i].x; d.translation[1] = pos[i].y; d.translation[2] = pos[i].z; }
                        if (!same(rot[i], up.rot)) { d.rotation[0] = rot[i].x; d.rotation[1] = rot[i].y; d.rotation[2] = rot[i].z; d.rotation[3] = rot[i].w; }
                        if (!same(lv[i], up.lv)) { d.linvel[0] = lv[i].x; d.linvel[1] = lv[i].y; d.linvel[2] = lv[i].z; }
                        if (!same(av[i], up.av)) { d.angvel[0] = av[i].x; d.angvel[1] = av[i].y; d.angvel[2] = av[i].z; }
                        if (!same(npos[i], up.npos) || !same(nrot[i], up.nrot)) {
                            hb.has_next = true; hb.next[0] = npos[i].x; hb.next[1] = npos[i].y; hb.next[2] = npos[i].z;
                            hb.next[3] = nrot[i].x; hb.next[4] = nrot[i].y; hb.next[5] = nrot[i].z; hb.next[6] = nrot[i].w;
                        }
                    }
                }
                if (nb0 > 0) {
                    HIPCHK(w, hipMemcpy(uf.data(), w->dw.b_uforce, (size_t)nb0 * sizeof(float4), hipMemcpyDeviceToHost));
                    HIPCHK(w, hipMemcpy(ut.data(), w->dw.b_utorque, (size_t)nb0 * sizeof(float4), hipMemcpyDeviceToHost));
                    HIPCHK(w, hipMemcpy(wr.data(), w->dw.b_wake_req, (size_t)nb0 * sizeof(int), hipMemcpyDeviceToHost));
                }
                const int wake_pending = fl[FL_WAKE_PENDING];
                free_device(w);
                r = finalize(w); if (r != RP_OK) return r;
                w->steps_requested = req; w->full_until = full_until;
                if (nb0 > 0) {
                    HIPCHK(w, hipMemcpy(w->dw.b_uforce, uf.data(), (size_t)nb0 * sizeof(float4), hipMemcpyHostToDevice));
                    HIPCHK(w, hipMemcpy(w->dw.b_utorque, ut.data(), (size_t)nb0 * sizeof(float4), hipMemcpyHostToDevice));
                    HIPCHK(w, hipMemcpy(w->dw.b_wake_req, wr.data(), (size_t)nb0 * sizeof(int), hipMemcpyHostToDevice));
                    if (wake_pending) HIPCHK(w, hipMemcpy(w->dw.flags + FL_WAKE_PENDING, &wake_pending, sizeof(int), hipMemcpyHostToDevice));
                }
                continue;
            }
            if (r != RP_OK) return r;
            break;
        }
        plan_from_hints(w, fl);
        memcpy(w->pinned_flags, fl, sizeof(fl));
        enqueue_solver(w); enqueue_finish(w);
        w->seq_enqueued++; w->full_steps++;
        w->never_stepped = false;
        w->hints_valid = true;
        w->full_until = w->steps_requested + 2;
        HIPCHK(w, hipGetLastError());
        return RP_OK;
    }
    // lazy hint refresh (values from some already finished step; correctness never depends on them)
    volatile int *pf = w->pinned_flags;
    {
        int fl[FL_COUNT];
        for (int k = 0; k < FL_COUNT; ++k) fl[k] = pf[k];
        int old_b = w->plan_blocks, old_g = w->plan_island_grid;
        plan_from_hints(w, fl);
        if (w->plan_blocks < old_b && w->plan_blocks * 2 >= old_b) w->plan_blocks = old_b; // hysteresis
        if (w->plan_island_grid < old_g && w->plan_island_grid * 2 >= old_g) w->plan_island_grid = old_g;
    }
    if (w->graph_stages != w->plan_stages || w->graph_blocks != w->plan_blocks || w->graph_single != w->plan_single ||
        w->graph_island_grid != w->plan_island_grid || w->graph_dense != w->plan_dense || w->graph_wide != w->plan_wide || w->graph_joint_stages != w->plan_joint_stages || w->graph_no_global != w->plan_no_global || w->graph_fused != w->plan_fused || w->graph_tile_grid != w->plan_tile_grid || w->graph_no_contacts != w->plan_no_contacts || w->graph_bare != w->plan_bare) {
        if (w->ge_whole[0] || w->ge_whole[1] || w->timed_ready[0] || w->timed_ready[1]) HIPCHK(w, hipStreamSynchronize(w->stream)); // replays of the old graphs may still be in flight
        destroy_graphs(w);
        w->graph_stages = w->plan_stages; w->graph_blocks = w->plan_blocks; w->graph_single = w->plan_single; w->graph_island_grid = w->plan_island_grid; w->graph_dense = w->plan_dense; w->graph_wide = w->plan_wide;
        w->graph_joint_stages = w->plan_joint_stages; w->graph_no_global = w->plan_no_global; w->graph_fused = w->plan_fused; w->graph_tile_grid = w->plan_tile_grid; w->graph_no_contacts = w->plan_no_contacts; w->graph_bare = w->plan_bare;
    }
    // keep the host at most a few steps ahead of the device so the hints stay fresh (the device
    // never idles: several step graphs are always queued)
    const long long max_ahead = 4;
    if (w->use_graph && !w->timers) {
        long long spins = 0;
        while ((long long)(int32_t)((uint32_t)w->seq_enqueued - (uint32_t)pf[FL_SEQ]) > max_ahead) {
            if (++spins > (1 << 14)) { if (hipStreamQuery(w->stream) != hipErrorNotReady) break; spins = 0; }
            __builtin_ia32_pause();
        }
    }
    // mode: fast graph only while the last observed steps were clean
    // sleep-enabled worlds take the fast graph while bodies are awake, nothing is about to fall asleep and no wake-up is pending (all
    // three verified on the device: k_fast_front, k_sleep_check); position-based kinematic bodies need the per-step velocity pass
    const bool sleep_fast_ok = !w->dw.sleep_enabled || (!w->dw.has_kinematic_pos && pf[FL_N_AWAKE] > 0 && !pf[FL_WAKE_PENDING]);
    bool fast = allow_fast && w->use_fast && sleep_fast_ok && w->plan_single && w->dw.n_colliders > 0 && w->steps_requested >= w->full_until;
    if (fast && (pf[FL_FAST_ABORT] || pf[FL_FULL_UPDATES] || pf[FL_LAYOUT_DIRTY] || pf[FL_TODO_COUNT] || pf[FL_PI_PENDING] || pf[FL_PJ_COUNT] || pf[FL_PI_JLINK])) {
        static const int full_after = getenv("RP_FULL_AFTER_ABORT") ? std::max(0, atoi(getenv("RP_FULL_AFTER_ABORT"))) : 3;
        fast = false;
        w->full_until = w->steps_requested + full_after;
    }
    // idle steps: the whole world sleeps (FL_N_AWAKE == 0 as of the last retired step) and nothing is pending; the
    // device re-checks and aborts otherwise (k_idle_step), the host then replays through the full graph
    if (!fast && allow_fast && w->use_fast && w->dw.sleep_enabled && !w->timers && w->steps_requested >= w->full_until) {
        if (pf[FL_N_AWAKE] == 0 && !pf[FL_FAST_ABORT] && !pf[FL_WAKE_PENDING] && !pf[FL_LAYOUT_DIRTY] && !pf[FL_BP_DIRTY]) {
            w->cur_fast = 1; w->cur_lean = 0; w->seq_enqueued++; w->fast_steps++;
            rp_launch_idle_step(w->dw, w->stream);
            HIPCHK(w, hipGetLastError());
            return RP_OK;
        } else if (pf[FL_FAST_ABORT]) w->full_until = w->steps_requested + 3;
    }
    // lean graph (rp_world.h "lean step graphs"): a MULTI-mode world on tiles whose last observed steps brought no new pair and no
    // layout change skips the nine rebuild launches; the device validates (lean_dead) and the full graph resumes a step that died
    int lean = 0;
    if (!fast && allow_fast && w->use_lean && w->use_graph && !w->timers && !w->plan_single && w->plan_tile_grid > 0 && !w->has_restitution && !w->dw.sleep_enabled &&
        !w->dw.has_force_events && !w->dw.has_sensors && !w->dw.has_kinematic_pos && w->dw.n_groups <= 1 && rp_ccd_launches(w->dw) && w->dw.n_colliders > 0) {
        if (pf[FL_FAST_ABORT]) {
            if (!w->lean_death_seen) { // once per death: stay on the full graph for a while, longer when deaths repeat
                w->lean_death_seen = true; w->lean_streak = 0;
                w->full_until = std::max(w->full_until, w->steps_requested + w->lean_backoff);
                w->lean_backoff = std::min<long long>(w->lean_backoff * 2, 256);
            }
        } else {
            w->lean_death_seen = false;
            if (pf[FL_TODO_COUNT] || pf[FL_LAYOUT_DIRTY] || pf[FL_FLOW_DIRTY] || (w->dw.n_joints > 0 && pf[FL_JOINT_DIRTY])) w->full_until = std::max(w->full_until, w->steps_requested + 3);
            else if (w->steps_requested >= w->full_until) { lean = 2; if (++w->lean_streak >= 64) { w->lean_streak = 0; w->lean_backoff = 3; } }
        }
    }
    return launch_step(w, fast ? 1 : lean);
}

// Make the device catch up with every requested step: fast steps that aborted are replayed through
// the full graph.  Synchronises the stream.
static int settle(rp_world *w) {
    if (!w->finalized) return RP_OK;
    for (int guard = 0; guard < 64; ++guard) {
        // the scalars ride the stream into the pinned hint buffer: ONE wait instead of a stream sync followed by a blocking copy
        int fl[FL_COUNT];
        HIPCHK(w, hipMemcpyAsync(w->pinned_flags, w->dw.flags, sizeof(fl), hipMemcpyDeviceToHost, w->stream));
        HIPCHK(w, hipStreamSynchronize(w->stream));
        memcpy(fl, w->pinned_flags, sizeof(fl));
        if (fl[FL_GRID_TIMEOUT]) {
            // a fused fast step waited ~1 s for a workgroup that was not resident (another process or stream holds CUs): that step
            // aborted without writing anything and is replayed below; this world stops using the single-kernel fused step
            w->use_fused = false;
            int zero = 0;
            HIPCHK(w, hipMemcpy(w->dw.flags + FL_GRID_TIMEOUT, &zero, sizeof(int), hipMemcpyHostToDevice));
            w->pinned_flags[FL_GRID_TIMEOUT] = 0;
        }
        // the device counters are 32-bit and wrap: compare modulo 2^32 (the host is never more than a few steps ahead)
        long long missing = (long long)(int32_t)((uint32_t)w->steps_requested - (uint32_t)fl[FL_STEP]);
        if (missing <= 0) {
            if (fl[FL_QUARANTINE] != w->quar_seen) {
                // the end-of-step chokepoint (body_writeback) rolled bodies back and stopped them: disable them now, as the reference
                // does at the start of the next step (quarantine.rs:131-195)
                w->quar_seen = fl[FL_QUARANTINE];
                int nb = w->dw.n_bodies;
                std::vector<int> q(std::max(nb, 1));
                if (nb > 0) HIPCHK(w, hipMemcpy(q.data(), w->dw.b_quar, nb * sizeof(int), hipMemcpyDeviceToHost));
                bool any = false;
                for (int b = 0; b < nb; ++b) if (q[b] && !w->bodies[b].quarantined && !w->bodies[b].removed) { int r = quarantine_body_at(w, b); if (r != RP_OK) return r; any = true; }
                if (any) { int r = after_topology_edit(w); if (r != RP_OK) return r; }
            }
            if (fl[FL_STEP] > w->rebase_at && !fl[FL_OVERFLOW]) {
                // 32-bit step stamps: move them back before they can wrap (the stream is idle, every requested step has retired);
                // the host's own step counters move with them
                const int delta = w->rebase_at;
                rp_launch_rebase_stamps(w->dw, w->stream, delta);
                HIPCHK(w, hipGetLastError());
                HIPCHK(w, hipStreamSynchronize(w->stream));
                w->steps_requested -= delta; w->full_until -= delta; w->eager_until -= delta;
                w->pinned_flags[FL_STEP] = fl[FL_STEP] - delta;
                w->rebases++;
            }
            return check_overflow(w, fl);
        }
        w->full_until = w->steps_requested + 3;
        w->replayed_steps += missing;
        for (long long i = 0; i < missing; ++i) { int r = step_once(w, false); if (r != RP_OK) return r; }
    }
    w->err = "settle: the device did not catch up with the requested steps";
    return RP_ERR_DEVICE;
}

extern "C" int32_t rp_step(rp_world *w, uint32_t nsteps) {
    if (!w) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) { int r = finalize(w); if (r != RP_OK) return r; }
    for (uint32_t i = 0; i < nsteps; ++i) {
        // the pair pool grows before it overflows: the hint buffer says how many slots the last retired step had in use; above 70 % the
        // world moves to arrays with twice the slots per collider (grow_begin / carry_over: every pair keeps its manifold, impulses and
        // colour — the state an insertion beyond the row capacity leaves behind).  Looked at every 16 steps: a pile has to gain 30 % more
        // pairs within that many steps to still overflow (RP_ERR_CAPACITY, as before).
        if ((i & 15) == 0 && w->hints_valid && !w->timers) {
            const volatile int *pf = w->pinned_flags;
            const int free_top = pf[FL_FREE_TOP];
            const long long live = (long long)pf[FL_POOL_TOP] - (free_top > 0 ? free_top : 0);
            if (live * 10 > (long long)w->dw.pool_cap * 7 && w->dw.pool_cap < (1 << 28)) {
                w->pairs_scale *= 2;
                int r = grow_begin(w); if (r != RP_OK) return r;
                r = finalize(w); if (r != RP_OK) return r;
            }
        }
        if ((w->steps_requested & 0xfffff) == 0xfffff) { int r = settle(w); if (r != RP_OK) return r; } // (every 2^20 steps: the step stamps' rebase lives in settle)
        w->steps_requested++;
        w->dead_pairs_possible = false; // (this step's broad-phase pass deletes the pairs of every collider removed so far)
        int r = step_once(w, true);
        if (r != RP_OK) return r;
        if (w->timers) { r = settle(w); if (r != RP_OK) return r; } // timed steps are observed one by one
    }
    return RP_OK;
}
extern "C" int32_t rp_sync(rp_world *w) {
    if (!w) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (w->finalized) return settle(w);
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return RP_OK;
}

extern "C" int32_t rp_bodies_read(rp_world *w, int32_t n, const uint64_t *handles, float *pos7_out, float *vel6_out) {
    if (!w) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) { int r = finalize(w); if (r != RP_OK) return r; }
    { int r = settle(w); if (r != RP_OK) return r; }
    int nb = w->dw.n_bodies;
    std::vector<float4> pos(nb), rot(nb), lv(nb), av(nb);
    HIPCHK(w, hipMemcpyAsync(pos.data(), w->dw.b_pos, nb * sizeof(float4), hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipMemcpyAsync(rot.data(), w->dw.b_rot, nb * sizeof(float4), hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipMemcpyAsync(lv.data(), w->dw.b_linvel, nb * sizeof(float4), hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipMemcpyAsync(av.data(), w->dw.b_angvel, nb * sizeof(float4), hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    int count = handles ? n : nb;
    for (int i = 0; i < count; ++i) {
        int b = handles ? body_of(w, handles[i], true) : i;
        if (b < 0 || b >= nb) { w->err = "rp_bodies_read: invalid handle"; return RP_ERR_INVALID; }
        if (pos7_out) { float *p = pos7_out + 7 * i; p[0] = pos[b].x; p[1] = pos[b].y; p[2] = pos[b].z; p[3] = rot[b].x; p[4] = rot[b].y; p[5] = rot[b].z; p[6] = rot[b].w; }
        if (vel6_out) { float *v = vel6_out + 6 * i; v[0] = lv[b].x; v[1] = lv[b].y; v[2] = lv[b].z; v[3] = av[b].x; v[4] = av[b].y; v[5] = av[b].z; }
    }
    return RP_OK;
}

extern "C" int32_t rp_bodies_write(rp_world *w, int32_t n, const uint64_t *handles, const float *pos7, const float *vel6) {
    if (!w || n < 0 || !handles) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) { int r = finalize(w); if (r != RP_OK) return r; }
    { int r = settle(w); if (r != RP_OK) return r; }
    bool quarantined_any = false;
    for (int i = 0; i < n; ++i) {
        int b = body_of(w, handles[i]);
        if (b < 0 || b >= w->dw.n_bodies || w->bodies[b].removed) { w->err = "rp_bodies_write: invalid handle"; return RP_ERR_INVALID; }
        if ((vel6 && !all_finite(vel6 + 6 * i, 6)) || (pos7 && !all_finite(pos7 + 7 * i, 7))) {
            // Quarantine::detect_user_changes (quarantine.rs:68-129): a non-finite user write never reaches the broad phase; the body
            // keeps its last valid pose, loses its velocities and forces and is disabled
            int r = quarantine_body_at(w, b);
            if (r != RP_OK) return r;
            quarantined_any = true;
            continue;
        }
        if (w->bodies[b].quarantined) continue; // disabled: writes are ignored
        if (vel6) {
            float4 l = mk4(vel6[6 * i], vel6[6 * i + 1], vel6[6 * i + 2], 0), a = mk4(vel6[6 * i + 3], vel6[6 * i + 4], vel6[6 * i + 5], 0);
            HIPCHK(w, hipMemcpy(w->dw.b_linvel + b, &l, sizeof(l), hipMemcpyHostToDevice));
            HIPCHK(w, hipMemcpy(w->dw.b_angvel + b, &a, sizeof(a), hipMemcpyHostToDevice));
        }
        if (pos7) {
            float4 t = mk4(pos7[7 * i], pos7[7 * i + 1], pos7[7 * i + 2], 0), q = mk4(pos7[7 * i + 3], pos7[7 * i + 4], pos7[7 * i + 5], pos7[7 * i + 6]);
            HIPCHK(w, hipMemcpy(w->dw.b_pos + b, &t, sizeof(t), hipMemcpyHostToDevice));
            HIPCHK(w, hipMemcpy(w->dw.b_rot + b, &q, sizeof(q), hipMemcpyHostToDevice));
            HIPCHK(w, hipMemcpy(w->dw.b_next_pos + b, &t, sizeof(t), hipMemcpyHostToDevice)); // set_position sets position AND next_position
            HIPCHK(w, hipMemcpy(w->dw.b_next_rot + b, &q, sizeof(q), hipMemcpyHostToDevice));
            if (w->bodies[b].d.body_type == RP_BODY_FIXED) {
                // the frame of a world-attached joint side is kept in world space (transform_to_solver_body_space): a moved fixed body
                // takes the frames of its joints along and wakes its joint partners (user_changes.rs:228-246)
                rp_body_desc &bd = w->bodies[b].d;
                for (int k = 0; k < 3; ++k) bd.translation[k] = pos7[7 * i + k];
                for (int k = 0; k < 4; ++k) bd.rotation[k] = pos7[7 * i + 3 + k];
                for (int k = 0; k < (int)w->active_joint_ids.size(); ++k) {
                    const rp_joint_desc &jd = w->joints[w->active_joint_ids[k]];
                    if (w->joint_removed[w->active_joint_ids[k]] || (jd.body1 != b && jd.body2 != b)) continue;
                    int r;
                    if (jd.body1 == b) {
                        Pose f = pose_mul(host_body_pose(w->bodies[b]), joint_local_frame(jd.local_anchor1, jd.local_basis1));
                        if ((r = poke(w, w->dw.j_f1t + k, mk4(f.t.x, f.t.y, f.t.z, 0))) != RP_OK || (r = poke(w, w->dw.j_f1r + k, mk4(f.r.x, f.r.y, f.r.z, f.r.w))) != RP_OK) return r;
                    }
                    if (jd.body2 == b) {
                        Pose f = pose_mul(host_body_pose(w->bodies[b]), joint_local_frame(jd.local_anchor2, jd.local_basis2));
                        if ((r = poke(w, w->dw.j_f2t + k, mk4(f.t.x, f.t.y, f.t.z, 0))) != RP_OK || (r = poke(w, w->dw.j_f2r + k, mk4(f.r.x, f.r.y, f.r.z, f.r.w))) != RP_OK) return r;
                    }
                    int partner = jd.body1 == b ? jd.body2 : jd.body1;
                    if (w->dw.sleep_enabled && partner != b && w->bodies[partner].d.body_type != RP_BODY_FIXED && !w->bodies[partner].removed && (r = queue_wake(w, partner, 2)) != RP_OK) return r;
                }
            }
        }
    }
    if (w->dw.sleep_enabled) {
        // set_linvel / set_position(.., wake_up = true): strong wake of the body (its whole island when asleep); a moved
        // body also wakes every body it has a pair with (pair_management.rs:236-258)
        for (int i = 0; i < n; ++i) {
            int b = body_of(w, handles[i]), lvl = pos7 ? 3 : 2;
            { int r = queue_wake(w, b, lvl); if (r != RP_OK) return r; }
        }
        if (pos7) rp_launch_wake_partners(w->dw, w->stream);
    }
    if (pos7) { // user_changes.rs: moved bodies refresh world mass properties, collider poses and AABBs
        rp_launch_init_bodies(w->dw, w->stream);
        rp_launch_collider_update(w->dw, w->stream);
    }
    if (quarantined_any) return after_topology_edit(w);
    return RP_OK;
}

// Queue a wake-up request for body b (consumed by the next full step, rp_sleep.hip) and stop enqueuing idle steps.
static int queue_wake(rp_world *w, int b, int lvl) {
    HIPCHK(w, hipMemcpy(w->dw.b_wake_req + b, &lvl, sizeof(int), hipMemcpyHostToDevice));
    int one = 1;
    HIPCHK(w, hipMemcpy(w->dw.flags + FL_WAKE_PENDING, &one, sizeof(int), hipMemcpyHostToDevice));
    w->pinned_flags[FL_WAKE_PENDING] = 1;
    return RP_OK;
}
// IslandManager::wake_up (island_manager/sleep.rs:31): takes effect at the start of the next step and wakes the
// body's whole island.
extern "C" int32_t rp_bodies_set_additional_solver_iterations(rp_world *w, int32_t n, const uint64_t *handles, const int32_t *counts) {
    if (!w || n < 0 || (n > 0 && (!handles || !counts))) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (w->finalized) { int r = settle(w); if (r != RP_OK) return r; }
    for (int i = 0; i < n; ++i) {
        int b = body_of(w, handles[i]);
        if (b < 0 || b >= (int)w->bodies.size() || w->bodies[b].removed) { w->err = "rp_bodies_set_additional_solver_iterations: invalid handle"; return RP_ERR_INVALID; }
        if (counts[i] < 0 || counts[i] > 4096) { w->err = "rp_bodies_set_additional_solver_iterations: count must be in [0, 4096]"; return RP_ERR_INVALID; }
    }
    { // validate the distinct-count limit on the prospective values before touching host or device state
        std::vector<int> prospective;
        for (size_t q = 0; q < w->bodies.size(); ++q) {
            const HostBody &hb = w->bodies[q];
            if (hb.removed || hb.quarantined) continue;
            int v = hb.d.additional_solver_iterations;
            for (int i = 0; i < n; ++i) if (body_of(w, handles[i]) == (int)q) v = counts[i];
            if (v > 0 && std::find(prospective.begin(), prospective.end(), v) == prospective.end()) prospective.push_back(v);
        }
        if ((int)prospective.size() + 1 > RP_MAX_GROUPS) { w->err = "more than 15 distinct positive additional_solver_iterations values in one world"; return RP_ERR_CAPACITY; }
    }
    for (int i = 0; i < n; ++i) {
        int b = body_of(w, handles[i]);
        w->bodies[b].d.additional_solver_iterations = counts[i];
        if (w->finalized) { int r = poke(w, w->dw.b_extra + b, (int)counts[i]); if (r != RP_OK) return r; }
    }
    if (!w->finalized) return RP_OK;
    { int r = upload_group_table(w); if (r != RP_OK) return r; }
    HIPCHK(w, hipStreamSynchronize(w->stream));
    destroy_graphs(w); // kernel arguments (DevWorld by value) hold the group count
    w->hints_valid = false; // the launch plan is rebuilt (the group solver is a plan of its own)
    return after_topology_edit(w);
}
extern "C" int32_t rp_bodies_wake_up(rp_world *w, int32_t n, const uint64_t *handles, int32_t strong) {
    if (!w || n < 0 || (n > 0 && !handles)) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) { int r = finalize(w); if (r != RP_OK) return r; }
    { int r = settle(w); if (r != RP_OK) return r; }
    for (int i = 0; i < n; ++i) {
        int b = body_of(w, handles[i]);
        if (b < 0) { w->err = "rp_bodies_wake_up: invalid handle"; return RP_ERR_INVALID; }
        { int r = queue_wake(w, b, strong ? 2 : 1); if (r != RP_OK) return r; }
    }
    return RP_OK;
}
// RigidBody::{reset_forces, reset_torques, add_force, add_torque} — rigid_body.rs:1145-1252
extern "C" int32_t rp_bodies_add_force(rp_world *w, int32_t n, const uint64_t *handles, const float *force3, const float *torque3, int32_t reset) {
    if (!w || n < 0 || (n > 0 && !handles)) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) { int r = finalize(w); if (r != RP_OK) return r; }
    { int r = settle(w); if (r != RP_OK) return r; }
    for (int i = 0; i < n; ++i) {
        int b = body_of(w, handles[i]);
        if (b < 0) { w->err = "rp_bodies_add_force: invalid handle"; return RP_ERR_INVALID; }
        float4 f, t; bool wake = false;
        HIPCHK(w, hipMemcpy(&f, w->dw.b_uforce + b, sizeof(f), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(&t, w->dw.b_utorque + b, sizeof(t), hipMemcpyDeviceToHost));
        if (reset) {
            if (f.x != 0.0f || f.y != 0.0f || f.z != 0.0f) { f = mk4(0, 0, 0, 0); wake = true; }
            if (t.x != 0.0f || t.y != 0.0f || t.z != 0.0f) { t = mk4(0, 0, 0, 0); wake = true; }
        }
        if (w->bodies[b].d.body_type == RP_BODY_DYNAMIC) {
            const float *ff = force3 ? force3 + 3 * i : nullptr, *tt = torque3 ? torque3 + 3 * i : nullptr;
            if (ff && (ff[0] != 0.0f || ff[1] != 0.0f || ff[2] != 0.0f)) { f.x = f.x + ff[0]; f.y = f.y + ff[1]; f.z = f.z + ff[2]; wake = true; }
            if (tt && (tt[0] != 0.0f || tt[1] != 0.0f || tt[2] != 0.0f)) { t.x = t.x + tt[0]; t.y = t.y + tt[1]; t.z = t.z + tt[2]; wake = true; }
        }
        HIPCHK(w, hipMemcpy(w->dw.b_uforce + b, &f, sizeof(f), hipMemcpyHostToDevice));
        HIPCHK(w, hipMemcpy(w->dw.b_utorque + b, &t, sizeof(t), hipMemcpyHostToDevice));
        if (wake && w->dw.sleep_enabled) { int r = queue_wake(w, b, 2); if (r != RP_OK) return r; }
    }
    return RP_OK;
}
// RigidBody::{apply_impulse, apply_torque_impulse} — rigid_body.rs:1304-1343
extern "C" int32_t rp_bodies_apply_impulse(rp_world *w, int32_t n, const uint64_t *handles, const float *impulse3, const float *torque_impulse3) {
    if (!w || n < 0 || (n > 0 && !handles)) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) { int r = finalize(w); if (r != RP_OK) return r; }
    { int r = settle(w); if (r != RP_OK) return r; }
    for (int i = 0; i < n; ++i) {
        int b = body_of(w, handles[i]);
        if (b < 0) { w->err = "rp_bodies_apply_impulse: invalid handle"; return RP_ERR_INVALID; }
        if (w->bodies[b].d.body_type != RP_BODY_DYNAMIC) continue;
        const float *p = impulse3 ? impulse3 + 3 * i : nullptr, *q = torque_impulse3 ? torque_impulse3 + 3 * i : nullptr;
        bool wake = false;
        if (p && (p[0] != 0.0f || p[1] != 0.0f || p[2] != 0.0f)) {
            float4 lv, im;
            HIPCHK(w, hipMemcpy(&lv, w->dw.b_linvel + b, sizeof(lv), hipMemcpyDeviceToHost));
            HIPCHK(w, hipMemcpy(&im, w->dw.b_eim + b, sizeof(im), hipMemcpyDeviceToHost));
            lv.x = lv.x + p[0] * im.x; lv.y = lv.y + p[1] * im.y; lv.z = lv.z + p[2] * im.z;
            HIPCHK(w, hipMemcpy(w->dw.b_linvel + b, &lv, sizeof(lv), hipMemcpyHostToDevice));
            wake = true;
        }
        if (q && (q[0] != 0.0f || q[1] != 0.0f || q[2] != 0.0f)) {
            float4 av, a, c;
            HIPCHK(w, hipMemcpy(&av, w->dw.b_angvel + b, sizeof(av), hipMemcpyDeviceToHost));
            HIPCHK(w, hipMemcpy(&a, w->dw.b_eii0 + b, sizeof(a), hipMemcpyDeviceToHost));
            HIPCHK(w, hipMemcpy(&c, w->dw.b_eii1 + b, sizeof(c), hipMemcpyDeviceToHost));
            // SdpMatrix3 * v with (m11 m12 m13 m22 | m23 m33)
            float rx = a.x * q[0] + a.y * q[1] + a.z * q[2], ry = a.y * q[0] + a.w * q[1] + c.x * q[2], rz = a.z * q[0] + c.x * q[1] + c.y * q[2];
            av.x = av.x + rx; av.y = av.y + ry; av.z = av.z + rz;
            HIPCHK(w, hipMemcpy(w->dw.b_angvel + b, &av, sizeof(av), hipMemcpyHostToDevice));
            wake = true;
        }
        if (wake && w->dw.sleep_enabled) { int r = queue_wake(w, b, 2); if (r != RP_OK) return r; }
    }
    return RP_OK;
}
// RigidBody::set_next_kinematic_position (rigid_body.rs:1085-1093)
extern "C" int32_t rp_bodies_set_next_kinematic_position(rp_world *w, int32_t n, const uint64_t *handles, const float *pos7) {
    if (!w || n < 0 || (n > 0 && (!handles || !pos7))) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) { int r = finalize(w); if (r != RP_OK) return r; }
    { int r = settle(w); if (r != RP_OK) return r; }
    bool quarantined_any = false;
    for (int i = 0; i < n; ++i) {
        int b = body_of(w, handles[i]);
        if (b < 0) { w->err = "rp_bodies_set_next_kinematic_position: invalid handle"; return RP_ERR_INVALID; }
        int type = w->bodies[b].d.body_type;
        if (type != RP_BODY_KINEMATIC_POSITION && type != RP_BODY_KINEMATIC_VELOCITY) continue; // "if self.is_kinematic()"
        if (!all_finite(pos7 + 7 * i, 7)) { // only the kinematic target is invalid: the pose keeps its valid half (quarantine.rs:93-99)
            int r = quarantine_body_at(w, b);
            if (r != RP_OK) return r;
            quarantined_any = true;
            continue;
        }
        float4 t = mk4(pos7[7 * i], pos7[7 * i + 1], pos7[7 * i + 2], 0), q = mk4(pos7[7 * i + 3], pos7[7 * i + 4], pos7[7 * i + 5], pos7[7 * i + 6]);
        float4 ct, cq;
        HIPCHK(w, hipMemcpy(&ct, w->dw.b_pos + b, sizeof(ct), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(&cq, w->dw.b_rot + b, sizeof(cq), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(w->dw.b_next_pos + b, &t, sizeof(t), hipMemcpyHostToDevice));
        HIPCHK(w, hipMemcpy(w->dw.b_next_rot + b, &q, sizeof(q), hipMemcpyHostToDevice));
        bool differs = ct.x != t.x || ct.y != t.y || ct.z != t.z || cq.x != q.x || cq.y != q.y || cq.z != q.z || cq.w != q.w;
        if (differs) { int r = queue_wake(w, b, 2); if (r != RP_OK) return r; } // wake_up(true)
    }
    if (quarantined_any) return after_topology_edit(w);
    return RP_OK;
}
// RigidBody::is_sleeping per handle (1 = asleep).
// RigidBodySet::iter / ColliderSet::iter as handles (rigid_body_set.rs, collider_set.rs; Arena::iter, arena.rs:665-700): the handle of
// every arena row in index order — generation << 32 | index of the occupant inserted last (free rows: of the occupant removed last,
// which no entry point but rp_bodies_read accepts any more).  Returns the number of rows; writes min(rows, cap) handles.
extern "C" int32_t rp_bodies_handles(const rp_world *w, int32_t cap, uint64_t *out) {
    if (!w || cap < 0 || (cap > 0 && !out)) return RP_ERR_INVALID;
    const int n = (int)w->bodies.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = ((uint64_t)w->body_gen[(size_t)i] << 32) | (uint64_t)(uint32_t)i;
    return n;
}
extern "C" int32_t rp_colliders_handles(const rp_world *w, int32_t cap, uint64_t *out) {
    if (!w || cap < 0 || (cap > 0 && !out)) return RP_ERR_INVALID;
    const int n = (int)w->colliders.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = ((uint64_t)w->coll_gen[(size_t)i] << 32) | (uint64_t)(uint32_t)i;
    return n;
}
extern "C" int32_t rp_bodies_is_sleeping(rp_world *w, int32_t n, const uint64_t *handles, int32_t *out) {
    if (!w || n < 0 || (n > 0 && (!handles || !out))) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) { int r = finalize(w); if (r != RP_OK) return r; }
    { int r = settle(w); if (r != RP_OK) return r; }
    std::vector<int> fl(w->dw.n_bodies);
    if (!fl.empty()) HIPCHK(w, hipMemcpy(fl.data(), w->dw.b_flags, fl.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        int b = body_of(w, handles[i], true);
        if (b < 0) { w->err = "rp_bodies_is_sleeping: invalid handle"; return RP_ERR_INVALID; }
        out[i] = ((fl[b] & RP_BF_TYPE_MASK) != RP_BODY_FIXED && (fl[b] & RP_BF_SLEEPING)) ? 1 : 0;
    }
    return RP_OK;
}

// IslandManager::persistent_island_of (manager.rs:214-220) per handle: -1 for fixed / removed bodies and in worlds that hold no
// sleepable body (such worlds keep no islands).  Only equality is meaningful in the reference; here the ids are the oracle's.
extern "C" int32_t rp_bodies_persistent_island(rp_world *w, int32_t n, const uint64_t *handles, int32_t *out) {
    if (!w || n < 0 || (n > 0 && (!handles || !out))) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) { int r = finalize(w); if (r != RP_OK) return r; }
    { int r = settle(w); if (r != RP_OK) return r; }
    std::vector<int> isl(std::max(w->dw.n_bodies, 1), -1);
    if (w->dw.sleep_enabled && w->dw.n_bodies > 0) HIPCHK(w, hipMemcpy(isl.data(), w->dw.b_isl, (size_t)w->dw.n_bodies * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        int b = body_of(w, handles[i], true);
        if (b < 0 || b >= w->dw.n_bodies) { w->err = "rp_bodies_persistent_island: invalid handle"; return RP_ERR_INVALID; }
        out[i] = w->bodies[b].removed ? -1 : isl[b];
    }
    return RP_OK;
}
// Proximity groups (include/rapier_hip.h): union-find on the host over the live pair slots (read back once) and the joints.
extern "C" int32_t rp_bodies_proximity_group(rp_world *w, int32_t n, const uint64_t *handles, int32_t *out) {
    if (!w || n < 0 || (n > 0 && (!handles || !out))) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) { int r = finalize(w); if (r != RP_OK) return r; }
    { int r = settle(w); if (r != RP_OK) return r; }
    const int nb = w->dw.n_bodies;
    std::vector<int> parent(std::max(nb, 1));
    for (int i = 0; i < nb; ++i) parent[i] = i;
    auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    auto unite = [&](int a, int b) { a = find(a); b = find(b); if (a != b) { if (a < b) parent[b] = a; else parent[a] = b; } };
    auto links = [&](int b) { return b >= 0 && b < nb && !w->bodies[b].removed && w->bodies[b].d.body_type != RP_BODY_FIXED; };
    int top = 0;
    HIPCHK(w, hipMemcpy(&top, w->dw.flags + FL_POOL_TOP, sizeof(int), hipMemcpyDeviceToHost));
    top = std::min(top, w->dw.pool_cap);
    if (top > 0) {
        std::vector<int> c1(top); std::vector<int2> rb(top);
        HIPCHK(w, hipMemcpy(c1.data(), w->dw.p_c1, (size_t)top * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(rb.data(), w->dw.p_rb, (size_t)top * sizeof(int2), hipMemcpyDeviceToHost));
        for (int s = 0; s < top; ++s) if (c1[s] >= 0 && links(rb[s].x) && links(rb[s].y)) unite(rb[s].x, rb[s].y);
    }
    for (size_t j = 0; j < w->joints.size(); ++j) if (!w->joint_removed[j] && links((int)w->joints[j].body1) && links((int)w->joints[j].body2)) unite((int)w->joints[j].body1, (int)w->joints[j].body2);
    for (int i = 0; i < n; ++i) {
        int b = body_of(w, handles[i], true);
        if (b < 0 || b >= nb) { w->err = "rp_bodies_proximity_group: invalid handle"; return RP_ERR_INVALID; }
        out[i] = links(b) ? find(b) : -1;
    }
    return RP_OK;
}
// The bodies the shard guard caught since the last call, and the world goes on: the guard bit leaves FL_OVERFLOW, so rp_sync / reads
// succeed again.  What the caller does with them is SURVEY section 8e's "migrate the smaller island": move the bodies' proximity group to
// the shard whose box they reached (rapier_amd/sharding.py: migrate_groups), refresh the guards, continue.
extern "C" int32_t rp_world_shard_guard_take_hits(rp_world *w, int32_t cap, uint64_t *bodies_out) {
    if (!w || cap < 0 || (cap > 0 && !bodies_out)) return RP_ERR_INVALID;
    if (!w->finalized) return 0;
    HIPCHK(w, hipSetDevice(w->device));
    {   // every requested step has run (an error return of settle() for the guard bit itself is what this call is for)
        int r = settle(w);
        if (r != RP_OK && !(r == RP_ERR_INVALID && w->err.find("shard guard") != std::string::npos)) return r;
    }
    int ovf = 0;
    HIPCHK(w, hipMemcpy(&ovf, w->dw.flags + FL_OVERFLOW, sizeof(int), hipMemcpyDeviceToHost));
    if (!(ovf & RP_OVF_SHARD)) return 0;
    const int nb = w->dw.n_bodies;
    std::vector<int> hit((size_t)std::max(nb, 1), 0);
    if (nb > 0) HIPCHK(w, hipMemcpy(hit.data(), w->dw.sg_hit, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost));
    int n = 0;
    for (int b = 0; b < nb; ++b) if (hit[(size_t)b] && !w->bodies[(size_t)b].removed) { if (n < cap) bodies_out[n] = ((uint64_t)w->body_gen[(size_t)b] << 32) | (uint64_t)(uint32_t)b; ++n; }
    if (n <= cap) { // everything was handed out: clear the marks and the bit (a short buffer leaves both for the next call)
        if (nb > 0) HIPCHK(w, hipMemsetAsync(w->dw.sg_hit, 0, (size_t)nb * sizeof(int), w->stream));
        ovf &= ~RP_OVF_SHARD;
        HIPCHK(w, hipMemcpy(w->dw.flags + FL_OVERFLOW, &ovf, sizeof(int), hipMemcpyHostToDevice));
        w->pinned_flags[FL_OVERFLOW] = ovf;
        w->err.clear();
    }
    return n;
}
// max |linvel| over the non-fixed bodies, reduced on the device (non-negative floats order like their bit patterns)
__global__ void k_max_linear_speed(DevWorld w, unsigned *out) {
    float m = 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w.n_bodies; i += gridDim.x * blockDim.x) {
        if ((w.b_flags[i] & RP_BF_TYPE_MASK) == RP_BODY_FIXED) continue;
        const float4 v = w.b_linvel[i];
        const float s = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
        if (s == s) m = fmaxf(m, s);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(out, __float_as_uint(m));
}
// How far the fastest body of this shard travels per step: what a caller that looks at the guard every k steps adds to the clearance of
// the boxes it hands to rp_world_set_shard_guard (2 * speed * dt * k: both sides may move) — rapier_amd/sharding.py: ShardSet.
extern "C" int32_t rp_world_max_linear_speed(rp_world *w, float *out) {
    if (!w || !out) return RP_ERR_INVALID;
    *out = 0.0f;
    if (!w->finalized || w->dw.n_bodies == 0) return RP_OK;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->d_speed) HIPCHK(w, hipMalloc((void **)&w->d_speed, sizeof(unsigned)));
    HIPCHK(w, hipMemsetAsync(w->d_speed, 0, sizeof(unsigned), w->stream));
    const int nb = w->dw.n_bodies;
    hipLaunchKernelGGL(k_max_linear_speed, dim3(std::min((nb + 255) / 256, 1024)), dim3(256), 0, w->stream, w->dw, w->d_speed);
    HIPCHK(w, hipGetLastError());
    unsigned bits = 0;
    HIPCHK(w, hipMemcpyAsync(&bits, w->d_speed, sizeof(unsigned), hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    memcpy(out, &bits, sizeof(float));
    return RP_OK;
}
extern "C" int32_t rp_world_set_shard_guard(rp_world *w, int32_t n, const float *bmin, const float *bmax) {
    if (!w || n < 0 || (n > 0 && (!bmin || !bmax))) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    w->guard_min.clear(); w->guard_max.clear(); w->guard_start.clear(); w->guard_items.clear();
    if (n > 0) {
        // coarse uniform grid over the boxes: cell = the largest box edge (a box then touches at most 2 x 2 x 2 cells), CSR lists
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, edge = 0.0f;
        for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) {
            const float a = bmin[3 * i + k], b = bmax[3 * i + k];
            if (!(a <= b) || !std::isfinite(a) || !std::isfinite(b)) { w->err = "rp_world_set_shard_guard: bad box"; return RP_ERR_INVALID; }
            lo[k] = std::min(lo[k], a); hi[k] = std::max(hi[k], b); edge = std::max(edge, b - a);
        }
        float cell = std::max(edge, 1.0e-3f);
        for (;;) { // at most 2^22 cells
            double cells = 1.0; for (int k = 0; k < 3; ++k) cells *= std::floor((hi[k] - lo[k]) / cell) + 1.0;
            if (cells <= (double)(1 << 22)) break;
            cell *= 2.0f;
        }
        for (int k = 0; k < 3; ++k) { w->guard_origin[k] = lo[k]; w->guard_dims[k] = (int)std::floor((hi[k] - lo[k]) / cell) + 1; }
        w->guard_cell = cell;
        const int nc = w->guard_dims[0] * w->guard_dims[1] * w->guard_dims[2];
        auto range = [&](int i, int k, int &a, int &b) {
            a = std::min(std::max((int)std::floor((bmin[3 * i + k] - lo[k]) / cell), 0), w->guard_dims[k] - 1);
            b = std::min(std::max((int)std::floor((bmax[3 * i + k] - lo[k]) / cell), 0), w->guard_dims[k] - 1);
        };
        std::vector<int> count(nc + 1, 0);
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = 0; i < n; ++i) {
                int x0, x1, y0, y1, z0, z1; range(i, 0, x0, x1); range(i, 1, y0, y1); range(i, 2, z0, z1);
                for (int z = z0; z <= z1; ++z) for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) {
                    const int c = (z * w->guard_dims[1] + y) * w->guard_dims[0] + x;
                    if (pass == 0) count[c + 1]++; else w->guard_items[count[c]++] = i;
                }
            }
            if (pass == 0) { for (int c = 0; c < nc; ++c) count[c + 1] += count[c]; w->guard_start = count; w->guard_items.assign((size_t)count[nc], 0); }
        }
        for (int i = 0; i < n; ++i) { w->guard_min.push_back(mk4(bmin[3 * i], bmin[3 * i + 1], bmin[3 * i + 2], 0)); w->guard_max.push_back(mk4(bmax[3 * i], bmax[3 * i + 1], bmax[3 * i + 2], 0)); }
    }
    if (!w->finalized) return RP_OK; // uploaded when the device world is built
    { int r = settle(w); if (r != RP_OK) return r; }
    { int r = upload_shard_guard(w); if (r != RP_OK) return r; }
    destroy_graphs(w); // the captured launches hold the old DevWorld
    return RP_OK;
}
// The time a guard hit may wait for the caller (the steps between two rp_world_shard_guard_take_hits x dt): the device tests every
// rewritten fat AABB INFLATED by |linvel of its body| x horizon, so a body is caught that many steps before it reaches a foreign box —
// per body, not one world-wide clearance that would merge shards which merely stand close (rapier_amd/sharding.py: ShardSet).
// A batch of small, independent worlds in ONE device world: everything inserted after this call belongs to a new
// sub-world; colliders of different sub-worlds never form a pair (the broad phase keys its cells with the sub-world, pair_allowed
// rejects what still meets), so the sub-worlds may occupy the same space.  They share the integration parameters, the step counter and
// every launch: a step of the batch costs what a step of one world with that many islands costs, not n small launches sequences.
// In reference terms: one World whose PhysicsHooks::filter_contact_pair rejects pairs across sub-worlds.
extern "C" int32_t rp_world_begin_subworld(rp_world *w) {
    if (!w) return RP_ERR_INVALID;
    if (w->colliders.empty() && w->bodies.empty() && w->n_sub == 1) return 0; // the implicit first sub-world is still empty
    w->cur_sub = w->n_sub++;
    if (w->finalized) { // kernels take n_sub from the DevWorld they are launched with
        HIPCHK(w, hipSetDevice(w->device));
        int r = settle(w); if (r != RP_OK) return r;
        if (w->n_sub > w->dw.sub_cap) { r = rebuild_begin(w); if (r != RP_OK) return r; } // the per-sub-world tables are full: the next step rebuilds the device world from the current state
        else {
            w->dw.n_sub = w->n_sub;
            destroy_graphs(w);
            // the broad phase's large list is segmented by sub-world: the next pass builds it afresh (and finds the newcomers)
            int one = 1; HIPCHK(w, hipMemcpyAsync(w->dw.flags + FL_BP_FORCE_FULL, &one, sizeof(int), hipMemcpyHostToDevice, w->stream)); HIPCHK(w, hipStreamSynchronize(w->stream));
        }
    }
    return w->cur_sub;
}
// rp_step for several worlds from one host thread: every world's steps are enqueued on its own stream before any of them is waited for,
// so worlds that do not fill the device overlap.  (Small worlds of ONE parameter set are better served as sub-worlds of one world.)
extern "C" int32_t rp_step_many(rp_world *const *worlds, int32_t n, int32_t steps) {
    if (!worlds || n < 0 || steps < 0) return RP_ERR_INVALID;
    for (int32_t i = 0; i < n; ++i) if (!worlds[i]) return RP_ERR_INVALID;
    for (int32_t s = 0; s < steps; ++s) // step-major: the worlds advance together and their launches interleave on the device
        for (int32_t i = 0; i < n; ++i) { int r = rp_step(worlds[i], 1); if (r != RP_OK) return r; }
    return RP_OK;
}
extern "C" int32_t rp_world_set_shard_guard_horizon(rp_world *w, float seconds) {
    if (!w || !(seconds >= 0.0f) || !std::isfinite(seconds)) return RP_ERR_INVALID;
    if (seconds == w->guard_horizon) return RP_OK;
    w->guard_horizon = seconds;
    if (!w->finalized) return RP_OK;
    HIPCHK(w, hipSetDevice(w->device));
    { int r = settle(w); if (r != RP_OK) return r; }
    w->dw.sg_horizon = seconds;
    destroy_graphs(w); // the captured launches hold the old DevWorld
    return RP_OK;
}
// Debug aid (not part of include/rapier_hip.h): how often the step stamps moved back (k_rebase_stamps)
extern "C" int64_t rp_debug_rebases(const rp_world *w) { return w ? (int64_t)w->rebases : -1; }
// Debug aid (not part of include/rapier_hip.h): the island machinery's counters (slots of the oracle's RO_IS_*), the scan stamp, the
// pending split (-1 = none) and, for `island` >= 0, its table row (in use, bodies, dirty, denied-until, sleeping).
extern "C" int32_t rp_debug_islands(rp_world *w, int32_t *stats16, int32_t *stamp_pending2, int32_t island, int32_t *row5) {
    if (!w || !w->finalized) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    { int r = settle(w); if (r != RP_OK) return r; }
    if (stats16) HIPCHK(w, hipMemcpy(stats16, w->dw.pi_stats, 16 * sizeof(int), hipMemcpyDeviceToHost));
    if (stamp_pending2) {
        unsigned long long w64 = 0; int pend = 0;
        HIPCHK(w, hipMemcpy(&w64, w->dw.pi_w64, sizeof(w64), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(&pend, w->dw.flags + FL_PI_PENDING, sizeof(int), hipMemcpyDeviceToHost));
        stamp_pending2[0] = (int)(unsigned)(w64 & 0xffffffffull); stamp_pending2[1] = pend - 1;
    }
    if (row5 && island >= 0 && island < w->dw.n_bodies) {
        int *src[5] = {w->dw.pi_used, w->dw.pi_nb, w->dw.pi_dirty, w->dw.pi_denied, w->dw.pi_sleeping};
        for (int k = 0; k < 5; ++k) HIPCHK(w, hipMemcpy(row5 + k, src[k] + island, sizeof(int), hipMemcpyDeviceToHost));
    }
    return RP_OK;
}

// ---- removal (RigidBodySet::remove / ColliderSet::remove / ImpulseJointSet::remove) ---------------
// Arena slots are kept as tombstones (indices stay stable, handles of removed items become invalid).
// A removed collider loses its interaction groups, so the next broad-phase pass deletes its pairs
// (DeletePair frees their colours; the other pairs keep their warm-start data, like
// NarrowPhase::handle_user_changes, pair_management.rs:24-203); a removed body becomes an inert fixed
// body without colliders or joints; a removed joint loses its rows.
template <typename T> static int poke(rp_world *w, T *dst, const T &v) {
    HIPCHK(w, hipMemcpy(dst, &v, sizeof(T), hipMemcpyHostToDevice));
    return RP_OK;
}
static int set_flag(rp_world *w, int slot, int v) { return poke(w, w->dw.flags + slot, v); }
// keep_grid: the edit only ADDED rows (bodies, colliders): the broad-phase grid still describes every collider it was built from, and a
// new collider — its fat AABB starts inverted, so the next k_collider_update rewrites it and queues it like a collider that moved — finds
// its partners in an incremental pass.  (b3d_large_world drops a sphere every five steps onto a million static boxes: a full rebuild
// per drop was 25 ms.)
void rp_launch_purge_dead_pairs(const DevWorld &w, hipStream_t st);
// NarrowPhase::handle_user_changes for removed colliders (pair_management.rs:24-203) ahead of time: the pairs of every removed collider
// leave the device pair set NOW, with the effects the next broad-phase pass would have had (Stopped | REMOVED events stamped with the
// coming step, wake-ups, freed colours).  The reference removes them by HANDLE at the start of the next step; here a pair names its
// colliders by index, so it must not outlive the slot: called before an arena slot is handed out again.
static int purge_dead_pairs(rp_world *w) {
    if (!w->finalized || !w->dead_pairs_possible) return RP_OK;
    rp_launch_purge_dead_pairs(w->dw, w->stream);
    HIPCHK(w, hipStreamSynchronize(w->stream));
    w->dead_pairs_possible = false;
    return RP_OK;
}
// every persistent row of one body / collider back to the state finalize() gives a fresh row (the allocation's fill byte): the slot is
// about to hold another occupant (colour masks, island ids, sleep state, warm-start words ... of the previous one must not leak)
static int reset_row(rp_world *w, int dom, int i) {
    // (tables that are merely SIZED like the body arrays — index = persistent-island id, sleep label, LDS-island id — are not rows of a
    // body: the island table above all must survive; the others are rebuilt by the layout / label passes)
    static const size_t not_rows[] = {offsetof(DevWorld, pi_used), offsetof(DevWorld, pi_nb), offsetof(DevWorld, pi_dirty), offsetof(DevWorld, pi_denied), offsetof(DevWorld, pi_sleeping),
                                      offsetof(DevWorld, pi_free), offsetof(DevWorld, lab_wake), offsetof(DevWorld, lab_awake), offsetof(DevWorld, isl_body_begin), offsetof(DevWorld, isl_nb),
                                      offsetof(DevWorld, isl_cons_begin), offsetof(DevWorld, isl_nc), offsetof(DevWorld, isl_fill_b), offsetof(DevWorld, isl_fill_c), offsetof(DevWorld, isl_bodies),
                                      offsetof(DevWorld, isl_sorted), offsetof(DevWorld, isl_nstages), offsetof(DevWorld, isl_ni), offsetof(DevWorld, isl_icons_begin), offsetof(DevWorld, isl_fill_i),
                                      offsetof(DevWorld, isl_inc_begin), offsetof(DevWorld, isl_inc_cnt), offsetof(DevWorld, r_nb), offsetof(DevWorld, r_nc), offsetof(DevWorld, r_ni), offsetof(DevWorld, r_island)};
    for (const AllocRec &a : w->allocs) {
        if (a.dom != dom) continue;
        if (dom == DOM_BODY && std::find(std::begin(not_rows), std::end(not_rows), a.off) != std::end(not_rows)) continue;
        for (int p = 0; p < a.planes; ++p)
            HIPCHK(w, hipMemsetAsync((char *)a.ptr + ((size_t)p * a.stride + (size_t)i) * a.per * a.elem, a.fill, a.per * a.elem, w->stream));
    }
    HIPCHK(w, hipStreamSynchronize(w->stream)); // (the row uploads that follow are small copies from pageable memory: the fills have landed before any of them is issued)
    return RP_OK;
}
void rp_launch_edit_flags(const DevWorld &w, hipStream_t st, int keep_grid);
static int after_topology_edit(rp_world *w, bool keep_grid) {
    if (!w->finalized) return RP_OK;
    // the dirty flags of an edit in ONE launch behind whatever the edit queued (round 3: five blocking 4-byte copies + a stream wait —
    // most of the 1.7 ms an insertion cost); nothing here waits: every entry point that reads the device settles the stream first
    rp_launch_edit_flags(w->dw, w->stream, keep_grid ? 1 : 0); // (!keep_grid: colliders changed their filters: the next broad-phase pass is a full rebuild)
    w->pinned_flags[FL_LAYOUT_DIRTY] = 1; // keeps the next steps on the full graph until the device reports a clean state
    w->full_until = w->steps_requested + 3;
    w->eager_until = w->steps_requested + 8; // (graphs are captured again once eight steps went by without another edit)
    rp_launch_init_bodies(w->dw, w->stream);
    HIPCHK(w, hipGetLastError());
    return RP_OK;
}
static int remove_joint_at(rp_world *w, int j) {
    if (w->joint_removed[j]) return RP_OK;
    w->joint_removed[j] = 1;
    if (!w->finalized) return RP_OK;
    if (!w->joints[j].contacts_enabled) { // the pairs it filtered are evaluated again
        std::vector<unsigned long long> nck = no_contact_keys(w);
        if (!nck.empty()) HIPCHK(w, hipMemcpy(w->dw.nc_keys, nck.data(), nck.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        w->dw.n_nc = (int)nck.size();
        HIPCHK(w, hipStreamSynchronize(w->stream));
        destroy_graphs(w); // kernel arguments (DevWorld by value) hold n_nc
        w->dw.n_nc = std::max(w->dw.n_nc, 1); rp_launch_clear_no_contact(w->dw, w->stream); w->dw.n_nc = (int)nck.size();
    }
    for (int k = 0; k < (int)w->active_joint_ids.size(); ++k) {
        if (w->active_joint_ids[k] != j) continue;
        const rp_joint_desc &jd = w->joints[j];
        int r;
        if ((r = poke(w, w->dw.j_b1 + k, -1)) != RP_OK || (r = poke(w, w->dw.j_b2 + k, -1)) != RP_OK || (r = poke(w, w->dw.j_locked + k, 0)) != RP_OK || (r = poke(w, w->dw.j_limited + k, 0)) != RP_OK || (r = poke(w, w->dw.j_motor + k, 0)) != RP_OK ||
            (r = poke(w, w->dw.j_imp + k, mk4(0, 0, 0, 0))) != RP_OK || (r = poke(w, w->dw.j_imp_ang + k, mk4(0, 0, 0, 0))) != RP_OK) return r;
        if (w->dw.sleep_enabled) rp_launch_pj_append_joint(w->dw, w->stream, k, jd.body1, jd.body2, j); // ImpulseJointIslandEvent::Unlink (journaled for resolve_removals)
        for (int b : {jd.body1, jd.body2}) {
            if (w->bodies[b].d.body_type == RP_BODY_FIXED || w->bodies[b].removed) continue;
            if (w->dw.sleep_enabled && (r = queue_wake(w, b, 2)) != RP_OK) return r; // ImpulseJointSet::remove(.., wake_up = true)
            int cnt = 0;
            for (size_t q = 0; q < w->joints.size(); ++q) if (!w->joint_removed[q] && (w->joints[q].body1 == b || w->joints[q].body2 == b)) cnt++;
            if ((r = poke(w, w->dw.b_njoints + b, cnt)) != RP_OK) return r;
        }
    }
    return RP_OK;
}
static int remove_collider_at(rp_world *w, int c) {
    if (w->collider_removed[c]) return RP_OK;
    w->collider_removed[c] = 1;
    w->coll_arena_gen++; w->coll_free.push_back(c); // Arena::remove (arena.rs:353-380): the slot heads the free list, the generation counts removals
    w->dead_pairs_possible = true;
    int parent = w->collider_parent[c];
    if (parent >= 0) { std::vector<int> &cl = w->bodies[parent].cols; cl.erase(std::remove(cl.begin(), cl.end(), c), cl.end()); } // (the slot may soon belong to another body)
    if (parent >= 0) { w->bodies[parent].ncolliders--; recompute_mass(w, parent); }
    if (!w->finalized) return RP_OK;
    uint2 none; none.x = 0; none.y = 0;
    int r = poke(w, w->dw.c_groups + c, none);
    if (r != RP_OK) return r;
    if (parent >= 0) {
        // the body's mass properties follow its remaining colliders (local centre of mass, principal inertia AND frame, the
        // sleep metric's max_extent), and so do the CoM-space frames of its joints
        if ((r = poke(w, w->dw.c_sibling + c, -1)) != RP_OK) return r;
        if ((r = upload_collider_chain(w, parent)) != RP_OK) return r;
        if ((r = upload_body_row_mass(w, parent)) != RP_OK) return r;
        if ((r = refresh_joint_frames(w, parent)) != RP_OK) return r;
    }
    return RP_OK;
}
static int handle_index(uint64_t h) { return (h >> 32) == 0 ? (int)(h & 0xffffffffull) : -1; } // (impulse joints: dense indices)
// body / collider handles: index + generation (Arena::get: the generations must match); -1 = unknown, stale, or — unless asked for — removed
static int body_of(const rp_world *w, uint64_t h, bool allow_removed) {
    const uint64_t i = h & 0xffffffffull;
    if (i >= w->bodies.size() || (uint32_t)(h >> 32) != w->body_gen[(size_t)i]) return -1;
    return (w->bodies[(size_t)i].removed && !allow_removed) ? -1 : (int)i;
}
static int collider_of(const rp_world *w, uint64_t h) {
    const uint64_t i = h & 0xffffffffull;
    if (i >= w->colliders.size() || (uint32_t)(h >> 32) != w->coll_gen[(size_t)i] || w->collider_removed[(size_t)i]) return -1;
    return (int)i;
}

extern "C" int32_t rp_impulse_joints_remove(rp_world *w, int32_t n, const uint64_t *handles) {
    if (!w || n < 0 || (n > 0 && !handles)) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (w->finalized) { int r = settle(w); if (r != RP_OK) return r; }
    for (int i = 0; i < n; ++i) {
        int j = handle_index(handles[i]);
        if (j < 0 || j >= (int)w->joints.size() || w->joint_removed[j]) { w->err = "rp_impulse_joints_remove: invalid handle"; return RP_ERR_INVALID; }
        int r = remove_joint_at(w, j);
        if (r != RP_OK) return r;
    }
    return after_topology_edit(w);
}
extern "C" int32_t rp_colliders_remove(rp_world *w, int32_t n, const uint64_t *handles) {
    if (!w || n < 0 || (n > 0 && !handles)) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (w->finalized) { int r = settle(w); if (r != RP_OK) return r; }
    for (int i = 0; i < n; ++i) {
        int c = collider_of(w, handles[i]);
        if (c < 0) { w->err = "rp_colliders_remove: invalid handle"; return RP_ERR_INVALID; }
        int r = remove_collider_at(w, c);
        if (r != RP_OK) return r;
    }
    { int r = purge_dead_pairs(w); if (r != RP_OK) return r; } // (see rp_bodies_remove)
    return after_topology_edit(w, true);
}
// A body leaves the simulation: its colliders and joints go, the device row becomes an inert fixed body.  Shared by
// rp_bodies_remove (the handle dies) and the quarantine (RigidBody::set_enabled(false): the handle stays readable).
static int detach_body_at(rp_world *w, int b) {
    int r;
    { // the attached colliders go in attachment order (rigid_body_set.rs:140-150 walks rb.colliders()): the order of the free list
        const std::vector<int> cols = w->bodies[b].cols;
        for (int c : cols) if (c >= 0 && c < (int)w->colliders.size() && w->collider_parent[c] == b && (r = remove_collider_at(w, c)) != RP_OK) return r;
    }
    for (size_t c = 0; c < w->colliders.size(); ++c) if (w->collider_parent[c] == b && (r = remove_collider_at(w, (int)c)) != RP_OK) return r;
    for (size_t j = 0; j < w->joints.size(); ++j) if ((w->joints[j].body1 == b || w->joints[j].body2 == b) && (r = remove_joint_at(w, (int)j)) != RP_OK) return r;
    HostBody &hb = w->bodies[b];
    if (w->finalized && w->dw.sleep_enabled && hb.d.body_type != RP_BODY_FIXED) rp_launch_pi_remove_body(w->dw, w->stream, b); // rigid_body_removed_or_disabled (manager.rs:62-78)
    hb.d.body_type = RP_BODY_FIXED; hb.isl = -1;
    for (int k = 0; k < 3; ++k) { hb.d.linvel[k] = 0.0f; hb.d.angvel[k] = 0.0f; }
    if (w->finalized) {
        int fl = RP_BODY_FIXED | (hb.d.gyroscopic ? RP_BF_GYRO : 0) | (hb.d.allow_fast_rotation ? RP_BF_FASTROT : 0) | (((int)(hb.d.dominance & 0xff)) << RP_BF_DOM_SHIFT);
        if ((r = poke(w, w->dw.b_flags + b, fl)) != RP_OK || (r = poke(w, w->dw.b_linvel + b, mk4(0, 0, 0, 0))) != RP_OK ||
            (r = poke(w, w->dw.b_angvel + b, mk4(0, 0, 0, 0))) != RP_OK || (r = poke(w, w->dw.b_njoints + b, 0)) != RP_OK ||
            (r = poke(w, w->dw.b_uforce + b, mk4(0, 0, 0, 0))) != RP_OK || (r = poke(w, w->dw.b_utorque + b, mk4(0, 0, 0, 0))) != RP_OK) return r;
    }
    return RP_OK;
}
extern "C" int32_t rp_bodies_remove(rp_world *w, int32_t n, const uint64_t *handles) {
    if (!w || n < 0 || (n > 0 && !handles)) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (w->finalized) { int r = settle(w); if (r != RP_OK) return r; }
    for (int i = 0; i < n; ++i) {
        int b = body_of(w, handles[i]);
        if (b < 0) { w->err = "rp_bodies_remove: invalid handle"; return RP_ERR_INVALID; }
        int r = detach_body_at(w, b);
        if (r != RP_OK) return r;
        w->bodies[b].removed = true;
        w->body_arena_gen++; w->body_free.push_back(b);
    }
    // the pairs of the removed colliders leave the pair set now (what the next pass would do: purge_dead_pairs), so the broad-phase grid
    // can stay in service — its entries of a dead collider pass no filter — and the next pass is incremental, not a rebuild
    { int r = purge_dead_pairs(w); if (r != RP_OK) return r; }
    return after_topology_edit(w, true);
}
// Quarantine::detect_user_changes / apply_end_step (quarantine.rs:68-195): the body keeps its last valid pose, its velocities and
// user forces are zeroed and it is disabled (RigidBody::set_enabled(false): no colliders in the broad phase, no joints, not in
// the active set).  Re-enabling is not offered by this ABI.
static int quarantine_body_at(rp_world *w, int b) {
    HostBody &hb = w->bodies[b];
    if (hb.quarantined || hb.removed) return RP_OK;
    int r = detach_body_at(w, b);
    if (r != RP_OK) return r;
    hb.quarantined = true;
    if (w->finalized && (r = poke(w, w->dw.b_quar + b, 1)) != RP_OK) return r;
    w->quarantine_log.push_back(b);
    return RP_OK;
}
static bool all_finite(const float *v, int n) { for (int k = 0; k < n; ++k) if (!std::isfinite(v[k])) return false; return true; }

// Event queues (EventHandler, pipeline/event_handler.rs:94-160): drained oldest first.
static int drain_count(rp_world *w, int slot, int *count) {
    int r = settle(w); if (r != RP_OK) return r;
    HIPCHK(w, hipMemcpy(count, w->dw.flags + slot, sizeof(int), hipMemcpyDeviceToHost));
    return RP_OK;
}
extern "C" int32_t rp_collision_events_read(rp_world *w, int32_t cap, rp_collision_event *out) {
    if (!w || cap < 0 || (cap > 0 && !out)) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) return 0;
    int n = 0; { int r = drain_count(w, FL_EV_COL, &n); if (r != RP_OK) return r; }
    int stored = std::min(n, w->dw.ev_cap);
    if (!out) return stored;
    if (n > stored) w->err = "rp_collision_events_read: the collision event queue overflowed; the newest events were dropped";
    std::vector<int4> ev(stored);
    if (stored) HIPCHK(w, hipMemcpy(ev.data(), w->dw.ev_col, stored * sizeof(int4), hipMemcpyDeviceToHost));
    std::sort(ev.begin(), ev.end(), [](const int4 &a, const int4 &b) { if (a.w != b.w) return a.w < b.w; if (a.x != b.x) return a.x < b.x; if (a.y != b.y) return a.y < b.y; return a.z < b.z; });
    const int written = std::min(stored, cap);
    for (int i = 0; i < written; ++i) { out[i].collider1 = ev[i].x; out[i].collider2 = ev[i].y; out[i].started = ev[i].z & 0xff; out[i].flags = ev[i].z >> 8; out[i].step = ev[i].w; }
    // only the events handed out leave the queue: the rest moves to its front (Started / Stopped are edge-triggered, a dropped one is lost for good)
    const int rest = stored - written;
    if (rest > 0) HIPCHK(w, hipMemcpy(w->dw.ev_col, ev.data() + written, rest * sizeof(int4), hipMemcpyHostToDevice));
    HIPCHK(w, hipMemcpy(w->dw.flags + FL_EV_COL, &rest, sizeof(int), hipMemcpyHostToDevice));
    return written;
}
extern "C" int32_t rp_intersection_pairs_read(rp_world *w, int32_t cap, int32_t *triples3) {
    if (!w || cap < 0 || (cap > 0 && !triples3)) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) return 0;
    { int r = settle(w); if (r != RP_OK) return r; }
    int top = 0;
    HIPCHK(w, hipMemcpy(&top, w->dw.flags + FL_POOL_TOP, sizeof(int), hipMemcpyDeviceToHost));
    top = std::min(top, w->dw.pool_cap);
    std::vector<int> c1(std::max(top, 1)), c2(std::max(top, 1)), pf(std::max(top, 1));
    if (top > 0) {
        HIPCHK(w, hipMemcpy(c1.data(), w->dw.p_c1, top * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(c2.data(), w->dw.p_c2, top * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(pf.data(), w->dw.p_pflags, top * sizeof(int), hipMemcpyDeviceToHost));
    }
    int m = 0;
    for (int s = 0; s < top; ++s) {
        if (c1[s] < 0 || !(w->colliders[c1[s]].sensor || w->colliders[c2[s]].sensor)) continue;
        if (m < cap) { triples3[3 * m] = c1[s]; triples3[3 * m + 1] = c2[s]; triples3[3 * m + 2] = (pf[s] & RP_PF_INTERSECTING) ? 1 : 0; }
        m++;
    }
    return m;
}
extern "C" int32_t rp_contact_force_events_read(rp_world *w, int32_t cap, rp_contact_force_event *out) {
    if (!w || cap < 0 || (cap > 0 && !out)) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) return 0;
    int n = 0; { int r = drain_count(w, FL_EV_FORCE, &n); if (r != RP_OK) return r; }
    int stored = std::min(n, w->dw.ev_cap);
    if (!out) return stored;
    if (n > stored) w->err = "rp_contact_force_events_read: the contact force event queue overflowed; the newest events were dropped";
    std::vector<int4> meta(stored); std::vector<float4> a(stored), b(stored); std::vector<int> order(stored);
    if (stored) {
        HIPCHK(w, hipMemcpy(meta.data(), w->dw.ev_force_meta, stored * sizeof(int4), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(a.data(), w->dw.ev_force_a, stored * sizeof(float4), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(b.data(), w->dw.ev_force_b, stored * sizeof(float4), hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < stored; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int p, int q) { const int4 &x = meta[p], &y = meta[q]; if (x.z != y.z) return x.z < y.z; if (x.x != y.x) return x.x < y.x; return x.y < y.y; });
    const int written = std::min(stored, cap);
    for (int i = 0; i < written; ++i) {
        int k = order[i];
        out[i].collider1 = meta[k].x; out[i].collider2 = meta[k].y; out[i].step = meta[k].z; out[i].started = meta[k].w;
        out[i].total_force[0] = a[k].x; out[i].total_force[1] = a[k].y; out[i].total_force[2] = a[k].z; out[i].total_force_magnitude = a[k].w;
        out[i].max_force_direction[0] = b[k].x; out[i].max_force_direction[1] = b[k].y; out[i].max_force_direction[2] = b[k].z; out[i].max_force_magnitude = b[k].w;
    }
    const int rest = stored - written; // the events not handed out stay queued, oldest first
    if (rest > 0) {
        std::vector<int4> m2(rest); std::vector<float4> a2(rest), b2(rest);
        for (int i = 0; i < rest; ++i) { int k = order[written + i]; m2[i] = meta[k]; a2[i] = a[k]; b2[i] = b[k]; }
        HIPCHK(w, hipMemcpy(w->dw.ev_force_meta, m2.data(), rest * sizeof(int4), hipMemcpyHostToDevice));
        HIPCHK(w, hipMemcpy(w->dw.ev_force_a, a2.data(), rest * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(w, hipMemcpy(w->dw.ev_force_b, b2.data(), rest * sizeof(float4), hipMemcpyHostToDevice));
    }
    HIPCHK(w, hipMemcpy(w->dw.flags + FL_EV_FORCE, &rest, sizeof(int), hipMemcpyHostToDevice));
    return written;
}

// Quarantine (quarantine.rs:68-131): bodies whose state went non-finite.  The device rolls such a body
// back to its last valid pose, stops it and keeps it in the simulation (the reference disables it);
// this returns the handles that were ever flagged.  Returns the count (may exceed cap).
extern "C" int32_t rp_quarantine_read(rp_world *w, int32_t cap, uint64_t *handles_out) {
    if (!w) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) return 0;
    { int r = settle(w); if (r != RP_OK) return r; }
    int nb = w->dw.n_bodies, m = 0;
    std::vector<int> q(std::max(nb, 1));
    if (nb > 0) HIPCHK(w, hipMemcpy(q.data(), w->dw.b_quar, nb * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < nb; ++i) if (q[i]) { if (handles_out && m < cap) handles_out[m] = ((uint64_t)w->body_gen[(size_t)i] << 32) | (uint64_t)(uint32_t)i; m++; }
    return m;
}

extern "C" int32_t rp_contacts_read(rp_world *w, int32_t cap, int32_t *meta, float *normal3, float *impulse4) {
    if (!w) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) return 0;
    { int r = settle(w); if (r != RP_OK) return r; }
    int fl[FL_COUNT];
    HIPCHK(w, hipMemcpy(fl, w->dw.flags, sizeof(fl), hipMemcpyDeviceToHost));
    int top = std::min(fl[FL_POOL_TOP], w->dw.pool_cap);
    size_t P = (size_t)w->dw.pool_cap;
    std::vector<int> c1(top), c2(top), col(top), nsc(top);
    std::vector<float4> nrm(top), imp(RP_MAX_PTS * (size_t)top), a2(4 * (size_t)top);
    if (top > 0) {
        HIPCHK(w, hipMemcpy(c1.data(), w->dw.p_c1, top * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(c2.data(), w->dw.p_c2, top * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(col.data(), w->dw.p_color, top * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(nsc.data(), w->dw.p_nsc, top * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(nrm.data(), w->dw.p_normal, top * sizeof(float4), hipMemcpyDeviceToHost));
        for (int k = 0; k < RP_MAX_PTS; ++k) HIPCHK(w, hipMemcpy(imp.data() + (size_t)k * top, w->dw.pt_imp + k * P, top * sizeof(float4), hipMemcpyDeviceToHost));
        for (int k = 0; k < 4; ++k) HIPCHK(w, hipMemcpy(a2.data() + (size_t)k * top, w->dw.sc_a2 + k * P, top * sizeof(float4), hipMemcpyDeviceToHost));
    }
    int m = 0;
    for (int s = 0; s < top; ++s) {
        if (c1[s] < 0 || nsc[s] == 0) continue;
        if (m < cap) {
            if (meta) { meta[4 * m] = c1[s]; meta[4 * m + 1] = c2[s]; meta[4 * m + 2] = col[s]; meta[4 * m + 3] = nsc[s]; }
            if (normal3) { normal3[3 * m] = nrm[s].x; normal3[3 * m + 1] = nrm[s].y; normal3[3 * m + 2] = nrm[s].z; }
            if (impulse4) for (int k = 0; k < 4; ++k) {
                float v = 0.0f;
                if (k < nsc[s]) { int cid; float f = a2[(size_t)k * top + s].w; memcpy(&cid, &f, 4); v = imp[(size_t)cid * top + s].x; }
                impulse4[4 * m + k] = v;
            }
        }
        m++;
    }
    return m;
}

// ImpulseJoint::impulses + the persistent solver colour, for n joint handles (NULL = all joints in
// insertion order).  Joints between two non-dynamic bodies are never solved: colour 255, zero impulse.
extern "C" int32_t rp_impulse_joints_read(rp_world *w, int32_t n, const uint64_t *handles, int32_t *color_out, float *impulse3_out) {
    if (!w) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) { int r = finalize(w); if (r != RP_OK) return r; }
    { int r = settle(w); if (r != RP_OK) return r; }
    int nj = w->dw.n_joints, total = (int)w->joints.size();
    std::vector<int> col(std::max(nj, 1)); std::vector<float4> imp(std::max(nj, 1));
    if (nj > 0) {
        HIPCHK(w, hipMemcpy(col.data(), w->dw.j_color, nj * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(imp.data(), w->dw.j_imp, nj * sizeof(float4), hipMemcpyDeviceToHost));
    }
    std::vector<int> dev_of(total, -1);
    for (int k = 0; k < nj; ++k) dev_of[w->active_joint_ids[k]] = k;
    int count = handles ? n : total;
    for (int i = 0; i < count; ++i) {
        int j = handles ? handle_index(handles[i]) : i;
        if (j < 0 || j >= total) { w->err = "rp_impulse_joints_read: invalid handle"; return RP_ERR_INVALID; }
        int k = dev_of[j];
        if (color_out) color_out[i] = k >= 0 ? col[k] : 255;
        if (impulse3_out) { impulse3_out[3 * i] = k >= 0 ? imp[k].x : 0.0f; impulse3_out[3 * i + 1] = k >= 0 ? imp[k].y : 0.0f; impulse3_out[3 * i + 2] = k >= 0 ? imp[k].z : 0.0f; }
    }
    return RP_OK;
}

// GenericJoint::set_motor* on ImpulseJointSet::get_mut(handle, true): the host descriptor and (once the world is resident) the
// device planes of the axis change, the motor axis is enabled, both bodies are woken.
extern "C" int32_t rp_impulse_joints_set_motor(rp_world *w, int32_t n, const uint64_t *handles, const int32_t *axes, const rp_joint_motor *motors) {
    if (!w || n < 0 || (n > 0 && (!handles || !axes || !motors))) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    for (int i = 0; i < n; ++i) {
        int j = handle_index(handles[i]);
        if (j < 0 || j >= (int)w->joints.size() || w->joint_removed[j]) { w->err = "rp_impulse_joints_set_motor: invalid handle"; return RP_ERR_INVALID; }
        if (axes[i] < 0 || axes[i] >= 6) { w->err = "rp_impulse_joints_set_motor: axis must be 0..5 (LinX..AngZ)"; return RP_ERR_INVALID; }
        if (motors[i].model != RP_MOTOR_ACCELERATION_BASED && motors[i].model != RP_MOTOR_FORCE_BASED) { w->err = "rp_impulse_joints_set_motor: unknown motor model"; return RP_ERR_INVALID; }
    }
    if (w->finalized && n > 0) { int r = settle(w); if (r != RP_OK) return r; }
    for (int i = 0; i < n; ++i) {
        int j = handle_index(handles[i]), a = axes[i];
        rp_joint_desc &jd = w->joints[j];
        jd.motor_axes |= 1u << a;
        jd.motors[a] = motors[i];
        if (!w->finalized) { w->pending_wake.push_back(jd.body1); w->pending_wake.push_back(jd.body2); continue; }
        int nj = w->dw.n_joints, r;
        for (int k = 0; k < nj; ++k) {
            if (w->active_joint_ids[k] != j) continue;
            const rp_joint_motor &m = motors[i];
            if ((r = poke(w, w->dw.j_motor + k, (int)(jd.motor_axes & 0x3fu))) != RP_OK ||
                (r = poke(w, w->dw.j_mot + (size_t)(2 * a) * nj + k, mk4(m.target_vel, m.target_pos, m.stiffness, m.damping))) != RP_OK ||
                (r = poke(w, w->dw.j_mot + (size_t)(2 * a + 1) * nj + k, mk4(m.max_force, (float)m.model, 0, 0))) != RP_OK) return r;
        }
        for (int b : {jd.body1, jd.body2}) {
            if (w->bodies[b].d.body_type == RP_BODY_FIXED || w->bodies[b].removed) continue;
            if (w->dw.sleep_enabled && (r = queue_wake(w, b, 2)) != RP_OK) return r;
        }
    }
    return RP_OK;
}
extern "C" int32_t rp_impulse_joints_read_motor_impulses(rp_world *w, int32_t n, const uint64_t *handles, float *impulse6_out) {
    if (!w || !impulse6_out) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) { int r = finalize(w); if (r != RP_OK) return r; }
    { int r = settle(w); if (r != RP_OK) return r; }
    int nj = w->dw.n_joints, total = (int)w->joints.size();
    std::vector<float4> lin(std::max(nj, 1)), ang(std::max(nj, 1));
    if (nj > 0) {
        HIPCHK(w, hipMemcpy(lin.data(), w->dw.j_imp_mot, nj * sizeof(float4), hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(ang.data(), w->dw.j_imp_mot_ang, nj * sizeof(float4), hipMemcpyDeviceToHost));
    }
    std::vector<int> dev_of(total, -1);
    for (int k = 0; k < nj; ++k) dev_of[w->active_joint_ids[k]] = k;
    int count = handles ? n : total;
    for (int i = 0; i < count; ++i) {
        int j = handles ? handle_index(handles[i]) : i;
        if (j < 0 || j >= total) { w->err = "rp_impulse_joints_read_motor_impulses: invalid handle"; return RP_ERR_INVALID; }
        int k = dev_of[j];
        float *o = impulse6_out + 6 * i;
        o[0] = k >= 0 ? lin[k].x : 0.0f; o[1] = k >= 0 ? lin[k].y : 0.0f; o[2] = k >= 0 ? lin[k].z : 0.0f;
        o[3] = k >= 0 ? ang[k].x : 0.0f; o[4] = k >= 0 ? ang[k].y : 0.0f; o[5] = k >= 0 ? ang[k].z : 0.0f;
    }
    return RP_OK;
}

extern "C" int32_t rp_counters_enable(rp_world *w, int32_t enable) {
    if (!w) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    if (enable && !w->ev[0]) for (auto &e : w->ev) HIPCHK(w, hipEventCreate(&e));
    if ((enable != 0) != w->timers) { int r = settle(w); if (r != RP_OK) return r; destroy_graphs(w); }
    w->timers = enable != 0;
    w->acc_isl_ms = w->acc_glob_ms = w->acc_col_ms = w->acc_step_ms = 0.0; w->acc_steps = 0;
    w->acc_bp_ms = w->acc_np_ms = w->acc_islc_ms = 0.0; w->acc_full_steps = 0;
    w->loop_ms_since_read = 0.0; w->loop_steps_since_read = 0;
    return RP_OK;
}

extern "C" int32_t rp_counters_read(rp_world *w, rp_counters *out) {
    if (!w || !out) return RP_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->finalized) return RP_OK;
    { int r = settle(w); if (r != RP_OK) return r; }
    int fl[FL_COUNT];
    HIPCHK(w, hipMemcpy(fl, w->dw.flags, sizeof(fl), hipMemcpyDeviceToHost));
    double n = w->acc_steps > 0 ? (double)w->acc_steps : 1.0;
    out->step_time_ms = (float)(w->acc_step_ms / n);
    out->collision_detection_ms = (float)(w->acc_col_ms / n);
    if (w->acc_full_steps > 0) { // averages over the timed FULL steps (the fast paths have no separate broad / narrow phase)
        double nf = (double)w->acc_full_steps;
        out->broad_phase_ms = (float)(w->acc_bp_ms / nf); out->narrow_phase_ms = (float)(w->acc_np_ms / nf); out->island_construction_ms = (float)(w->acc_islc_ms / nf);
    }
    out->solver_ms = (float)((w->acc_isl_ms + w->acc_glob_ms) / n);
    out->velocity_assembly_ms = 0.0f; // assembly is fused into the solve kernels
    out->velocity_resolution_ms = (float)(w->acc_isl_ms / n);  // k_island_solve (LDS-resident islands)
    out->velocity_update_ms = (float)(w->acc_glob_ms / n);     // global path (islands too large for LDS, free bodies)
    int live = 0;
    {
        int top = std::min(fl[FL_POOL_TOP], w->dw.pool_cap);
        live = top - fl[FL_FREE_TOP];
        if (w->dw.has_composite && top > 0) { // (clusters of composite pairs hold pool slots but are not pairs)
            std::vector<int> pc1((size_t)top), pfl((size_t)top);
            HIPCHK(w, hipMemcpy(pc1.data(), w->dw.p_c1, (size_t)top * sizeof(int), hipMemcpyDeviceToHost)); HIPCHK(w, hipMemcpy(pfl.data(), w->dw.p_pflags, (size_t)top * sizeof(int), hipMemcpyDeviceToHost));
            for (int q = 0; q < top; ++q) if (pc1[(size_t)q] >= 0 && (pfl[(size_t)q] & RP_PF_AUX)) --live;
        }
    }
    out->num_pairs = live;
    out->num_manifolds = fl[FL_N_CONS_ALL];
    out->num_solver_contacts = fl[FL_N_SC];
    out->num_colors = fl[FL_N_COLORS];
    out->num_parallel_stages = fl[FL_N_PARALLEL];
    int nd = 0; for (auto &b : w->bodies) nd += b.d.body_type == RP_BODY_DYNAMIC;
    out->num_dynamic_bodies = nd;
    out->bp_rebuilds = fl[FL_BP_REBUILDS];
    out->full_updates = fl[FL_FULL_UPDATES];
    out->overflow_flags = fl[FL_OVERFLOW];
    out->quarantined = fl[FL_QUARANTINE];
    out->ccd_active_count = fl[FL_CCD_ACTIVE]; out->ccd_clamp_count = fl[FL_CCD_CLAMPS];
    out->num_tiles = fl[FL_N_TILES]; out->tile_sweeps = (w->graph_tile_grid > 0 && !w->plan_single) ? 1 : 0; out->lean_steps = (int32_t)w->lean_steps; out->bp_large_list = fl[FL_N_LARGE];
    out->fast_steps = (int32_t)w->fast_steps; out->full_steps = (int32_t)w->full_steps; out->replayed_steps = (int32_t)w->replayed_steps; out->fused_steps = (int32_t)w->fused_steps;
    if (w->dw.sleep_enabled && w->dw.n_bodies > 0) {
        std::vector<int> bfl(w->dw.n_bodies);
        HIPCHK(w, hipMemcpy(bfl.data(), w->dw.b_flags, bfl.size() * sizeof(int), hipMemcpyDeviceToHost));
        int ns = 0; for (int f : bfl) ns += (f & RP_BF_TYPE_MASK) != RP_BODY_FIXED && (f & RP_BF_SLEEPING);
        out->num_sleeping_bodies = ns;
    }
    return RP_OK;
}

// Debug aid (not part of include/rapier_hip.h): cycle stamps written by k_island_solve for island 0
// when the library is built with -DRP_ISL_PROFILE.
extern "C" int32_t rp_debug_cycles(rp_world *w, long long *out64) {
    if (!w || !w->finalized || !out64) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    HIPCHK(w, hipMemcpy(out64, w->dw.dbg, 64 * sizeof(long long), hipMemcpyDeviceToHost));
    return RP_OK;
}

// debug aid: raw slice of the device debug counters (slots 64.. hold the event timeline of one traced body, rp_flow.hip)
extern "C" int32_t rp_debug_read(rp_world *w, int32_t offset, int32_t n, long long *out) {
    if (!w || !w->finalized || !out || offset < 0 || n < 0 || offset + n > 1024) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    HIPCHK(w, hipMemcpy(out, w->dw.dbg + offset, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
    return RP_OK;
}

extern "C" int32_t rp_solver_loop_time_ms(rp_world *w, float *avg, int32_t *steps) {
    if (!w) return RP_ERR_INVALID;
    if (avg) *avg = w->loop_steps_since_read > 0 ? (float)(w->loop_ms_since_read / w->loop_steps_since_read) : 0.0f;
    if (steps) *steps = w->loop_steps_since_read;
    w->loop_ms_since_read = 0.0; w->loop_steps_since_read = 0;
    return RP_OK;
}

// debug aid (not in the header; tools/composite_diag.py): the solver manifolds of pair (c1, c2) as the device holds them.
// out: [0] = cluster count (p_aux.w), [1..2] = p_sub, then per solver manifold k < max(1, count): slot, npts, nsc, then npts x (lp1.xyz, dist, impulse, warmstart_impulse)
extern "C" int32_t rp_debug_pair_points(rp_world *w, int32_t c1, int32_t c2, int32_t cap, float *out) {
    if (!w || !w->finalized) return RP_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    { int r = settle(w); if (r != RP_OK) return r; }
    int fl[FL_COUNT]; HIPCHK(w, hipMemcpy(fl, w->dw.flags, sizeof(fl), hipMemcpyDeviceToHost));
    const int top = std::min(fl[FL_POOL_TOP], w->dw.pool_cap); const size_t P = (size_t)w->dw.pool_cap;
    std::vector<int> pc1((size_t)top), pc2((size_t)top), pfl((size_t)top);
    HIPCHK(w, hipMemcpy(pc1.data(), w->dw.p_c1, (size_t)top * sizeof(int), hipMemcpyDeviceToHost)); HIPCHK(w, hipMemcpy(pc2.data(), w->dw.p_c2, (size_t)top * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(w, hipMemcpy(pfl.data(), w->dw.p_pflags, (size_t)top * sizeof(int), hipMemcpyDeviceToHost));
    int s = -1;
    for (int q = 0; q < top; ++q) if (pc1[(size_t)q] == c1 && pc2[(size_t)q] == c2 && !(pfl[(size_t)q] & RP_PF_AUX)) { s = q; break; }
    if (s < 0) return 0;
    int4 aux; int2 sub;
    HIPCHK(w, hipMemcpy(&aux, w->dw.p_aux + s, sizeof(int4), hipMemcpyDeviceToHost)); HIPCHK(w, hipMemcpy(&sub, w->dw.p_sub + s, sizeof(int2), hipMemcpyDeviceToHost));
    int n = 0;
    auto put = [&](float v) { if (n < cap) out[n] = v; ++n; };
    put((float)aux.w); put((float)sub.x); put((float)sub.y);
    const int nsm = aux.w > 1 ? aux.w : 1;
    for (int k = 0; k < nsm; ++k) {
        const int slot = k == 0 ? s : (k == 1 ? aux.x : (k == 2 ? aux.y : aux.z));
        int npts = 0, nsc = 0;
        if (slot >= 0) { HIPCHK(w, hipMemcpy(&npts, w->dw.p_npts + slot, sizeof(int), hipMemcpyDeviceToHost)); HIPCHK(w, hipMemcpy(&nsc, w->dw.p_nsc + slot, sizeof(int), hipMemcpyDeviceToHost)); }
        put((float)slot); put((float)npts); put((float)nsc);
        for (int i = 0; i < npts; ++i) {
            float4 a, im;
            HIPCHK(w, hipMemcpy(&a, w->dw.pt_lp1d + (size_t)i * P + slot, sizeof(float4), hipMemcpyDeviceToHost)); HIPCHK(w, hipMemcpy(&im, w->dw.pt_imp + (size_t)i * P + slot, sizeof(float4), hipMemcpyDeviceToHost));
            put(a.x); put(a.y); put(a.z); put(a.w); put(im.x); put(im.y);
        }
    }
    return n;
}
