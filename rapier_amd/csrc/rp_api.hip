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
void rp_launch_island_solve_steps(const DevWorld &w, hipStream_t st, int grid, int has_restitution, int nsteps, int dense);
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
int rp_joint_net_cap(void); // rp_tiles.hip
int rp_tile_step_cap(void);  // rp_tiles.hip

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
    hipStream_t stream2 = nullptr;           // the solver branch of a joint-net lean step (enqueue_whole): runs beside the collision stage
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
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
    std::vector<rp_joint_desc> joints; // the edge list in insertion order (removed joints stay as tombstones); body1 / body2 hold the resolved ARENA INDICES
    // ImpulseJointSet's handle arena (joint_ids: Arena<..>, impulse_joint_set.rs:48; data/arena.rs:260-380): slot -> position in `joints`
    // (-1: free), LIFO free list, generation = the arena's removal count when the slot was handed out; joint_slot: position -> slot
    std::vector<int> jslot_pos, jslot_free, joint_slot; std::vector<uint32_t> jslot_gen; uint32_t joint_arena_gen = 0;
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
    // rp_world_pack_bodies / rp_shard_all_gather (rp_api_comm.inc): global body ids (host, per arena row; empty = the arena index), their
    // device copy + the free-slot mask, the packed rows of this shard, the gathered rows of every rank
    std::vector<int64_t> global_ids; bool global_ids_dirty = true;
    void *d_gid = nullptr, *d_skip = nullptr, *d_pack = nullptr, *d_gather = nullptr; int *d_pack_count = nullptr;
    int pack_cap = 0, pack_tables_rows = -1; uint32_t pack_tables_edit = 0; size_t gather_cap = 0;
    void *d_puts = nullptr;      // PutBatch's record buffer (rp_api_device.inc)
    float guard_horizon = 0.0f;  // rp_world_set_shard_guard_horizon
    // shard guard (rp_world_set_shard_guard): host copy, re-uploaded whenever the device world is rebuilt
    std::vector<float4> guard_min, guard_max; std::vector<int> guard_start, guard_items; float guard_origin[3] = {0, 0, 0}, guard_cell = 0.0f; int guard_dims[3] = {0, 0, 0};
    // launch plan + graph
    int plan_stages = 0, plan_blocks = 1, plan_single = 1, plan_island_grid = 1, plan_joint_stages = 0, plan_no_global = 0, plan_fused = 0, plan_tile_grid = 0, plan_no_contacts = 0, plan_bare = 0, plan_jn = 0, plan_ts = 0, plan_dense = 0, plan_wide = 0, plan_islands_hint = 0;
    bool has_restitution = false;
    // [0] = full path, [1] = fast path, [2] = lean path; "whole" = one graph per step, col/loop/fin = timed thirds (full / fast only)
    hipGraph_t g_whole[3] = {nullptr, nullptr, nullptr}, g_col[3] = {nullptr, nullptr, nullptr}, g_loop[3] = {nullptr, nullptr, nullptr}, g_fin[3] = {nullptr, nullptr, nullptr};
    hipGraphExec_t ge_whole[3] = {nullptr, nullptr, nullptr}, ge_col[3] = {nullptr, nullptr, nullptr}, ge_loop[3] = {nullptr, nullptr, nullptr}, ge_fin[3] = {nullptr, nullptr, nullptr};
    int graph_stages = -1, graph_blocks = -1, graph_single = -1, graph_island_grid = -1, graph_dense = -1, graph_wide = -1, graph_joint_stages = -1, graph_no_global = -1, graph_fused = -1, graph_tile_grid = -1, graph_no_contacts = -1, graph_bare = -1, graph_jn = -1, graph_ts = -1;
    bool use_graph = true, use_fast = true, use_fused = true;
    long long last_periodic_settle = 0;
    bool dense_seen = false;         // the register-lean island kernel was planned at least once (step_once: such a world starts wary)
    bool abort_seen = false;         // a fast step aborted and fewer than 1,024 clean fast steps went by since (step_once)
    long long clean_fast_steps = 40; // fast steps enqueued since the host last saw an aborted one (step_once: how many steps a launch may carry); a world that has
                                     // not aborted yet counts as fairly clean (40: a 20-step call of a settled world is ONE launch, not a ramp of seven)
    bool replaying = false;         // settle() is replaying steps that were requested before: step_once must not count them again
    bool use_multi = true; int cur_multi = 1; long long multi_launches = 0, fused_launches = 0, jn_steps = 0, ts_steps = 0; // launches of several fused steps (k_island_solve_steps): allowed / steps of the launch being enqueued
    bool auto_dense = true;        // RP_ISL_DENSE=0: never
    bool force_dense = false;      // RP_ISL_DENSE=1 (tests): the dense form of k_island_solve whatever the island count
    bool use_jn = true;            // the joint-net form of a bare lean graph (k_joint_net_step, rp_tiles.hip); RP_NO_JOINT_NET=1: never
    bool use_ts = true;            // the one-launch form of a tiled contact world's lean graph (k_tile_step, rp_tiles.hip); RP_NO_TILE_STEP=1: never
    bool use_lean = true;          // the lean step graph of MULTI-mode worlds (below: "lean graph"); RP_NO_LEAN=1: never
    int cur_lean = 0;              // the enqueue_* callbacks capture / launch the lean graph (with dw_lean)
    bool ts_world_ok = false;      // step_once: nothing in this world (sleeping, events, sensors, ...) that a step dying behind its solver launch would have touched
    int cur_ts = 0;                // ... a FULL graph whose TGS loop is the one-launch form (k_tile_step): its solver, write-back and k_ccd launches get dw_ts
    DevWorld dw_ts;                // dw with lean = 8 | grid << 8 (no bit 0: the graph is a full one; lean_dead then only speaks about the launch itself)
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
    bool timed_ready[3] = {false, false, false};
    bool never_stepped = true;     // no step has retired and the device world was never rebuilt / grown: what the host mirrors hold is the whole state
    int pairs_scale = 1;           // the pair pool holds RP_PAIRS_PER_COLLIDER x pairs_scale slots per collider row: doubled when the pool fills up (rp_step)
    int cur_fast = 0;              // mode the enqueue_* callbacks capture
    long long steps_requested = 0; // steps asked for since finalize (device FL_STEP counts the executed ones)
    int rebase_at = 1 << 29;       // FL_STEP beyond which the device's 32-bit step stamps move back (k_rebase_stamps); the wake stamp is 2 * step + phase and
                                   // settles that look at the counter may be 2^20 steps apart: 2 * (2^29 + 2^20) stays inside an int
    long long rebases = 0, rebased_steps = 0; // (rebased_steps: what the stamps moved back so far — events are handed out with it added back)
    long long seq_enqueued = 0;    // step graphs enqueued since finalize (device FL_SEQ counts the retired ones)
    long long full_until = 0;      // stay on the full graph until this many steps were requested
    long long hints_from_seq = 0; bool hints_from_seq_valid = false; // hints published by launches before this sequence number are stale (step_once)
    long long eager_until = 0;     // launch the kernels directly until this many steps were requested: a world that is being edited (bodies /
                                   // colliders / joints coming and going every few steps) would re-capture its graphs — ~10 ms — after every edit
    long long fast_steps = 0, full_steps = 0, replayed_steps = 0, fused_steps = 0; int fused_disabled = 0, jn_disabled = 0;
    // a one-launch form lost to a workgroup that never became resident (FL_GRID_TIMEOUT / FL_JN_TIMEOUT: the GPU was shared at that moment) is
    // tried again once this many steps were requested (-1: not waiting); every loss in a row waits four times as long (4,096 steps ... 2^22)
    long long fused_retry_at = -1, jn_retry_at = -1, fused_backoff = 4096, jn_backoff = 4096; // (RP_ONE_LAUNCH_RETRY=<steps>: the first wait; 0 = a lost form stays lost)
    bool retry_jn = false, retry_ts = false; // what the joint-net / tile-step loss took away (a form switched off by the environment stays off)
    // timers
    bool timers = false;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // ([6], [7]: around the one-launch TGS loop of a timed step)
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
static int joint_of(const rp_world *w, uint64_t h, bool allow_removed = false);
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
    p->min_ccd_dt = 1.0f / 60.0f / 100.0f;
    p->contact_clustering = 1;
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
    if (!real(p.min_ccd_dt) || p.min_ccd_dt < 0.0f) return "min_ccd_dt must be finite and >= 0";
    if (p.contact_clustering != 0 && p.contact_clustering != 1) return "contact_clustering is a bool (0 | 1)";
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
    { const char *m = getenv("RP_NO_MULTI_STEP"); if (m && m[0] == '1') w->use_multi = false; }
    g = getenv("RP_NO_LEAN");
    if (g && g[0] == '1') w->use_lean = false;
    g = getenv("RP_NO_JOINT_NET");
    if (g && g[0] == '1') w->use_jn = false;
    g = getenv("RP_NO_TILE_STEP");
    if (g && g[0] == '1') w->use_ts = false;
    g = getenv("RP_NO_JOINT_NET_FORK");
    if (!(g && g[0] == '1') && w->use_jn) { // (the fork is an optimisation: a world without it runs the lean step in one line)
        if (hipStreamCreateWithFlags(&w->stream2, hipStreamNonBlocking) != hipSuccess) w->stream2 = nullptr;
        if (w->stream2 && (hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming) != hipSuccess)) { hipStreamDestroy(w->stream2); w->stream2 = nullptr; }
    }
    g = getenv("RP_NO_FLOW");
    if (g && g[0] == '1') w->use_flow = false;
    g = getenv("RP_FLOW");
    if (g && g[0] == '1') w->force_flow = true;
    if (w->use_flow) { w->flow_grid = rp_flow_grid(device); if (w->flow_grid <= 0) w->use_flow = false; }
    if (getenv("RP_LAUNCH_RAMP")) w->clean_fast_steps = 0;
    if (w->use_fused) { w->fused_grid = rp_fused_grid(device); if (w->fused_grid <= 0) w->use_fused = false; }
    { const char *rs = getenv("RP_ONE_LAUNCH_RETRY"); if (rs) w->fused_backoff = w->jn_backoff = std::max(0ll, atoll(rs)); }
#ifdef RP_TESTING
    { extern int rp_test_ts_stall_tile; const char *js = getenv("RP_TEST_TS_STALL"); rp_test_ts_stall_tile = js ? atoi(js) : -1; } // (... of k_tile_step)
    { extern int rp_test_jn_stall_tile; const char *js = getenv("RP_TEST_JN_STALL"); rp_test_jn_stall_tile = js ? atoi(js) : -1; } // (test hook: that workgroup of k_joint_net_step never arrives)
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
    w->timed_ready[0] = w->timed_ready[1] = w->timed_ready[2] = false; w->graph_jn = -1; w->graph_ts = -1;
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
    for (void **b : {&w->d_gid, &w->d_skip, &w->d_pack, &w->d_gather, (void **)&w->d_pack_count}) if (*b) { hipFree(*b); *b = nullptr; }
    if (w->d_puts) { hipFree(w->d_puts); w->d_puts = nullptr; }
    for (auto &e : w->ev) if (e) hipEventDestroy(e);
    if (w->ev_fork) hipEventDestroy(w->ev_fork);
    if (w->ev_join) hipEventDestroy(w->ev_join);
    if (w->stream2) hipStreamDestroy(w->stream2);
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
    if (!in->contact_clustering && !w->comps.empty()) { w->err = "rp_params_set: contact_clustering = false in a world that holds composite shapes (their unclustered form is not built)"; return RP_ERR_INVALID; }
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

#include "rp_api_host.inc"
#include "rp_api_insert.inc"
#include "rp_api_device.inc"
#include "rp_api_step.inc"
#include "rp_api_access.inc"
#include "rp_api_edits.inc"
#include "rp_api_readback.inc"
#include "rp_api_comm.inc"
