// rp_world.h — HBM data layout of one physics world (Structure-of-Arrays) shared by every kernel.
//
// Everything a step touches lives in device memory; kernels receive a `DevWorld` by value (plain
// pointers + sizes).  Layout choices:
//   * bodies / colliders: one float4 array per attribute, indexed by arena index (coalesced per-item kernels);
//   * contact pairs: pool slot s owns column s of every pair array; manifold points are
//     [point k][slot] planes so a wavefront reading point k of 64 consecutive slots is coalesced;
//   * solver constraints: float4 planes C[plane][position], position = rank of the manifold in the
//     colour-major (stage-major) order, so one colour stage reads contiguous columns.
#pragma once
#include "rp_math.h"
#include "../../include/rapier_hip.h"

#define RP_NUM_COLORS 129
#define RP_COLOR_OVERFLOW 128
#define RP_COLOR_UNCOLORED 255
#define RP_DYNAMIC_COLOR_COUNT 120   // contact_pair.rs:167
#define RP_MAX_PTS 8                 // manifold points kept per pair (face/face clip max)
#define RP_PARALLEL_MIN_MANIFOLDS 125 // ceil(n/4) >= 32 chunks  (init.rs:169, mod.rs:41-57)
#define RP_EMPTY_KEY 0xffffffffffffffffull
#define RP_TOMB_KEY 0xfffffffffffffffeull   // a pair deleted by an incremental broad-phase pass (probing continues past it)
#define RP_BP_MOVED_CAP 2048                // stale colliders an incremental pass may leave behind before a full rebuild is due
#define RP_ISL_NB_MAX 64             // bodies per LDS-resident island
#define RP_ISL_NC_MAX 160            // solver manifolds per LDS-resident island
#define RP_FID_UNKNOWN 0xffffu
#define RP_TILE_BCAP 1024            // cone bodies (owned + halo) of one LDS tile (rp_tiles.hip)
#define RP_TILE_CCAP 3072            // cone constraints of one tile
#define RP_JN_THREADS 512            // threads of k_joint_net_step: one cone joint each (rp_tiles.hip)
#define RP_TILE_STAGES 127           // sweep stages a tiling handles (every colour but the overflow one)
#define RP_TILE_CELLS 4096           // cells of the Morton counting sort that orders bodies into tiles (12 bits)
#define RP_TILE_MIN_BODIES 1024      // global-path bodies below which the colour stages stay launches

// body flag bits
#define RP_BF_TYPE_MASK 0x3
#define RP_BF_GYRO 0x4
#define RP_BF_FASTROT 0x8
#define RP_BF_SLEEPING 0x10          // RigidBodyActivation::sleeping (dynamic bodies only)
#define RP_BF_CCD_ENABLED 0x20       // RigidBodyCcd::ccd_enabled: a bullet (dynamics/ccd/sweeps.rs:29-31)
#define RP_BF_DOM_SHIFT 8
#define RP_BF_LOCK_SHIFT 16           // LockedAxes (6 bits): translation x,y,z then rotation x,y,z

// pair flag bits
#define RP_PF_AUX 0x10                 // an auxiliary slot: solver manifold (cluster) 2+ of a composite pair (p_aux)
#define RP_PF_RECYCLE 0x1
#define RP_PF_FORCE_EMITTED 0x2
#define RP_PF_INTERSECTING 0x8       // IntersectionPair::intersecting of a sensor pair (narrow_phase/intersections.rs)
#define RP_EVENTS_SENSOR_BIT 0x100    // internal: the collider is a sensor (kept next to the ActiveEvents bits in c_events.x)
#define RP_PF_NO_CONTACT 0x4          // cleared by a joint with contacts_enabled = false (pair_update.rs:191-201); skipped until the joint set changes       // PairEventStatus::INITIAL_FORCE_THRESHOLD_EVENT_EMITTED

// overflow / error flag bits (dev flags[FL_OVERFLOW])
#define RP_OVF_POOL 0x1
#define RP_OVF_HASH 0x2
#define RP_OVF_CELLS 0x4 // (unused since the grid has fixed-slot buckets: a full bucket sends the collider to the large list)
#define RP_BP_BUCKET 32 // slots per hash bucket of the broad-phase grid (rp_broadphase.hip)
#define RP_OVF_LARGE 0x8
#define RP_OVF_CONS 0x10
#define RP_OVF_GRID 0x20   // a workgroup of a fused rebuild kernel was not resident (rp_gridbar.h: its grid barrier timed out)
#define RP_OVF_FLOW 0x40   // the dataflow solver (rp_flow.hip) gave up waiting for a body record (its grid was not fully resident)
#define RP_OVF_SHARD 0x80  // a collider of this shard moved into a cell that holds another shard's bodies (rp_world_set_shard_guard)

// device scalar slots (int32) in DevWorld::flags
enum {
    FL_BP_DIRTY = 0,    // some fat AABB changed -> pair set must be rebuilt this step
    FL_BP_EPOCH,        // rebuild counter; parity selects the live hash table
    FL_N_LARGE,         // colliders spanning > 3 grid cells on an axis
    FL_N_ENTRIES,       // grid cell entries
    FL_POOL_TOP,        // bump allocator of pair slots
    FL_FREE_TOP,        // free-slot stack height
    FL_TODO_COUNT,      // begin-touch pairs waiting for a colour
    FL_LAYOUT_DIRTY,    // the active-manifold set changed -> rebuild buckets
    FL_N_CONS,          // M: active solver manifolds
    FL_N_STAGES,        // colour stages (non-empty colours, overflow excluded)
    FL_N_PARALLEL,      // stages with >= RP_PARALLEL_MIN_MANIFOLDS
    FL_MAX_STAGE,       // largest stage size
    FL_OVERFLOW,        // RP_OVF_* bits
    FL_FULL_UPDATES,    // narrow-phase full updates this step
    FL_N_SC,            // solver contacts
    FL_QUARANTINE,      // non-finite poses detected
    FL_ANY_BOUNCY,      // some constraint holds a restitution seed
    FL_HAS_OVERFLOW_COLOR, // overflow bucket non-empty
    FL_N_COLORS,
    FL_BP_REBUILDS,
    FL_STEP,            // step counter
    FL_N_ISLANDS,       // small islands solved by the LDS island kernel
    FL_N_GLOB_BODIES,   // dynamic bodies outside small islands (global path)
    FL_N_CONS_ALL,      // M: all active solver manifolds (island + global)
    FL_ISL_BODY_CURSOR, FL_ISL_CONS_CURSOR,
    FL_SEQ,             // step graphs retired (executed or aborted)
    FL_JOINT_DIRTY,     // joint colours / stage layout must be rebuilt
    FL_NJ_STAGES, FL_NJ_OVF_BEGIN, FL_NJ_OVF_COUNT, // joint stage layout: parallel colours, serial overflow range in j_order
    FL_ARRIVE, FL_DEPART, // fused fast step: workgroups that validated their islands / that finished (reset by the last one)
    FL_ISL_ICONS_CURSOR,
    FL_BP_CLOSE,        // an incremental broad-phase pass waits to be closed by the next kernel (rp_pairs.h bp_close_incremental)
    FL_FAST_ABORT,      // steady-state fast path found work it cannot do (see rp_api.hip); sticky until a full step
    FL_EV_COL, FL_EV_FORCE, // events appended to the collision / contact-force queues (may exceed the queue capacity)
    FL_N_AWAKE,         // awake non-fixed bodies after the last sleep pass (0 = the whole world sleeps: idle steps, rp_sleep.hip)
    FL_WAKE_PENDING,    // the host queued wake-up requests (b_wake_req) that no step has consumed yet
    FL_NP_COUNT,        // pairs queued for a full narrow-phase update this step (np_list)
    FL_WAKE_STAMP,      // 2 * step + phase of the last wake pass that found a sleeping island to wake (rp_sleep.hip)
    FL_FLOW_DIRTY,      // the constraint / joint layout changed: the per-body toucher ranks of the dataflow solver must be rebuilt
    FL_FLOW_ABORT,      // a dataflow-solver wave timed out: every wave leaves its wait loops (rp_flow.hip)
    FL_FLOW_CURSOR, FL_FLOW_JCURSOR, // bump allocators of the per-body toucher lists (contacts, joints)
    FL_CCD_ACTIVE,      // (body, step) occurrences of the CCD fast-body criterion so far (body_writeback): the reference would have swept
    FL_GRID_TIMEOUT,    // a fused fast step gave up waiting for a workgroup that was not resident: the step was aborted (nothing written),
                        // the host replays it on the full graph and stops using the fused launch (rp_api.hip settle())
    // persistent islands (rp_sleep.hip; PersistentIslands, island_manager/persistent.rs:128-171)
    FL_PI_NEXT,         // island ids handed out so far without reuse (alloc_island: `islands.len()` once the free list is empty)
    FL_PI_NFREE,        // height of the free-id stack (free_islands)
    FL_PI_PENDING,      // split_island + 1: the island chosen last step for this step's global split, 0 = None
    FL_PJ_COUNT,        // removal_journal entries waiting for resolve_removals
    FL_PI_MERGED,       // scratch of one sleep pass: some touching pair joins two islands
    FL_PI_JLINK,        // first device joint whose ImpulseJointIslandEvent::Link is not applied yet, + 1 (0 = none)
    // incremental broad phase (rp_broadphase.hip)
    FL_BP_NCHG,         // colliders whose fat AABB was rewritten since the last broad-phase pass (bp_chg_list)
    FL_BP_NFREED,       // pair slots freed by the running incremental pass, waiting in free_pending for its epilogue (rp_broadphase.hip)
    FL_BP_GRID_OK,      // the grid describes every collider that is not on bp_moved_list (cleared by topology edits)
    FL_BP_SEQ,          // broad-phase passes run so far (stamps c_chgstamp)
    FL_BP_FORCE_FULL,   // scratch of one pass: the incremental update met a case it leaves to the full rebuild
    FL_BP_TOMBS,        // tombstones in the live pair hash table since the last full rebuild
    FL_CCD_N,           // bodies body_writeback found moving fast this step (ccd_list): the input of k_ccd
    FL_CCD_CLAMPS,      // (body, step) cases in which k_ccd clamped a pose to a time of impact
    FL_UF_NPAIRS,       // scratch of a layout rebuild: active dynamic-dynamic pairs listed for the island union-find (uf_pairs)
    FL_N_TILES,         // LDS tiles the global path's big component is cut into (rp_tiles.hip); 0 = no valid tiling: colour stages as launches
    FL_JN_TIMEOUT,      // a tile of k_joint_net_step gave up waiting for a neighbouring tile (~2 s: a workgroup of the launch was not resident): the
                        // step died like any lean step (nothing committed, resumed by the full graph); settle() takes the joint-net form from this world
    FL_TILE_JMAX,       // joints of the largest cone of the current tiling (k_tiles_cones): k_joint_net_step holds one per thread
    FL_COUNT = 72       // (publish_flags copies the slots in strides of the workgroup)
};

// constraint float4 planes (per solver manifold) — restates ContactWithTwistFriction +
// ContactWithTwistFrictionBuilder (contact_with_twist_friction.rs:18-55,600-630)
enum {
    CP_H0 = 0,  // dir1.xyz, limit (friction coefficient)
    CP_H1,      // im1.xyz, twist r
    CP_H2,      // im2.xyz, tangent r[2]
    CP_H3,      // ii1: m11 m12 m13 m22
    CP_H4,      // ii1: m23 m33 ; ii2: m11 m12
    CP_H5,      // ii2: m13 m22 m23 m33
    CP_H6,      // tangent1.xyz, tangent rhs_wo_bias[0]
    CP_H7,      // tangent rhs_wo_bias[1], tangent r[0], r[1], -
    CP_H8,      // twist_dists[0..3]
    CP_HM0,     // mutable: twist impulse, twist accumulator, tangent impulse[0], [1]
    CP_HM1,     // mutable: tangent accumulator[0],[1], tangent rhs[0],[1]
    CP_T0,      // tangent torque_dir1[0].xyz
    CP_T1,      // tangent torque_dir1[1]
    CP_T2,      // tangent torque_dir2[0]
    CP_T3,      // tangent torque_dir2[1]
    CP_T4,      // tangent ii_torque_dir1[0]
    CP_T5,      // tangent ii_torque_dir1[1]
    CP_T6,      // tangent ii_torque_dir2[0]
    CP_T7,      // tangent ii_torque_dir2[1]
    CP_B0,      // builder: local_friction_center1.xyz
    CP_B1,      // builder: local_friction_center2.xyz
    CP_B2,      // builder: tangent_vel.xyz
    CP_N0,      // first per-point plane; point k uses CP_N0 + 7*k + {NM..NF}
    CP_COUNT = CP_N0 + 7 * 4
};
// FrictionModel::Coulomb adds 9 tangent planes per point behind CP_COUNT (rp_coulomb.h)
#define CP_SHADOW_COUNT 6 // shadow copies of the mutable planes (worlds that may tile: DevWorld::c_par)
#define CQ_PER_POINT 9
#define CQ_COUNT (CP_COUNT + CQ_PER_POINT * 4)
enum {
    NP_M = 0,   // mutable: rhs, cfm_factor, impulse, impulse_accumulator
    NP_A,       // torque_dir1.xyz, r
    NP_B,       // torque_dir2.xyz, restitution_seed
    NP_C,       // ii_torque_dir1.xyz, builder dist
    NP_D,       // ii_torque_dir2.xyz, -
    NP_E,       // builder local_p1.xyz
    NP_F        // builder local_p2.xyz
};

#define RP_JR_COUNT 74 // planes of DevWorld::JR: im1, im2 + 12 rows x 6 (rp_joints.h)

struct SimParams {
    rp_integration_params p;
    float gravity[3];
    // derived per substep (host-computed in f32 exactly as SpringCoefficients does)
    float dt_sub, inv_dt_sub;
    float dyn_cfm, static_cfm, dyn_erp_inv_dt, static_erp_inv_dt;
    float prediction, recycle_distance, max_corrective_velocity, max_lin, max_ang;
    float bp_skin;
    float cell_size, inv_cell_size;
    float joint_erp_inv_dt, joint_cfm_coeff; // SpringCoefficients::{erp_inv_dt, cfm_coeff}(dt_sub) of the joint softness
    int num_substeps;
};

// the substep-dependent part of SimParams for one solve group (RigidBody::additional_solver_iterations, rp_groups.h)
#define RP_MAX_GROUPS 16
struct SubParams { float dt_sub, inv_dt_sub, dyn_cfm, static_cfm, dyn_erp_inv_dt, static_erp_inv_dt, joint_erp_inv_dt, joint_cfm_coeff; int num_substeps, extra; };

struct DevWorld {
    int n_bodies, n_colliders, n_joints;
    int n_nc;          // joints that disable the contacts between their two bodies
    int pool_cap;      // pair slots
    int hash_cap;      // power of two
    int grid_cap;      // power of two (cell hash buckets)
    int large_cap;
    int cons_cap;      // solver manifolds
    int sleep_enabled; // some body may fall asleep (can_sleep dynamic bodies, any kinematic body): the sleep kernels run and pairs carry solver hints
    int has_force_events;  // some collider has ActiveEvents::CONTACT_FORCE_EVENTS: k_force_events runs after every step
    int ev_cap;            // slots per event queue
    int has_kinematic_pos; // some body is KinematicPositionBased: k_kinematic_velocities runs
    int isl_generic;       // RP_ISL_GENERIC=1: islands through k_island_generic even under the twist model (tests of that kernel)
    int n_groups;          // distinct additional_solver_iterations counts in the world (1 = no elevated body: the plain paths)
    int bp_incremental;    // the broad phase may update incrementally (0: RP_NO_BP_INCR=1, every pass is a full rebuild)
    int bp_incr_div;       // an incremental pass serves up to n_colliders / bp_incr_div (+ 16) rewritten fat AABBs (1: A/B)
    int gbar_blocks;       // most workgroups (of 1024 threads) a grid-barrier kernel may use on this device: all of them resident at once (rp_gridbar.h)
    int has_sensors;       // some collider is a sensor: its pairs are intersection-tested every step (full step path)
    int isl_route_tiny;    // 1 (RP_NO_TINY_ROUTING=1: 0): worlds with thousands of tiny islands solve them on the global path (rp_islands.hip, lay_isl_number)
    int isl_bundle_tiny;   // 1 (RP_NO_TINY_BUNDLES=1: 0): worlds without sleeping pack the tiny islands into shared islands instead (lay_isl_number)
    int isl_tiny_nc;       // ... and "tiny" = at most this many manifolds (8)
    int isl_many;          // ... "thousands" = more island candidates than this in the previous rebuild (960; RP_ISL_MANY overrides it)
    int has_convex;        // some collider is a cylinder / cone / convex polyhedron: the narrow-phase, sensor and CCD launches use their CONVEX instantiations (rp_convex.h)
    // convex polyhedra (rp_polyhedron.h), flattened: per shape {first point, points, first face, faces}; points (w: max |p|); face normals;
    // per face {first loop entry, entries}; loop entries {vertex of the shape, edge of the shape}.  A collider's c_he.w holds its shape's row, as bits
    int4 *cv_hdr; float4 *cv_pts; float4 *cv_fn; int2 *cv_fl; int2 *cv_loop;
    // composite shapes (rp_composite.h; RP_SHAPE_COMPOUND / RP_SHAPE_TRIMESH: c_he.w = the shape's row in cm_hdr, as bits)
    int has_composite;     // some collider is a composite shape: k_np_composite runs behind k_np_update
    int4 *cm_hdr;          // per composite: kind (RP_SHAPE_COMPOUND | RP_SHAPE_TRIMESH), first sub-shape row, sub-shape count, -
    float4 *cm_min, *cm_max; // per sub-shape: its AABB in the composite's (recentred) frame
    float4 *cm_a, *cm_b, *cm_c; // compound part: a = c_he of the part (w: polyhedron row bits), b = pose translation (w: shape bits), c = pose rotation;
                           // mesh triangle: a, b, c = its vertices
    float *cm_border;      // compound part: border radius of a round part (0 otherwise)
    float4 *cm_ws; int cm_ws_threads; // k_np_composite's per-thread workspace: clusters while they are built (RP_CM_WS_F4 float4 per thread)
    int joints_spherical;  // every impulse joint locks the three linear axes and nothing else (no limit, no motor): tile sweeps may rebuild the rows themselves
    int lean;              // bit 0 (bit 1: a bare lean graph, see lean_dead) set in the copy the LEAN step graph is captured with (rp_api.hip "lean graph"): its kernels check lean_dead / collision_done
    SimParams prm;
    int *flags;        // FL_* scalars
    unsigned *bar;     // [8] grid-barrier words of the fused rebuild kernels (rp_gridbar.h): {arrivals, base} per kernel
    long long *dbg;    // [64] cycle stamps of island 0 (only written when built with -DRP_ISL_PROFILE)
    int *host_flags;   // host-mapped (pinned) copy of the scalars, published by the last kernel of a step

    // ---- bodies (index = arena index) ----
    float4 *b_pos, *b_rot, *b_linvel, *b_angvel;
    float4 *b_lcom_invm;   // local_com.xyz, inv_mass
    float4 *b_invpi;       // inv principal inertia xyz
    float4 *b_pframe;      // principal inertia local frame (quat)
    float4 *b_wcom;        // world centre of mass
    float4 *b_eim;         // effective inverse mass (per axis)
    float4 *b_eii0, *b_eii1; // effective world inverse inertia: (m11 m12 m13 m22), (m23 m33 - -)
    float4 *b_damp;        // linear damping, angular damping, gravity scale, -
    float4 *b_uforce, *b_utorque;
    int *b_flags;
    int *b_collider;       // the LAST live collider of a dynamic body (-1 = none); the others follow through c_sibling
    int *c_sibling;        // [colliders] the previous live collider of the same dynamic body (-1 = none): compound bodies
    int *b_quar;           // sticky: non-finite state was detected (and rolled back) for this body
    float4 *b_ccd0_pos, *b_ccd0_rot; int *ccd_list; // continuous-collision pass (rp_ccd.h): start-of-step pose of the bodies on ccd_list
    // ---- sleeping (RigidBodyActivation + whole-island sleep, rp_sleep.hip) ----
    float4 *b_sleep;       // time_since_can_sleep, normalized_linear_threshold, angular_threshold, time_until_sleep
    float4 *b_sprev_t;     // sleep_prev_pose translation xyz, max_extent
    float4 *b_sprev_r;     // sleep_prev_pose rotation
    int *b_slabel;         // sleep-island label = smallest body index of the component (union-find parent while awake)
    int *b_sleep_stamp;    // step whose sleep observation this body's timer already holds (the observation runs at most once per step number:
                           // a fast step that aborts after observing is replayed on the full graph without counting the step twice)
    int *b_slept_at;       // step at which the body last fell asleep (clears the solver hints of its pairs)
    int *b_wake_req;       // pending wake-up: 1 weak, 2 strong, 3 strong + the user moved the body
    int *lab_wake;         // [n_bodies] per island id: step of the last wake-up of that sleeping island
    int *lab_awake;        // [n_bodies] per island id: step at which some member was found not eligible for sleep
    // ---- persistent islands (PersistentIslands, island_manager/persistent.rs; maintained while sleep_enabled) ----
    int *b_isl;            // RigidBodyIds::island_id per body, -1 = INVALID_ISLAND (fixed / removed bodies)
    int *pi_used, *pi_nb, *pi_dirty, *pi_denied, *pi_sleeping; // [n_bodies] per island id: in use, bodies.len(), constraint_remove_count > 0, split_denied_until, sleeping
    int *pi_free;          // [n_bodies] free_islands (stack)
    int *pi_uf, *pi_new;   // [n_bodies] scratch of a merge pass: union-find over island ids, surviving id of every island (-1 = not in use)
    unsigned long long *pi_best; // [n_bodies] scratch: per merge-group root, max of (bodies << 32 | ~id) = the identity that survives
    int *pi_csize, *pi_cisl, *pi_list; // [n_bodies] scratch: bodies per connected component (index = its smallest body), island of a component, compaction output
    unsigned long long *pi_w64;  // [4]: [0] = step << 32 | sleep_scan_stamp (the step whose scan bumped it), [1] = best split bid of the step (score bits << 32 | island id)
    int *pi_stats;         // [16] counters, same slots as the oracle's RO_IS_* (tests)
    unsigned long long *pj_key; int2 *pj_b; int pj_cap; // removal journal: phase << 62 | key, the two endpoints; capacity (a power of two)
    float4 *b_next_pos, *b_next_rot; // RigidBodyPosition::next_position of kinematic bodies (set_next_kinematic_position)
    // ---- solver bodies (index = arena index; non-dynamic = world-attached) ----
    float4 *s_lin, *s_ang, *s_rot, *s_trans, *s_incl, *s_inca;
    unsigned int *b_cmask; // 4 x u32 colour mask per body (body_solver_color_masks)
    // ---- substep solve-groups (rp_groups.h) ----
    SubParams *grp_sub;    // [RP_MAX_GROUPS] substep parameters per distinct count, descending count
    int *grp_extra;        // [RP_MAX_GROUPS] the counts
    int *b_extra;          // RigidBody::additional_solver_iterations per body
    int *b_group, *g_parent, *g_key; // per body: group index this step; union-find label and component count (scratch)
    int *k_group, *j_group;          // group of every constraint position / device joint this step
    unsigned long long *b_min; // colouring scratch: min pending key per body

    // ---- colliders ----
    int *c_parent, *c_shape;
    int *c_sub; int n_sub;  // sub-world of every collider (rp_world_begin_subworld; n_sub <= 1: one world, c_sub is not read)
    int *c_ord;            // ordinal of the collider among the colliders of its parent (attachment order, < 4096)
    float4 *c_lpos, *c_lrot, *c_pos, *c_rot, *c_he;
    float4 *c_mat;         // friction, restitution, density, -
    int2 *c_rules;
    uint2 *c_groups;
    float4 *c_fatmin, *c_fatmax;
    float2 *c_events;      // ActiveEvents bits (as int bits), contact_force_event_threshold
    // ---- event queues (rp_collision_event / rp_contact_force_event) ----
    int4 *ev_col;          // collider1, collider2, started | flags << 8, step
    int4 *ev_force_meta;   // collider1, collider2, step, started
    float4 *ev_force_a, *ev_force_b; // total_force xyz + magnitude ; max_force_direction xyz + max magnitude

    // ---- broad phase ----
    int *bk_cnt[2];        // [grid_cap] entries handed out per hash bucket (may exceed RP_BP_BUCKET: the surplus went to the large list); two
                           // copies: [FL_BP_EPOCH & 1] is in service, the other one rests at zero until the next rebuild fills it
    int *bk_items[2];      // [grid_cap][RP_BP_BUCKET] collider | which of its cells << 24 | range version << 29 (bp_entry)
    int *scan_block;       // [1024 + 8] scratch counters of a running rebuild ([1024]: its large list)
    int *large_list;
    int *large_sub_begin, *large_sub_cur, *large_tmp; int sub_cap; // the large list by sub-world (rp_grid.h large_range_of): begins [sub_cap + 2], scratch
    float4 *c_fatold_min, *c_fatold_max; // the fat AABB a collider queued in this pass had at the LAST pass (valid while c_chgstamp == the pass's stamp): rp_broadphase.hip bp_incr_insert
    int *c_chgstamp, *c_stale, *c_inlarge; // per collider: pass (FL_BP_SEQ + 1) that already queued it on bp_chg_list; (unused since round 4); on large_list
    int *c_rver;           // per collider: cell-range changes since the last full rebuild (the version its live grid entries carry: bp_grid_follow)
    int *bp_chg_list, *free_pending;       // [colliders] fat AABBs rewritten since the last pass; [pool_cap] slots freed by a running incremental pass
    unsigned long long *h_key[2]; int *h_slot[2];
    int *free_stack;

    // ---- pair pool ----
    int *p_c1, *p_c2, *p_stamp, *p_color, *p_nsc, *p_npts, *p_pflags, *p_reldom;
    int2 *p_colorb;
    int *p_hint_seq;            // step at which the pair's solver hint was last computed (pair_update.rs:141-161,636-650)
    int2 *p_rb;                 // parent bodies of the two colliders (c_parent is static), saves a dependent load
    int4 *p_aux;                // composite pairs: x, y, z = the AUX slots of solver manifolds (clusters) 2..4 or -1, w = cluster count (0 = plain manifold);
                                // an aux slot (RP_PF_AUX): x = its parent slot, y = its cluster index.  An aux slot holds ONE solver manifold of its
                                // parent's pair and is a pair slot to the solver only: no hash entry, skipped by the broad and narrow phase
    int2 *p_sub;                // composite pairs: the sub-shapes manifold 0 belongs to while the pair takes the plain path (-1: the collider itself)
    float4 *p_ln1, *p_ln2;      // manifold local normals
    float4 *p_normal;           // world normal xyz, friction
    float4 *p_misc;             // restitution, recycle max_extent, recycle max_drift, -
    float4 *r_t, *r_r, *r_rot1, *r_rot2; // recycle state
    float4 *pt_lp1d;            // [RP_MAX_PTS][pool]: local_p1.xyz, dist
    float4 *pt_lp2f;            // local_p2.xyz, (fid1 | fid2<<16) bits
    float4 *pt_imp;             // impulse, warmstart_impulse, warmstart_twist, -
    float4 *pt_wst;             // warmstart_tangent_world.xyz
    float4 *pt_dp1, *pt_dp2;    // frozen solver lever arms
    float4 *sc_a1;              // [4][pool]: anchor1.xyz, dist
    float4 *sc_a2;              // anchor2.xyz, contact id bits

    // ---- colouring / buckets ----
    int *todo_slot; unsigned long long *todo_key; int *todo_tmp;
    // k_color_pairs scratch (rp_narrowphase.hip): per-body lists of the queued pairs and the dependency DAG over them
    int *col_cnt, *col_fill, *col_begin; // [bodies] (cnt / fill rest at zero between launches)
    int *col_list, *col_sorted;          // [2 * pool] per-body lists: fill order, key order
    int4 *col_rec;                       // [pool] per queued pair: dynamic body 1, dynamic body 2 (-1 = none), first-toucher bits, pair slot
    int2 *col_rank, *col_succ;           // [pool] rank in either body's list; the next pair (queue index) at either body
    int *col_deps;                       // [pool] uncoloured predecessors
    int *col_q;                          // [2 * pool] frontier queues (double buffer)
    int *np_list;               // [pool] pair slots that failed the recycle test this step (k_np_test -> k_np_update)
    int *color_count, *color_begin, *color_cursor, *stage_color, *stage_begin, *stage_count;
    int *cons_pair;             // [cons_cap] position -> pair slot
    int *p_conspos;             // pair slot -> position (or -1)
    int *color_count_glob;      // per colour: manifolds on the global (non-island) path
    int *color_rank;            // colour -> stage index in the sweep order
    // positions inside a colour stage ascend with the manifold's owner body (its first awake dynamic body; unique inside a colour,
    // whose manifolds are body-disjoint): neighbouring lanes of the solver kernels then hold neighbouring bodies — coalesced body
    // records, and lanes that become ready together on the dataflow path
    int cb_words;               // ceil(body capacity / 32)
    unsigned *cb_bits;          // [128][cb_words] bit b of colour c: body b owns a global-path manifold of colour c
    int *cb_prefix;             // [128][cb_words] exclusive prefix popcount of cb_bits along the words of a colour

    // ---- contact islands (connected components of dynamic bodies over active manifolds) ----
    int *b_label;               // union-find labels
    int2 *uf_pairs;             // [pool] scratch: the two bodies of every active pair that links two awake non-fixed bodies
    int *b_island, *b_local;    // island id (>= 0: LDS island path, -1: global path), index inside the island
    int *r_nb, *r_nc, *r_island; // per-root scratch: body count, manifold count, island id (<= -2: -2 - the bundle of tiny components it joined)
    int *bun_nb, *bun_nc, *bun_ni, *bun_id; // per bundle of tiny components (rp_islands.hip, lay_isl_number): totals, the island it became
    int *p_island;              // pair slot -> island id or -1
    int *isl_body_begin, *isl_nb, *isl_cons_begin, *isl_nc, *isl_fill_b, *isl_fill_c;
    int *isl_bodies, *isl_cons;
    int *isl_cstage;            // [pool] local stage index of isl_cons[i] once the island list is sorted
    int *isl_cg1, *isl_cg2, *isl_cl1, *isl_cl2; // [pool] per sorted manifold: attached body of each side (arena / island-local index, -1 = world)
    int *isl_inc_pos, *isl_inc_begin, *isl_inc_cnt; // warm-start rows: per lane (2m + side) its row = rank in its body's sweep-ordered list ([2*pool]); per island body the row range ([n_bodies] x2)
    int *r_ni, *isl_ni, *isl_icons_begin, *isl_fill_i, *isl_icons; // inactive pairs (no solver contact) owned by an island: recycle-tested by the fused fast step
    int *isl_sorted, *isl_nstages; // per island: list sorted by sweep stage?, number of local stages

    // ---- impulse joints (active joints only, edge order) ----
    int *j_b1, *j_b2;           // arena index of the dynamic body on each side, -1 = world-attached side
    float4 *j_f1t, *j_f1r, *j_f2t, *j_f2r; // local frames in solver-body (CoM) space: translation, rotation
    int *j_locked, *j_limited, *j_color, *j_tmp, *j_order; // JointAxesMask of the locked / limited axes
    float4 *j_lim;              // [6][n_joints] limit of axis a: linear (min, max, -, -), angular AngularLimitParams (cos, sin, half_range, -)
    float4 *j_imp_lim, *j_imp_lim_ang; // JointLimits::impulse of the linear / angular axes
    int *j_motor;               // JointAxesMask of the motorised axes (GenericJoint::motor_axes)
    float4 *j_mot;              // [12][n_joints] JointMotor of axis a: plane 2a = (target_vel, target_pos, stiffness, damping), 2a + 1 = (max_force, model, -, -)
    float4 *j_imp_mot, *j_imp_mot_ang; // JointMotor::impulse of the linear / angular axes
    int *j_stage_begin, *j_stage_count;    // parallel joint colour stages inside j_order
    int *jc_first, *jc_list, *jc_sorted, *jc_deps, *jc_q; int2 *jc_rank, *jc_succ; // k_joint_color scratch: per-body joint lists and the dependency DAG (lists / queues: 2 x joints)
    float4 *j_imp, *j_imp_ang;  // per-dof impulses written back at the end of the step (linear dofs, angular dofs)
    unsigned int *bj_cmask;     // [4 * n_bodies] colours taken by joints (bodies_color workspace)
    unsigned long long *bj_min; // joint colouring scratch
    unsigned long long *nc_keys; // [n_nc] sorted (min body << 32 | max body) keys of the joints with contacts_enabled = false
    int *b_njoints;             // joints attached to a body (bodies with joints stay on the global path)
    float4 *JR;                 // [JR_COUNT][n_joints] constraint rows, up to 12 per joint (rp_joints.h)
    float2 *jm;                 // [2][12][n_joints] (impulse, rhs) of every row, two copies (worlds that may tile: rp_joints.h jm_get; null otherwise)

    // ---- constraints ----
    float4 *C;                  // [CP_COUNT][cons_cap]
    int *k_b1, *k_b2, *k_n, *k_cid;

    // ---- dataflow solver (rp_flow.hip): per-body toucher lists in sweep order ----
    float4 *f_rec;              // [2 * n_bodies] velocity records: (lin.xyz, tag), (ang.xyz, tag) side by side
    int2 *fk_rank;              // [cons_cap] position -> rank among the contact touchers of its body 1 / body 2 (-1 = world-attached side)
    int2 *fj_rank;              // [n_joints] joint -> rank among the joints of its body 1 / body 2 (in joint sweep order)
    int2 *fb_deg;               // [n_bodies] contact touchers, joint touchers of a solver body
    int2 *fb_begin, *fb_fill;   // [n_bodies] list begin / fill cursor inside f_adj, f_jadj
    int *f_adj, *f_jadj;        // [2 * cons_cap] positions, [2 * n_joints] joint sweep indices
    int *f_sorted;              // [2 * cons_cap] the contact touchers of every body in sweep order (f_adj ranked), as term rows 2 * position + side (side 0 = the manifold's body 1): the body-centric warm start, the cone walk of the tiles
    int *f_jsorted, *f_jother;  // [2 * n_joints] the joint touchers of every body in joint sweep order (indices into j_order) and the body on the other side (worlds that may tile)
    int *f_other;               // [2 * cons_cap] the solver body on the other side of f_sorted[i] (-1 = world-attached): the cone walk of rp_tiles.hip
    float4 *ws_terms;           // [11][2 * cons_cap] warm-start velocity terms per constraint side (rp_solver.hip: k_ws_prepare / k_increment_ws)

    // ---- LDS tiles of the global path (rp_tiles.hip): a whole colour sweep inside one CU per tile, halo constraints solved redundantly ----
    // ---- shard guard (multi-GPU island sharding: this world holds one shard, the cells below belong to the others) ----
    float4 *sg_bmin, *sg_bmax;  // [boxes] AABBs that hold the bodies of OTHER shards (one per foreign proximity group); null = no guard
    int *sg_cell_start, *sg_cell_items; // coarse uniform grid over the boxes: CSR lists of the boxes touching each cell (x fastest)
    int *ov_owner;              // [cons_cap] per manifold of the overflow colour (in sweep order): the dynamic body that owns it, -1 = two dynamic sides (lay_rank_overflow)
    int *lay_state;             // [16] layout-rebuild state that survives between rebuilds: [0] the flat component labels of the last rebuild are still valid (cleared by every edit of the world), [1] rebuilds so far, [2] global-path bodies the last rebuild counted; the broad phase keeps [8] the parity of the grid copy in service and [9] the rebuilds that kept the grid since its last build pass (rp_broadphase.hip); [3] / [4] island candidates of the last / the running layout rebuild, [5] the overflow colour may be swept owner-parallel (rp_islands.hip)
    int *sg_hit;                // [bodies] 1 = a rewritten fat AABB of the body overlapped a foreign box (read and cleared by rp_world_shard_guard_take_hits)
    float sg_origin[3], sg_inv_cell; int sg_dims[3];
    float sg_horizon;           // seconds a hit may wait for the caller: the tested box grows by |linvel| x horizon (rp_world_set_shard_guard_horizon)
    int c_par;                  // which copy of the MUTABLE constraint planes (impulses, accumulators, rhs: NP_M x 4, CP_HM0, CP_HM1) is current:
                                // 0 = in place, 1 = the shadow planes behind CP_COUNT.  A tile sweep reads one copy and its owner instances write
                                // the other (a halo instance must not see the owner's result of the same sweep); every other kernel works in
                                // place on the current copy (cplane(), rp_constraint.h).  Always 0 outside a tiled solver loop.
    int tile_cap;               // tiles the arrays below hold (0 = the world never tiles: joints, Coulomb, solve groups, RP_NO_TILES=1)
    int tile_target;            // tiles wanted for the global component (~ one per CU)
    int tile_min;               // global-path bodies / manifolds below which no tiling is attempted (RP_TILE_MIN_BODIES; a test hook: RP_TILE_MIN=<n>)
    float4 *t_lin, *t_ang;      // [n_bodies] the other half of the solver-velocity double buffer: a tile sweep reads s_lin / s_ang and writes here
    float4 *t_rot, *t_trans;    // [n_bodies] ... and of the solver poses: written by the sweep that also integrates (another tile may still read s_rot / s_trans)
    int2 *fk_ids;               // [cons_cap] the two solver bodies of every position (flow_ids), refreshed with the tiling
    int *tl_body_tile;          // [n_bodies] owner tile of a global-path body, -1 otherwise
    int *tl_owned;              // [n_bodies] global-path bodies in tile order (Morton cell order of the centres of mass)
    int *tl_hist, *tl_cellofs;  // [2][RP_TILE_CELLS] counting sort by cell: every body / global-path bodies (hist rests at zero between rebuilds)
    int *tl_cell, *tl_sorted;   // [n_bodies] Morton cell of a body (-1 - cell off the global path); the bodies cell by cell
    int *b_order;               // [n_bodies] curve rank of every body (a permutation of the indices the last sort covered, i beyond them): the order
                                // of the owner bodies inside a colour stage (k_layout_rebuild); null = arena-index order
    unsigned *tl_bbox;          // [16] ordered-uint min xyz, max xyz of the centres of mass (rest state: min = ~0, max = 0), [6] global-path bodies, [7] T, [8] sorted?, [9] bodies sorted
    int4 *tl_hdr;               // [tile_cap] cone bodies, cone constraints, owned bodies, -
    int *tl_soff;               // [tile_cap][RP_TILE_STAGES + 1] begin of every sweep stage inside the tile's constraint list
    int *tl_bodies;             // [tile_cap][RP_TILE_BCAP] arena index of every cone body (owned + halo), index = tile-local id
    int4 *tl_cons;              // [tile_cap][RP_TILE_CCAP] cone constraints in stage order: position, local body 1, local body 2, owner?
    unsigned *tl_nbr;           // [tile_cap][RP_TILE_NBR_WORDS(tile_cap)] bit B of row A: tiles A and B exchange bodies between sweeps (one holds a body the
                                // other owns; symmetric) — whom a tile of k_joint_net_step waits for (rebuilt with the cones)
    unsigned *tl_flag;          // [tile_cap][32] (one 128-byte line each) the sweep a tile has published: 16 x FL_SEQ of the launch + sweeps done
};
#define RP_TILE_NBR_WORDS(tile_cap) (((tile_cap) + 31) / 32)
// ---- lean step graphs (rp_api.hip "lean graph") ----------------------------------------------------------------------------------
// A MULTI-mode world whose contact graph did not change this step needs none of the launches that rebuild the colouring, the joint
// colouring, the layout, the toucher ranks or the tiling — nine early exits per step.  The LEAN graph leaves them out: collision
// detection, then straight to the solver.  Whether that was right is known on the device once the narrow phase has run: the
// solver kernels of a lean graph all evaluate the same condition (the flags below do not change after k_np_update in a lean graph)
// and exit when work for the left-out launches turned up; the last kernel of the graph (k_ccd) then raises FL_FAST_ABORT = 2, which
//   - makes every later lean graph a no-op (collision_done / lean_dead), and
//   - makes the next FULL graph resume the step: its collision kernels skip (that stage already ran for this step, its results are
//     in place), the rebuild launches run, the solver runs; k_color_pairs — first kernel behind the collision stage, full graphs
//     only — clears the marker.
// The host sees the marker in the hint buffer and stops enqueueing lean graphs; steps that died are replayed by settle() like
// aborted fast steps.
#if defined(__HIPCC__)
__device__ __forceinline__ bool lean_dead(const DevWorld &w) {
    if (!w.lean) return false;
    // (bit 1, a BARE lean graph: it also left out the launches that only manifolds and LDS islands give work to.  Both counts are
    // results of the layout rebuild, which no lean graph runs: they cannot change while this graph executes)
    const int bare_wrong = (w.lean & 2) ? (w.flags[FL_N_CONS_ALL] | w.flags[FL_N_ISLANDS]) : 0;
    // (bit 2, a bare lean graph whose TGS loop is ONE launch — k_joint_net_step, rp_tiles.hip —, its grid in bits 8 and up: a valid
    // tiling of at most that many tiles, no contact stage, no cone with more joints than the kernel has threads.  All four are results
    // of the layout rebuild / the tiling, which no lean graph runs)
    int jn_wrong = 0;
    if (w.lean & 4) { const int nt = w.flags[FL_N_TILES], jmax = w.flags[FL_TILE_JMAX]; jn_wrong = (w.flags[FL_JN_TIMEOUT] != 0 || nt <= 0 || nt > (w.lean >> 8) || nt > RP_JN_THREADS || w.prm.num_substeps > 8 || w.flags[FL_N_STAGES] != 0 || jmax <= 0 || jmax > RP_JN_THREADS) ? 1 : 0; }
    // (bit 3, a lean graph of a tiled contact world whose TGS loop is ONE launch — k_tile_step, rp_tiles.hip —, its grid in bits 8 and up: a
    // valid tiling of at most that many tiles, at most five substeps — a launch owns sixteen values of a tile's flag, three per substep)
    if (w.lean & 8) { const int nt = w.flags[FL_N_TILES]; jn_wrong = (w.flags[FL_JN_TIMEOUT] != 0 || nt <= 0 || nt > (w.lean >> 8) || w.prm.num_substeps > 5 || w.n_joints > 0 || w.flags[FL_N_ISLANDS] != 0) ? 1 : 0; } // (no LDS island: k_island_solve commits its bodies before this launch could give up)
    // (no bit 0: a FULL graph in the one-launch form — every rebuild ran, the flags below speak about work a lean graph leaves out (FL_TODO_COUNT
    // stays up behind the colouring until the next step's first kernel); only the launch itself can fail)
    if (!(w.lean & 1)) return jn_wrong != 0;
    return (w.flags[FL_FAST_ABORT] | w.flags[FL_TODO_COUNT] | w.flags[FL_LAYOUT_DIRTY] | w.flags[FL_FLOW_DIRTY] | (w.n_joints > 0 ? w.flags[FL_JOINT_DIRTY] : 0) | bare_wrong | jn_wrong) != 0;
}
// collision kernels: this step's collision stage already ran (a dead lean step waits for its resume), or an earlier lean step died
__device__ __forceinline__ bool collision_done(const DevWorld &w) { const int a = w.flags[FL_FAST_ABORT]; return a == 2 || (w.lean && a != 0); }
#endif

