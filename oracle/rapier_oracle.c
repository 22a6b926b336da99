/*
 * oracle/rapier_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Single-threaded scalar CPU restatement of rapier3d/f32 `PhysicsPipeline::step()`
 * (reference v0.35.2).  Every function cites the reference lines it follows; paths are
 * relative to /root/reference/src.  See rapier_oracle.h for the parity-pinning statement.
 *
 * PARITY UNPINNED: the reference is Rust with un-vendored crates (parry3d, nalgebra, glam) and cannot be built in this image, so
 * there is no oracle/_ref; the one bitwise golden of the reference reachable without cargo (the FNV-1a state hash of
 * crates/rapier3d/tests/simd_backend_determinism.rs:144) does NOT match this restatement (tests/test_reference_golden.py, strict
 * xfail).  What pins the oracle are the reference's outcome-level tests restated in tests/test_oracle_kat.py
 * and tests/test_reference_kats.py.
 *
 * Ordering rules that are part of the numerical contract and are
 * reproduced here exactly:
 *   1. pair colour = greedy first-fit over per-body u128 masks on begin-touch pairs sorted by
 *      (min body index, max body index, edge)  — geometry/narrow_phase/contacts.rs:369-385,
 *      geometry/narrow_phase/mod.rs:90-154;
 *   2. sweep order = colours with >= 32 four-lane chunks ascending, then the smaller colours
 *      ascending, then the overflow colour — dynamics/solver/staged_island_solver/init.rs:163-254;
 *   3. per substep: increment(+gyro) -> per colour update+warmstart -> biased solve (no friction)
 *      -> integrate -> refreshed unbiased solve with friction — worker.rs:207-650.
 * Deliberate simplification (documented in DESIGN.md): the overflow colour is swept in bucket
 * order instead of the body-mask regrouped order of interaction_groups.rs:240-355 (it only
 * fills when a body has >120 simultaneous dynamic neighbours).
 */
#include "rapier_oracle.h"
#include "ro_shapes.h"
#include "ro_convex.h"
#include "ro_ccd.h"
/* Optional OpenMP (bench.py's cpu_baseline leg): loops over items that touch pairwise-disjoint state
 * — the pairs of the narrow phase, the bodies, the constraints of one colour (the reference runs
 * exactly these loops on its rayon pool, staged_island_solver/worker.rs) — are parallel; the colour
 * order and every f32 expression are unchanged, so results do not depend on the thread count. */
#ifdef _OPENMP
#include <omp.h>
#define RO_PRAGMA(x) _Pragma(#x)
#define RO_PARALLEL_FOR RO_PRAGMA(omp parallel for schedule(static) if (ro_threads > 1) num_threads(ro_threads))
#define RO_PARALLEL_FOR_RED(a, b) RO_PRAGMA(omp parallel for schedule(static) reduction(+ : a, b) if (ro_threads > 1) num_threads(ro_threads))
#define RO_PRAGMA_IF_PAR(serial) RO_PRAGMA(omp parallel for schedule(static) if (ro_threads > 1 && !(serial)) num_threads(ro_threads))
#else
#define RO_PARALLEL_FOR
#define RO_PARALLEL_FOR_RED(a, b)
#define RO_PRAGMA_IF_PAR(serial)
#endif
static int ro_threads = 1;
void ro_set_threads(int32_t n) { ro_threads = n < 1 ? 1 : n; }
int32_t ro_get_threads(void) { return ro_threads; }
#include <stdio.h>
#include <stdlib.h>

#define RO_NUM_COLORS 129
#define RO_COLOR_OVERFLOW 128
#define RO_COLOR_UNCOLORED 255
#define RO_DYNAMIC_COLOR_COUNT 120 /* contact_pair.rs:167 */
#define RO_NO_BODY 0xffffffffu

typedef struct { uint64_t lo, hi; } u128;

typedef struct {
    int body_type;
    pose position, next_position;
    v3 linvel, angvel;
    /* local mass properties (parry MassProperties) */
    v3 local_com; float inv_mass; v3 inv_principal_inertia; quat principal_frame;
    /* world mass properties — rigid_body_components.rs:528-578 */
    v3 world_com; v3 effective_inv_mass; sym3 effective_world_inv_inertia;
    v3 force, torque, user_force, user_torque;
    float linear_damping, angular_damping, gravity_scale, additional_mass;
    int dominance, gyroscopic, allow_fast_rotation;
    uint32_t locked_axes; /* LockedAxes — rigid_body_components.rs:271-288 */
    int ncolliders, first_collider;
    uint32_t solver_id; /* active_set_id, RO_NO_BODY for bodies outside the active set */
    /* RigidBodyActivation — rigid_body_components.rs:1300-1480 */
    float normalized_linear_threshold, angular_threshold, time_until_sleep, time_since_can_sleep;
    int sleeping; pose sleep_prev_pose;
    float max_extent;   /* RigidBodyMassProps::max_extent — rigid_body_components.rs:491-515 */
    int slept_at;       /* step at which the body last fell asleep (pair hints cleared then) */
    int island_id;      /* RigidBodyIds::island_id: persistent island of a non-fixed body, -1 = INVALID_ISLAND */
    int wake_req;       /* pending IslandManager::wake_up */
    /* RigidBodyCcd — rigid_body_components.rs:1170-1230 */
    int ccd_enabled, ccd_active; float ccd_thickness;
    int additional_solver_iterations; /* RigidBody::additional_solver_iterations — extra substeps for the body's whole component */
    int last_group_extra;             /* extra substeps of the solve group the body was in during the last step (-1: not in the active set) */
} Body;

typedef struct { v3 mins, maxs; } Aabb;

typedef struct {
    int parent; pose pos_wrt_parent, pos;
    int shape; v3 he; float radius; int axis; /* capsule: he.x = half height, radius, axis */
    float border;       /* round shapes (RO_SHAPE_ROUND_*): RoundShape::border_radius; 0 otherwise */
    const RoPolyhedron *poly; /* RO_SHAPE_CONVEX_POLYHEDRON: the registered polyhedron (recentred; the centre is folded into pos_wrt_parent) */
    const struct RoComposite *comp; /* RO_SHAPE_COMPOUND / RO_SHAPE_TRIMESH: the registered composite (recentred on its local AABB like a polyhedron; he = that box) */
    v3 tri[3];          /* RO_SHAPE_TRIANGLE (a mesh triangle handed to the shape dispatcher): its vertices */
    int sensor;         /* ColliderBuilder::sensor(true): intersection events only, no contacts (oracle only so far) */
    float density, friction, restitution; int friction_rule, restitution_rule;
    uint32_t memberships, filter;
    uint32_t active_events; float force_threshold;
    int ord;            /* ordinal among the colliders attached to the same parent (attachment order) */
    Aabb fat; int has_fat;
    int sub;            /* sub-world (ro_begin_subworld): colliders of different sub-worlds never pair */
} Collider;

/* SolverContact — contact_pair.rs:617-629 */
typedef struct { v3 anchor1, anchor2; float dist; v3 tangent_velocity; int cid; } SolverContact;

typedef struct {
    int c1, c2;
    int alive;
    Manifold m;
    /* ContactManifoldData */
    int intersecting;   /* IntersectionPair::intersecting (sensor pairs) */
    v3 normal; float friction, restitution; int relative_dominance;
    SolverContact sc[4]; int nsc;
    uint32_t solver_body_ids[2];
    /* recycle state — contact_pair.rs:262-278 */
    int has_recycle; pose rec_pos12; quat rec_rot1, rec_rot2; float rec_max_extent, rec_max_drift;
    uint8_t color; uint32_t color_bodies[2];
    int force_emitted;  /* PairEventStatus::INITIAL_FORCE_THRESHOLD_EVENT_EMITTED */
    int hint_seq;       /* step at which pair_solver_hints[edge] was last computed (pair_update.rs:141-161,636-650) */
    /* Composite pairs (ro_composite.h).  The fields above (m, normal, sc, nsc) are SOLVER MANIFOLD 0 of the pair: the plain manifold of the
     * one candidate sub-shape pair, or — when contact clustering applies (pair_update.rs:350) — cluster 0; ex[k - 1] = cluster k.  The
     * clusters with solver contacts come first (stable), so nsc > 0 still means has_any_active_contact. */
    int ncl;            /* 0 = plain manifold (or none); >= 1 = that many clusters (ContactPair::solver_clusters) */
    int plain_sub[2];   /* the sub-shapes manifold 0 belongs to while ncl == 0 (-1: the collider itself) */
    struct ExtraCluster *ex; /* [RO_MAX_CLUSTERS - 1], allocated on first use */
} Pair;
#define RO_MAX_CLUSTERS 4      /* solver manifolds per pair (the reference: unbounded; a 5th normal direction joins the closest cluster) */
#define RO_CLUSTER_PTS 32      /* points of one cluster while it is built (the reference: 255) */
#define RO_MAX_SUBPAIRS 64     /* candidate sub-shape pairs of one collider pair per step (in index order; the rest is ignored and counted) */
typedef struct ExtraCluster { Manifold m; v3 normal; SolverContact sc[4]; int nsc; } ExtraCluster;
/* solver manifold k of a pair */
static inline Manifold *sm_m(Pair *p, int k) { return k == 0 ? &p->m : &p->ex[k - 1].m; }
static inline SolverContact *sm_sc(Pair *p, int k) { return k == 0 ? p->sc : p->ex[k - 1].sc; }
static inline int *sm_nsc(Pair *p, int k) { return k == 0 ? &p->nsc : &p->ex[k - 1].nsc; }
static inline v3 *sm_normal(Pair *p, int k) { return k == 0 ? &p->normal : &p->ex[k - 1].normal; }
static inline int pair_num_sm(const Pair *p) { return p->ncl > 1 ? p->ncl : 1; }
static inline void pair_clear_clusters(Pair *p) { p->ncl = 0; p->plain_sub[0] = p->plain_sub[1] = -1; if (p->ex) for (int k = 0; k < RO_MAX_CLUSTERS - 1; ++k) { p->ex[k].nsc = 0; p->ex[k].m.npoints = 0; } }
#define RO_SM_SHIFT 28          /* solver-manifold reference = pair index | k << 28 */
#define RO_SM_PAIR(x) ((x) & ((1 << RO_SM_SHIFT) - 1))
#define RO_SM_K(x) ((int)((unsigned)(x) >> RO_SM_SHIFT))

/* ContactWithTwistFriction + builder, one lane — contact_with_twist_friction.rs:18-55,600-630 */
typedef struct {
    v3 torque_dir1, torque_dir2, ii_torque_dir1, ii_torque_dir2;
    float rhs, rhs_wo_bias, impulse, impulse_accumulator, r, cfm_factor;
} NormalPart;
typedef struct {
    v3 dir1, im1, im2; sym3 ii1, ii2; float cfm_factor, limit; v3 tangent1;
    NormalPart normal_part[4];
    struct { v3 dp1, dp2, torque_dir1[2], torque_dir2[2], ii_torque_dir1[2], ii_torque_dir2[2];
             float rhs[2], rhs_wo_bias[2], impulse[2], impulse_accumulator[2], r[3]; } tangent_part;
    struct { float rhs, impulse, impulse_accumulator, r; } twist_part;
    float twist_dists[4];
    /* ContactWithCoulombFriction: one ContactConstraintTangentPart per point (contact_constraint_element.rs:17-36) */
    struct { v3 torque_dir1[2], torque_dir2[2], ii_torque_dir1[2], ii_torque_dir2[2];
             float rhs[2], rhs_wo_bias[2], impulse[2], impulse_accumulator[2], r[3]; } ctangent[4];
    uint32_t solver_vel1, solver_vel2; int pair; int num_contacts; int contact_id[4];
    /* builder */
    struct { float restitution_seed; v3 local_p1, local_p2; float dist; } infos[4];
    v3 local_friction_center1, local_friction_center2, tangent_vel, local_n1; float restitution;
} Constraint;

typedef struct { v3 linear, angular; } SolverVel;
typedef struct { quat rotation; v3 translation; sym3 ii; v3 im; } SolverPose;

/* ImpulseJoint (joint/impulse_joint/impulse_joint.rs) + JointConstraintBuilder
 * (solver/joint_constraint/joint_constraint_builder.rs:19-60), restricted to locked linear axes
 * (spherical joints: JointAxesMask::LIN_AXES). */
typedef struct {
    int body1, body2;
    pose local_frame1, local_frame2;      /* GenericJoint::local_frame1/2 (body space) */
    uint32_t locked_axes; int contacts_enabled;
    uint32_t coupled_axes; /* GenericJoint::coupled_axes */
    uint32_t limit_axes; float limits[6][2];   /* GenericJoint::{limit_axes, limits} */
    float ang_limit_center[3][2], ang_limit_half_range[3]; /* AngularLimitParams (joint_constraint_helper.rs:34-72) */
    float limit_impulses[6];                    /* JointLimits::impulse */
    uint32_t motor_axes; ro_joint_motor motors[6]; /* GenericJoint::{motor_axes, motors} */
    float motor_impulses[6];                    /* JointMotor::impulse */
    uint8_t solver_color;                 /* persistent colour, impulse_joint.rs:38 */
    uint32_t solver_body_ids[2];          /* stamped by select_active_interactions */
    float impulses[6];                    /* per-dof impulses written back last step */
    pose sb_frame1, sb_frame2;            /* frames in solver-body (CoM) space */
    int first_row;
    int removed;                          /* ImpulseJointSet::remove */
    int linked;                           /* ImpulseJointIslandEvent::Link applied (persistent.rs:13-24) */
} Joint;
/* JointConstraint<Real, 1> — joint_velocity_constraint.rs:71-95 */
typedef struct {
    uint32_t solver_vel1, solver_vel2;
    float impulse, impulse_bounds[2];
    v3 lin_jac, ang_jac1, ang_jac2, ii_ang_jac1, ii_ang_jac2;
    float inv_lhs, rhs, rhs_wo_bias, cfm_gain, cfm_coeff;
    v3 im1, im2;
    int dof;                              /* WritebackId: Dof(i) = i, Limit(i) = 6 + i, Motor(i) = 12 + i */
} JointRow;
typedef struct { int enabled; v3 principal_inertia, inv_principal_inertia; quat principal_frame; } Gyro;

/* PersistentIsland — island_manager/persistent.rs:74-93.  The body / link vectors of the reference are replaced by a per-body
 * island id + counts: every decision below needs the PARTITION, the body count, and these four scalars. */
typedef struct PIsland {
    int used, nbodies;
    int dirty;      /* constraint_remove_count > 0 (only ever tested against zero: persistent.rs:184, :508) */
    int denied;     /* split_denied_until */
    int sleeping;
} PIsland;
/* Removal (persistent.rs:98-103) + the canonical position of the unlink in the step: phase 0 = joint edits drained at the top of
 * the step, 1 = pairs deleted by the broad phase, 2 = end-touch transitions of the narrow phase; key orders a phase. */
typedef struct Removal { int body1, body2; int phase; uint64_t key; } Removal;
#define RO_SPLIT_RETRY_COOLDOWN 16   /* persistent.rs:31 */
#define RO_SEARCH_BUDGET 1024        /* local_split.rs:21 */

struct ro_world {
    ro_params params; v3 gravity;
    Body *bodies; int nbodies, cap_bodies;
    Collider *colliders; int ncolliders, cap_colliders;
    int cur_sub, n_sub; /* ro_begin_subworld */
    RoPolyhedron **polys; int npolys; /* ro_add_convex_polyhedron */
    struct RoComposite **comps; int ncomps; /* ro_add_compound / ro_add_trimesh / ro_add_heightfield */
    int subpair_overflows; /* collider pairs that met more than RO_MAX_SUBPAIRS candidate sub-shape pairs (cumulative) */
    Pair *pairs; int npairs, cap_pairs;
    /* open-addressing map (c1,c2) -> pair index */
    int64_t *map_keys; int *map_vals; int map_cap;
    u128 *color_masks; int cap_masks;
    int bp_dirty;
    /* solver scratch */
    SolverVel *vels, *incr; SolverPose *poses; Gyro *gyro; uint8_t *flags; int *dyn_bodies; int ndyn;
    Constraint *cons; int ncons, cap_cons;
    int bucket_begin[RO_NUM_COLORS + 1];
    int stage_color[RO_NUM_COLORS]; int nstages;
    /* joints */
    Joint *joints; int njoints, cap_joints;
    int *active_joints; int nactive_joints;       /* select_active_interactions output (edge order) */
    int *joint_order; int njoint_parallel;         /* solve order: parallel colours ascending, then the serial overflow */
    JointRow *joint_rows; u128 *joint_body_colors;
    ro_stats stats;
    int step_seq;       /* 1-based number of the step in progress */
    int *uf;            /* union-find scratch of the sleep islands */
    /* PersistentIslands — island_manager/persistent.rs:128-171 */
    struct PIsland *isl; int isl_cap, isl_next; int *isl_free; int n_isl_free, cap_isl_free;
    int scan_stamp;     /* sleep_scan_stamp */
    int pending_split;  /* split_island: the candidate chosen last step, -1 = None */
    struct Removal *journal; int njournal, cap_journal; /* removal_journal */
    int32_t pi_stats[RO_ISLAND_STATS];
    int32_t ccd_active_count, ccd_clamp_count; /* (body, step) cases of the fast-body criterion / of a clamped next_position */
    int nfree_colliders; /* colliders inserted without a parent */
    /* Arena free lists (data/arena.rs:28-90, 260-380): a removed slot is handed out again, LIFO, before a fresh index is; the
     * generation an Index carries is the arena's removal count at insertion time (ro_body_generation / ro_collider_generation). */
    int *body_free; int nbody_free, cap_body_free; uint32_t body_arena_gen; uint32_t *body_gen; int cap_body_gen;
    int *coll_free; int ncoll_free, cap_coll_free; uint32_t coll_arena_gen; uint32_t *coll_gen; int cap_coll_gen;
    int dead_pairs;      /* colliders were removed since the last broad-phase pass: their pairs are still in the pair set */
    uint64_t *nc_keys; int n_nc, nc_dirty; /* sorted (min body, max body) keys of the joints with contacts_enabled = false */
    int32_t *col_events; int ncol_events, cap_col_events;       /* 5 ints per event */
    int32_t *force_meta; float *force_vals; int nforce_events, cap_force_events;
};

/* ------------------------------------------------------------------------------------ */
void ro_default_params(ro_params *p) {
    /* integration_parameters.rs:379-408 */
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
    p->warmstart_joints = 0;
    p->max_ccd_substeps = 1;
    p->min_ccd_dt = 1.0f / 60.0f / 100.0f;
    p->contact_clustering = 1;
    p->friction_model = RO_FRICTION_SIMPLIFIED;
}

/* SpringCoefficients — integration_parameters.rs:86-149 */
static float spring_erp_inv_dt(float freq, float damping, float dt) {
    float ang_freq = freq * 6.283185307179586f; /* simd_two_pi */
    return ang_freq / (dt * ang_freq + 2.0f * damping);
}
static float spring_cfm_factor(float freq, float damping, float dt) {
    float erp = dt * spring_erp_inv_dt(freq, damping, dt);
    float cfm_coeff = 0.0f;
    if (erp != 0.0f) {
        float inv_erp_minus_one = 1.0f / erp - 1.0f;
        cfm_coeff = inv_erp_minus_one * inv_erp_minus_one / ((1.0f + inv_erp_minus_one) * 4.0f * damping * damping);
    }
    return 1.0f / (1.0f + cfm_coeff);
}

static float spring_cfm_coeff(float freq, float damping, float dt) {
    float erp = dt * spring_erp_inv_dt(freq, damping, dt);
    if (erp == 0.0f) return 0.0f;
    float inv_erp_minus_one = 1.0f / erp - 1.0f;
    return inv_erp_minus_one * inv_erp_minus_one / ((1.0f + inv_erp_minus_one) * 4.0f * damping * damping);
}

float ro_combine_coefficient(float a, float b, int32_t ra, int32_t rb) {
    /* coefficient_combine_rule.rs:58-86 */
    int rule = ra > rb ? ra : rb;
    switch (rule) {
    case RO_RULE_AVERAGE: return (a + b) / 2.0f;
    case RO_RULE_MIN: return fabsf(a < b ? a : b);
    case RO_RULE_MULTIPLY: return a * b;
    case RO_RULE_MAX: return a > b ? a : b;
    case RO_RULE_CLAMPED_SUM: return ro_clampf(a + b, 0.0f, 1.0f);
    default: return sqrtf(ro_maxf(a, 0.0f) * ro_maxf(b, 0.0f));
    }
}

ro_world *ro_world_new(const ro_params *params, const float gravity[3]) {
    ro_world *w = (ro_world *)calloc(1, sizeof(ro_world));
    w->n_sub = 1;
    w->params = *params;
    w->gravity = V3(gravity[0], gravity[1], gravity[2]);
    w->map_cap = 1 << 12;
    w->map_keys = (int64_t *)malloc(sizeof(int64_t) * w->map_cap);
    w->map_vals = (int *)malloc(sizeof(int) * w->map_cap);
    for (int i = 0; i < w->map_cap; ++i) w->map_keys[i] = -1;
    w->bp_dirty = 1;
    w->pending_split = -1;
    return w;
}
void ro_set_params(ro_world *w, const ro_params *params) { w->params = *params; }
struct RoComposite; static void comp_free_fwd(struct RoComposite *C);
void ro_world_free(ro_world *w) {
    if (w) { for (int i = 0; i < w->npairs; ++i) free(w->pairs[i].ex); for (int i = 0; i < w->ncomps; ++i) comp_free_fwd(w->comps[i]); free(w->comps); }
    if (!w) return;
    for (int i = 0; i < w->npolys; ++i) { ro_poly_free(w->polys[i]); free(w->polys[i]); }
    free(w->polys);
    free(w->bodies); free(w->colliders); free(w->pairs); free(w->map_keys); free(w->map_vals);
    free(w->body_free); free(w->body_gen); free(w->coll_free); free(w->coll_gen);
    free(w->color_masks); free(w->vels); free(w->incr); free(w->poses); free(w->gyro); free(w->flags);
    free(w->dyn_bodies); free(w->cons); free(w->joints); free(w->active_joints); free(w->joint_order);
    free(w->joint_rows); free(w->joint_body_colors); free(w->uf); free(w->col_events); free(w->force_meta); free(w->force_vals); free(w->nc_keys); free(w->isl); free(w->isl_free); free(w->journal); free(w);
}

/* ---- Persistent islands: allocation and membership — island_manager/persistent.rs:196-288 ------------------------------------
 * Island ids are handed out exactly like PersistentIslands::alloc_island: the most recently freed id first, otherwise the next
 * unused one.  Fixed bodies are never members; a removed body becomes an inert fixed row here, so "member" = non-fixed. */
static int pi_alloc(ro_world *w) {
    int id = w->n_isl_free ? w->isl_free[--w->n_isl_free] : w->isl_next++;
    if (id >= w->isl_cap) {
        int nc = w->isl_cap ? w->isl_cap : 1024; while (nc <= id) nc *= 2;
        w->isl = (PIsland *)realloc(w->isl, sizeof(PIsland) * (size_t)nc);
        memset(w->isl + w->isl_cap, 0, sizeof(PIsland) * (size_t)(nc - w->isl_cap));
        w->isl_cap = nc;
    }
    PIsland z = {1, 0, 0, 0, 0};
    w->isl[id] = z;
    return id;
}
static void pi_free(ro_world *w, int id) { /* free_island: a pending split of the freed island is dropped (:211-213) */
    w->isl[id].used = 0;
    if (w->pending_split == id) w->pending_split = -1;
    if (w->n_isl_free == w->cap_isl_free) { w->cap_isl_free = w->cap_isl_free ? 2 * w->cap_isl_free : 1024; w->isl_free = (int *)realloc(w->isl_free, sizeof(int) * (size_t)w->cap_isl_free); }
    w->isl_free[w->n_isl_free++] = id;
}
/* ensure_body (:217-232): a first-seen non-fixed body gets a singleton island */
static void pi_ensure_body(ro_world *w, int body) {
    Body *b = &w->bodies[body];
    if (b->body_type == RO_BODY_FIXED || b->island_id >= 0) return;
    int id = pi_alloc(w);
    b->island_id = id; w->isl[id].nbodies = 1; w->isl[id].sleeping = b->sleeping;
}
/* remove_body_raw (:254-288): losing a body can split the island exactly like losing a constraint (a body can be a cut vertex),
 * so the island is dirtied eagerly; the last body frees it */
static void pi_remove_body(ro_world *w, int body) {
    Body *b = &w->bodies[body];
    int id = b->island_id;
    b->island_id = -1;
    if (id < 0 || !w->isl[id].used) return;
    w->isl[id].nbodies--; w->isl[id].dirty = 1;
    if (w->isl[id].nbodies == 0) pi_free(w, id);
}
/* unlink_contact / unlink_joint -> journal_removal (:331-343, :395-418): only the endpoints are recorded, self-loops are not */
static void pi_journal(ro_world *w, int body1, int body2, int phase, uint64_t key) {
    if (body1 == body2 || body1 < 0 || body2 < 0) return; /* a side without a body carries no connectivity (local_split.rs:188-196) */
    if (w->njournal == w->cap_journal) { w->cap_journal = w->cap_journal ? 2 * w->cap_journal : 256; w->journal = (Removal *)realloc(w->journal, sizeof(Removal) * (size_t)w->cap_journal); }
    Removal r = {body1, body2, phase, key};
    w->journal[w->njournal++] = r;
}

/* parry MassProperties::world_inv_inertia */
static sym3 world_inv_inertia(v3 inv_pi, quat frame, quat rot) {
    sym3 r = {0, 0, 0, 0, 0, 0};
    if (inv_pi.x == 0.0f && inv_pi.y == 0.0f && inv_pi.z == 0.0f) return r;
    float m[3][3]; quat_to_mat(qmul(rot, frame), m);
    float d[3] = {inv_pi.x, inv_pi.y, inv_pi.z};
#define E(i, j) (m[i][0] * d[0] * m[j][0] + m[i][1] * d[1] * m[j][1] + m[i][2] * d[2] * m[j][2])
    r.m11 = E(0, 0); r.m12 = E(0, 1); r.m13 = E(0, 2); r.m22 = E(1, 1); r.m23 = E(1, 2); r.m33 = E(2, 2);
#undef E
    return r;
}
/* RigidBodyMassProps::update_world_mass_properties — rigid_body_components.rs:528-578 */
static void update_world_mass_properties(Body *b) {
    b->world_com = pose_tp(b->position, b->local_com);
    if (b->body_type == RO_BODY_DYNAMIC) {
        b->effective_inv_mass = V3(b->inv_mass, b->inv_mass, b->inv_mass);
        b->effective_world_inv_inertia = world_inv_inertia(b->inv_principal_inertia, b->principal_frame, b->position.r);
        /* translation / rotation locking (:533-571) */
        uint32_t la = b->locked_axes;
        if (la & 1u) b->effective_inv_mass.x = 0.0f;
        if (la & 2u) b->effective_inv_mass.y = 0.0f;
        if (la & 4u) b->effective_inv_mass.z = 0.0f;
        sym3 *ii = &b->effective_world_inv_inertia;
        if (la & 8u) { ii->m11 = 0.0f; ii->m12 = 0.0f; ii->m13 = 0.0f; }
        if (la & 16u) { ii->m22 = 0.0f; ii->m12 = 0.0f; ii->m23 = 0.0f; }
        if (la & 32u) { ii->m33 = 0.0f; ii->m13 = 0.0f; ii->m23 = 0.0f; }
    } else {
        b->effective_inv_mass = V3(0, 0, 0);
        sym3 z = {0, 0, 0, 0, 0, 0}; b->effective_world_inv_inertia = z;
    }
}

/* parry Shape::mass_properties (cuboid / ball / capsule), SURVEY Appendix C.  `frame` = principal inertia local frame of the
 * shape (identity except for capsules along X / Z: MassProperties::from_capsule rotates Y onto the segment direction). */
static void ro_diagonalise(float a[3][3], float pi[3], float frame[4]);
/* the inner shape of a round one (parry RoundShape<S>::inner_shape) */
static int ro_core_shape(int shape) { return (shape >= RO_SHAPE_ROUND_CUBOID && shape <= RO_SHAPE_ROUND_CONVEX_POLYHEDRON) ? (shape == RO_SHAPE_ROUND_CUBOID ? RO_SHAPE_CUBOID : shape - RO_SHAPE_ROUND_CYLINDER + RO_SHAPE_CYLINDER) : shape; }
static void shape_mass_props(const Collider *c, float density, float *mass, v3 *principal_inertia, float frame[4], float com[3]) {
    frame[0] = 0.0f; frame[1] = 0.0f; frame[2] = 0.0f; frame[3] = 1.0f;
    com[0] = com[1] = com[2] = 0.0f;
    if (c->shape == RO_SHAPE_CONVEX_POLYHEDRON) { /* MassProperties::from_convex_polyhedron -> with_inertia_matrix(com, volume * density, tensor * density) */
        const RoPolyhedron *P = c->poly;
        float a[3][3], pi[3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = P->inertia[i][j] * density;
        ro_diagonalise(a, pi, frame);
        *mass = P->volume * density; *principal_inertia = V3(pi[0], pi[1], pi[2]);
        com[0] = P->com.x - P->centre.x; com[1] = P->com.y - P->centre.y; com[2] = P->com.z - P->centre.z; /* (in the recentred collider frame) */
        return;
    }
    if (c->shape == RO_SHAPE_CYLINDER) { /* MassProperties::from_cylinder: cylinder_y_volume_unit_inertia */
        float hh = c->he.y, r = c->radius;
        float vol = hh * r * r * 3.14159265358979323846f * 2.0f;
        float sq_radius = r * r, sq_height = hh * hh * 4.0f;
        float off_principal = (sq_radius * 3.0f + sq_height) / 12.0f;
        float m = vol * density;
        *mass = m; *principal_inertia = V3(off_principal * m, sq_radius / 2.0f * m, off_principal * m);
    } else if (c->shape == RO_SHAPE_CONE) { /* MassProperties::from_cone: cone_y_volume_unit_inertia, centre of mass a quarter of the height above the base */
        float hh = c->he.y, r = c->radius;
        float vol = r * r * 3.14159265358979323846f * hh * 2.0f / 3.0f;
        float sq_radius = r * r, sq_height = hh * hh * 4.0f;
        float off_principal = sq_radius * 3.0f / 20.0f + sq_height * 3.0f / 80.0f;
        float principal = sq_radius * 3.0f / 10.0f;
        float m = vol * density;
        *mass = m; *principal_inertia = V3(off_principal * m, principal * m, off_principal * m);
        com[1] = -hh / 2.0f;
    } else if (c->shape == RO_SHAPE_CUBOID) {
        float vol = c->he.x * c->he.y * c->he.z * 8.0f;
        float m = vol * density;
        float ix = (c->he.y * c->he.y + c->he.z * c->he.z) / 3.0f;
        float iy = (c->he.x * c->he.x + c->he.z * c->he.z) / 3.0f;
        float iz = (c->he.x * c->he.x + c->he.y * c->he.y) / 3.0f;
        *mass = m; *principal_inertia = V3(ix * m, iy * m, iz * m);
    } else if (c->shape == RO_SHAPE_CAPSULE) {
        /* MassProperties::from_capsule: a Y cylinder (cylinder_y_volume_unit_inertia) + a ball split in two caps */
        float hh = c->he.x, r = c->radius;
        float cyl_vol = hh * r * r * 3.14159265358979323846f * 2.0f;
        float sq_radius = r * r, sq_height = hh * hh * 4.0f;
        float off_principal = (sq_radius * 3.0f + sq_height) / 12.0f;
        float ball_vol = 3.14159265358979323846f * r * r * r * 4.0f / 3.0f;
        float ball_i = r * r * 0.4f;
        float cap_mass = (cyl_vol + ball_vol) * density;
        float ix = (off_principal * cyl_vol + ball_i * ball_vol) * density;
        float iy = (sq_radius / 2.0f * cyl_vol + ball_i * ball_vol) * density;
        float h = hh * 2.0f;
        float extra = (h * h * 0.25f + h * r * 3.0f / 8.0f) * ball_vol * density;
        *mass = cap_mass; *principal_inertia = V3(ix + extra, iy, ix + extra);
        /* rotation_between(Y, segment direction): -90 deg about Z for the X axis, +90 deg about X for the Z axis */
        if (c->axis == 0) { frame[2] = -0.70710678118654752f; frame[3] = 0.70710678118654752f; }
        else if (c->axis == 2) { frame[0] = 0.70710678118654752f; frame[3] = 0.70710678118654752f; }
    } else if (c->shape == RO_SHAPE_HALFSPACE) {
        *mass = 0.0f; *principal_inertia = V3(0, 0, 0); /* MassProperties::zero(): an unbounded shape weighs nothing */
    } else {
        float r = c->radius;
        float vol = 3.14159265358979323846f * r * r * r * 4.0f / 3.0f;
        float m = vol * density;
        float i = r * r * 0.4f;
        *mass = m; *principal_inertia = V3(i * m, i * m, i * m);
    }
}
/* radius of the shape's local bounding sphere (centred on the collider origin) — Shape::compute_local_bounding_sphere */
static float shape_bounding_radius_core(const Collider *c);
static float shape_bounding_radius(const Collider *c) { float r = shape_bounding_radius_core(c); return c->border > 0.0f ? r + c->border : r; } /* RoundShape: the inner sphere + the border */
static float shape_bounding_radius_core(const Collider *c) {
    if (c->shape == RO_SHAPE_CUBOID || c->shape == RO_SHAPE_COMPOUND || c->shape == RO_SHAPE_TRIMESH) return vlen(c->he); /* (a composite: the sphere about its local box) */
    if (c->shape == RO_SHAPE_CAPSULE) return c->he.x + c->radius;
    if (c->shape == RO_SHAPE_HALFSPACE) return FLT_MAX;
    if (c->shape == RO_SHAPE_CYLINDER || c->shape == RO_SHAPE_CONE) return sqrtf(c->radius * c->radius + c->he.y * c->he.y);
    if (c->shape == RO_SHAPE_CONVEX_POLYHEDRON) return c->poly->origin_radius; /* about the collider origin (the CCD pre-filter); max_extent uses the point cloud's own sphere */
    return c->radius;
}

/* ---- parry MassProperties algebra (not in /root/reference; restated from its public definition) ----------------
 * A MassProperties value = (mass, local_com, principal inertia, principal frame).  `transform_by(pos)` moves the
 * centre and rotates the frame; `a + b` = total mass, mass-weighted centre, sum of the two inertia tensors shifted
 * to the common centre (parallel-axis theorem), re-diagonalised.  parry diagonalises with nalgebra's
 * symmetric_eigen (Householder + QR); a cyclic Jacobi iteration is used here instead — same eigen-system, rounding
 * differs (unpinned like every other parry quantity).  Arithmetic is plain f32, no contraction. */
typedef struct { float mass; float com[3]; float pi[3]; float frame[4]; } ro_mp;   /* frame: quaternion x,y,z,w */

static void ro_quat_to_rot(const float q[4], float r[3][3]) {
    float x2 = q[0] + q[0], y2 = q[1] + q[1], z2 = q[2] + q[2];
    float xx = q[0] * x2, xy = q[0] * y2, xz = q[0] * z2;
    float yy = q[1] * y2, yz = q[1] * z2, zz = q[2] * z2;
    float wx = q[3] * x2, wy = q[3] * y2, wz = q[3] * z2;
    r[0][0] = 1.0f - (yy + zz); r[0][1] = xy - wz; r[0][2] = xz + wy;
    r[1][0] = xy + wz; r[1][1] = 1.0f - (xx + zz); r[1][2] = yz - wx;
    r[2][0] = xz - wy; r[2][1] = yz + wx; r[2][2] = 1.0f - (xx + yy);
}
/* reconstruct_inertia_matrix: R diag(pi) R^T */
static void ro_inertia_matrix(const ro_mp *m, float out[3][3]) {
    float r[3][3]; ro_quat_to_rot(m->frame, r);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            out[i][j] = r[i][0] * m->pi[0] * r[j][0] + r[i][1] * m->pi[1] * r[j][1] + r[i][2] * m->pi[2] * r[j][2];
}
/* construct_shifted_inertia_matrix: I + (|s|^2 Id - s s^T) * mass */
static void ro_shifted_inertia(const ro_mp *m, const float s[3], float out[3][3]) {
    ro_inertia_matrix(m, out);
    float d = s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            out[i][j] = out[i][j] + ((i == j ? d : 0.0f) - s[i] * s[j]) * m->mass;
}
/* rotation matrix (columns = axes) -> unit quaternion (Shepperd's method) */
static void ro_rot_to_quat(float v[3][3], float q[4]) {
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
static void ro_diagonalise(float a[3][3], float pi[3], float frame[4]) {
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
    ro_rot_to_quat(v, frame);
}
/* MassProperties::transform_by(pose): centre moved, frame rotated */
static void ro_mp_transform(ro_mp *m, const float t[3], const float q[4]) {
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
static void ro_mp_add(ro_mp *acc, const ro_mp *o) {
    if (acc->mass == 0.0f) { *acc = *o; return; }
    if (o->mass == 0.0f) return;
    float m1 = acc->mass, m2 = o->mass, total = m1 + m2, inv = 1.0f / total;
    float com[3], s1[3], s2[3];
    for (int k = 0; k < 3; ++k) com[k] = (acc->com[k] * m1 + o->com[k] * m2) * inv;
    for (int k = 0; k < 3; ++k) { s1[k] = com[k] - acc->com[k]; s2[k] = com[k] - o->com[k]; }
    float i1[3][3], i2[3][3], sum[3][3];
    ro_shifted_inertia(acc, s1, i1); ro_shifted_inertia(o, s2, i2);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) sum[i][j] = i1[i][j] + i2[i][j];
    /* symmetrise exactly (the shifted tensors are symmetric up to rounding) */
    sum[1][0] = sum[0][1]; sum[2][0] = sum[0][2]; sum[2][1] = sum[1][2];
    acc->mass = total; acc->com[0] = com[0]; acc->com[1] = com[1]; acc->com[2] = com[2];
    ro_diagonalise(sum, acc->pi, acc->frame);
}

static int collider_enabled(const Collider *c) { return !(c->memberships == 0 && c->filter == 0); }
/* sum of the attached colliders' mass properties at `density_override` (< 0: each collider's own density) */
static void comp_mass_props(const Collider *c, float density, ro_mp *out); /* ro_composite.h */
static float comp_ccd_thickness(const Collider *c);                           /* ro_composite.h */
static void sum_collider_mass_props(const ro_world *w, int body, float density_override, ro_mp *acc) {
    memset(acc, 0, sizeof(*acc)); acc->frame[3] = 1.0f;
    /* attachment order (rb.colliders(): ascending `ord`) — not index order: a collider may sit in a reused arena slot */
    int cnt = 0, cap = 0, *list = NULL;
    for (int i = 0; i < w->ncolliders; ++i) {
        const Collider *c = &w->colliders[i];
        if (c->parent != body || !collider_enabled(c)) continue;
        if (cnt == cap) { cap = cap ? 2 * cap : 8; list = (int *)realloc(list, sizeof(int) * cap); }
        int k = cnt++;
        while (k > 0 && w->colliders[list[k - 1]].ord > c->ord) { list[k] = list[k - 1]; --k; }
        list[k] = i;
    }
    for (int q = 0; q < cnt; ++q) {
        const Collider *c = &w->colliders[list[q]];
        ro_mp m; memset(&m, 0, sizeof(m)); m.frame[3] = 1.0f;
        if (c->shape == RO_SHAPE_COMPOUND || c->shape == RO_SHAPE_TRIMESH) comp_mass_props(c, density_override < 0.0f ? c->density : density_override, &m);
        else {
        v3 pi; shape_mass_props(c, density_override < 0.0f ? c->density : density_override, &m.mass, &pi, m.frame, m.com);
        m.pi[0] = pi.x; m.pi[1] = pi.y; m.pi[2] = pi.z;
        }
        float t[3] = {c->pos_wrt_parent.t.x, c->pos_wrt_parent.t.y, c->pos_wrt_parent.t.z};
        float q[4] = {c->pos_wrt_parent.r.x, c->pos_wrt_parent.r.y, c->pos_wrt_parent.r.z, c->pos_wrt_parent.r.w};
        ro_mp_transform(&m, t, q);
        ro_mp_add(acc, &m);
    }
    free(list);
}
/* RigidBodyMassProps::recompute_mass_properties_from_colliders — rigid_body_components.rs:421-489: the attached
 * colliders' MassProperties (transformed by pos_wrt_parent) are summed in attachment order, then the additional mass. */
static void recompute_mass_properties(ro_world *w, Body *b) {
    int body = (int)(b - w->bodies);
    ro_mp acc; sum_collider_mass_props(w, body, -1.0f, &acc);
    if (b->additional_mass != 0.0f) {
        if (acc.mass > 0.0f) {
            /* MassProperties::set_mass(prev + add, adjust_angular_inertia = true) */
            float nm = acc.mass + b->additional_mass;
            float k = nm / acc.mass;
            acc.pi[0] = acc.pi[0] * k; acc.pi[1] = acc.pi[1] * k; acc.pi[2] = acc.pi[2] * k; acc.mass = nm;
        } else {
            ro_mp unit; sum_collider_mass_props(w, body, 1.0f, &unit);
            if (unit.mass > 0.0f) {
                float k = b->additional_mass / unit.mass;
                unit.pi[0] = unit.pi[0] * k; unit.pi[1] = unit.pi[1] * k; unit.pi[2] = unit.pi[2] * k; unit.mass = b->additional_mass;
                acc = unit; /* local_mprops (zero) += unit_mprops */
            } else {
                acc.mass = b->additional_mass;
            }
        }
    }
    b->local_com = V3(acc.com[0], acc.com[1], acc.com[2]);
    /* recompute_max_extent (:491-515): bounding spheres of the attached shapes about the local CoM */
    b->max_extent = 0.0f;
    for (int i = 0; i < w->ncolliders; ++i) {
        const Collider *c = &w->colliders[i];
        if (c->parent != body || !collider_enabled(c)) continue;
        float radius = shape_bounding_radius(c);
        v3 centre = c->pos_wrt_parent.t;
        if (c->shape == RO_SHAPE_CONVEX_POLYHEDRON) { /* point_cloud_bounding_sphere: centred on the mean of the points */
            centre = pose_tp(c->pos_wrt_parent, vsub(c->poly->sphere_centre, c->poly->centre)); radius = c->border > 0.0f ? c->poly->sphere_radius + c->border : c->poly->sphere_radius;
        }
        float extent = vlen(vsub(centre, b->local_com)) + radius;
        b->max_extent = ro_maxf(b->max_extent, extent);
    }
    /* RigidBodyCcd::ccd_thickness (rigid_body_components.rs:1227): the thinnest attached shape (Shape::ccd_thickness: ball radius,
     * smallest cuboid half extent, capsule radius; a half-space has none), Real::MAX without colliders */
    b->ccd_thickness = FLT_MAX;
    for (int i = 0; i < w->ncolliders; ++i) {
        const Collider *c = &w->colliders[i];
        if (c->parent != body || !collider_enabled(c) || c->shape == RO_SHAPE_HALFSPACE) continue;
        if (c->shape == RO_SHAPE_TRIMESH) continue; /* a mesh is never swept and stays out of ccd_thickness (sweeps.rs:86-97: shape_never_ccd_swept) */
        if (c->shape == RO_SHAPE_COMPOUND) { b->ccd_thickness = ro_minf(b->ccd_thickness, comp_ccd_thickness(c)); continue; } /* Compound::ccd_thickness: the thinnest part */
        float th = c->shape == RO_SHAPE_BALL ? c->radius : c->shape == RO_SHAPE_CAPSULE ? c->radius : ro_minf(c->he.x, ro_minf(c->he.y, c->he.z));
        if (c->border > 0.0f) th = th + c->border; /* RoundShape::ccd_thickness = inner + border */
        b->ccd_thickness = ro_minf(b->ccd_thickness, th);
    }
    b->inv_mass = ro_inv(acc.mass);
    b->inv_principal_inertia = V3(ro_inv(acc.pi[0]), ro_inv(acc.pi[1]), ro_inv(acc.pi[2]));
    b->principal_frame = Q(acc.frame[0], acc.frame[1], acc.frame[2], acc.frame[3]);
    update_world_mass_properties(b);
}

static void purge_dead_pairs(ro_world *w);
static void gen_reserve(uint32_t **gen, int *cap, int n) {
    if (n <= *cap) return;
    int nc = *cap ? *cap : 1024; while (nc < n) nc *= 2;
    *gen = (uint32_t *)realloc(*gen, sizeof(uint32_t) * nc);
    memset(*gen + *cap, 0, sizeof(uint32_t) * (nc - *cap));
    *cap = nc;
}
static void free_push(int **list, int *n, int *cap, int v) {
    if (*n == *cap) { *cap = *cap ? *cap * 2 : 64; *list = (int *)realloc(*list, sizeof(int) * *cap); }
    (*list)[(*n)++] = v;
}
uint32_t ro_body_generation(const ro_world *w, int32_t body) { return (body >= 0 && body < w->cap_body_gen) ? w->body_gen[body] : 0u; }
uint32_t ro_collider_generation(const ro_world *w, int32_t collider) { return (collider >= 0 && collider < w->cap_coll_gen) ? w->coll_gen[collider] : 0u; }
int32_t ro_add_body(ro_world *w, const ro_body_desc *d) {
    /* Arena::insert (arena.rs:260-290): the head of the free list — the slot removed last — before any fresh index */
    int idx = w->nbodies, reused = 0;
    if (w->nbody_free > 0 && !getenv("RP_NO_ARENA_REUSE")) {
        purge_dead_pairs(w);
        idx = w->body_free[--w->nbody_free]; reused = 1;
        /* the removed colliders of the slot's previous occupant no longer name it */
        for (int i = 0; i < w->ncolliders; ++i) if (w->colliders[i].parent == idx && w->colliders[i].memberships == 0 && w->colliders[i].filter == 0) w->colliders[i].parent = -1;
        if (idx < w->cap_masks) memset(&w->color_masks[idx], 0, sizeof(w->color_masks[idx]));
    } else if (w->nbodies == w->cap_bodies) {
        w->cap_bodies = w->cap_bodies ? w->cap_bodies * 2 : 1024;
        w->bodies = (Body *)realloc(w->bodies, sizeof(Body) * w->cap_bodies);
    }
    gen_reserve(&w->body_gen, &w->cap_body_gen, idx + 1);
    w->body_gen[idx] = w->body_arena_gen;
    Body *b = &w->bodies[idx];
    memset(b, 0, sizeof(*b));
    b->body_type = d->body_type;
    b->position.t = V3(d->translation[0], d->translation[1], d->translation[2]);
    b->position.r = qnormalize(Q(d->rotation[0], d->rotation[1], d->rotation[2], d->rotation[3]));
    b->next_position = b->position;
    b->linvel = V3(d->linvel[0], d->linvel[1], d->linvel[2]);
    b->angvel = V3(d->angvel[0], d->angvel[1], d->angvel[2]);
    b->linear_damping = d->linear_damping; b->angular_damping = d->angular_damping;
    b->gravity_scale = d->gravity_scale; b->additional_mass = d->additional_mass;
    b->dominance = d->dominance; b->gyroscopic = d->gyroscopic; b->allow_fast_rotation = d->allow_fast_rotation;
    b->locked_axes = d->locked_axes & 0x3fu;
    b->principal_frame = qident();
    b->solver_id = RO_NO_BODY;
    /* RigidBodyActivation::active() / cannot_sleep() — rigid_body_components.rs:1354-1385 */
    b->normalized_linear_threshold = d->can_sleep ? 0.05f : -1.0f;
    b->angular_threshold = d->can_sleep ? 0.5f : -1.0f;
    b->time_until_sleep = 0.5f; b->time_since_can_sleep = 0.0f; b->sleeping = 0;
    b->sleep_prev_pose = pose_ident(); b->island_id = -1; b->slept_at = 0;
    recompute_mass_properties(w, b);
    if (!reused) w->nbodies++;
    pi_ensure_body(w, idx); /* IslandManager::rigid_body_updated -> PersistentIslands::ensure_body (manager.rs:303) */
    return idx;
}

struct RoComposite; static v3 comp_half(const struct RoComposite *C); static v3 comp_centre(const struct RoComposite *C); /* ro_composite.h */
int32_t ro_add_collider(ro_world *w, const ro_collider_desc *d, int32_t parent) {
    int idx = w->ncolliders, reused = 0;
    if (w->ncoll_free > 0 && !getenv("RP_NO_ARENA_REUSE")) { purge_dead_pairs(w); idx = w->coll_free[--w->ncoll_free]; reused = 1; }
    else if (w->ncolliders == w->cap_colliders) {
        w->cap_colliders = w->cap_colliders ? w->cap_colliders * 2 : 1024;
        w->colliders = (Collider *)realloc(w->colliders, sizeof(Collider) * w->cap_colliders);
    }
    gen_reserve(&w->coll_gen, &w->cap_coll_gen, idx + 1);
    w->coll_gen[idx] = w->coll_arena_gen;
    Collider *c = &w->colliders[idx];
    memset(c, 0, sizeof(*c));
    c->parent = parent; c->sub = w->cur_sub;
    c->shape = ro_core_shape(d->shape); c->border = (d->shape >= RO_SHAPE_ROUND_CUBOID && d->shape <= RO_SHAPE_ROUND_CONVEX_POLYHEDRON) ? d->border_radius : 0.0f; /* a round shape = its inner shape + a border radius */
    c->he = V3(d->half_extents[0], d->half_extents[1], d->half_extents[2]);
    c->radius = d->half_extents[0];
    if (c->shape == RO_SHAPE_CAPSULE) { c->radius = d->half_extents[1]; c->axis = (int)d->half_extents[2]; if (c->axis < 0 || c->axis > 2) c->axis = 1; }
    if (c->shape == RO_SHAPE_CYLINDER || c->shape == RO_SHAPE_CONE) { c->radius = d->half_extents[1]; c->he = V3(c->radius, d->half_extents[0], c->radius); c->axis = 1; } /* he = the local AABB's half extents */
    if (c->shape == RO_SHAPE_CONVEX_POLYHEDRON) { c->poly = w->polys[(int)d->half_extents[0]]; c->he = c->poly->half; c->radius = 0.0f; c->axis = 1; }
    if (d->shape == RO_SHAPE_COMPOUND || d->shape == RO_SHAPE_TRIMESH) { c->shape = d->shape; c->border = 0.0f; c->comp = w->comps[(int)d->half_extents[0]]; c->he = comp_half(c->comp); c->radius = 0.0f; c->axis = 1; }
    c->pos_wrt_parent.t = V3(d->translation[0], d->translation[1], d->translation[2]);
    c->pos_wrt_parent.r = qnormalize(Q(d->rotation[0], d->rotation[1], d->rotation[2], d->rotation[3]));
    if (c->shape == RO_SHAPE_CONVEX_POLYHEDRON) c->pos_wrt_parent.t = vadd(qrot(c->pos_wrt_parent.r, c->poly->centre), c->pos_wrt_parent.t); /* the recentring offset rides in the pose */
    if (c->comp) c->pos_wrt_parent.t = vadd(qrot(c->pos_wrt_parent.r, comp_centre(c->comp)), c->pos_wrt_parent.t); /* likewise: a composite is stored recentred on its local AABB */
    c->density = d->density; c->friction = d->friction; c->restitution = d->restitution;
    c->friction_rule = d->friction_rule; c->restitution_rule = d->restitution_rule;
    c->memberships = d->collision_memberships; c->filter = d->collision_filter;
    c->active_events = d->active_events; c->force_threshold = d->contact_force_event_threshold;
    c->ord = parent >= 0 ? w->bodies[parent].ncolliders++ : w->nfree_colliders++;
    if (parent >= 0) c->pos = pose_mul(w->bodies[parent].position, c->pos_wrt_parent);
    else c->pos = c->pos_wrt_parent;
    if (!reused) w->ncolliders++;
    if (parent >= 0) recompute_mass_properties(w, &w->bodies[parent]);
    w->bp_dirty = 1;
    return idx;
}

/* everything added from now on belongs to a new sub-world (the twin of rp_world_begin_subworld) */
int32_t ro_begin_subworld(ro_world *w) { if (w->ncolliders == 0 && w->nbodies == 0 && w->n_sub == 1) return 0; w->cur_sub = w->n_sub++; return w->cur_sub; }
int32_t ro_num_bodies(const ro_world *w) { return w->nbodies; }
void ro_read_bodies(const ro_world *w, float *pos7, float *vel6) {
    for (int i = 0; i < w->nbodies; ++i) {
        const Body *b = &w->bodies[i];
        if (pos7) {
            float *p = pos7 + 7 * i;
            p[0] = b->position.t.x; p[1] = b->position.t.y; p[2] = b->position.t.z;
            p[3] = b->position.r.x; p[4] = b->position.r.y; p[5] = b->position.r.z; p[6] = b->position.r.w;
        }
        if (vel6) {
            float *v = vel6 + 6 * i;
            v[0] = b->linvel.x; v[1] = b->linvel.y; v[2] = b->linvel.z;
            v[3] = b->angvel.x; v[4] = b->angvel.y; v[5] = b->angvel.z;
        }
    }
}
static void wake_request(ro_world *w, int body, int strong);
void ro_set_body_vel(ro_world *w, int32_t body, const float lv[3], const float av[3]) {
    w->bodies[body].linvel = V3(lv[0], lv[1], lv[2]);
    w->bodies[body].angvel = V3(av[0], av[1], av[2]);
    wake_request(w, body, 1); /* set_linvel(.., wake_up = true) -> RigidBody::wake_up(true) */
}
static void bp_set_aabb(ro_world *w, Collider *c);
static void update_world_mass_properties(Body *b);
/* RigidBody::set_position(.., wake_up = true) + user_changes.rs: world mass properties, collider poses and AABBs follow;
 * the body is woken and so is every body it has a contact pair with (pair_management.rs:236-258). */
void ro_set_body_pose(ro_world *w, int32_t body, const float pos7[7]) {
    Body *b = &w->bodies[body];
    b->position.t = V3(pos7[0], pos7[1], pos7[2]);
    b->position.r = Q(pos7[3], pos7[4], pos7[5], pos7[6]);
    b->next_position = b->position;
    update_world_mass_properties(b);
    for (int i = 0; i < w->ncolliders; ++i) {
        Collider *c = &w->colliders[i];
        if (c->parent != body) continue;
        c->pos = pose_mul(b->position, c->pos_wrt_parent);
        bp_set_aabb(w, c);
    }
    wake_request(w, body, 1);
    for (int i = 0; i < w->npairs; ++i) {
        const Pair *p = &w->pairs[i];
        if (!p->alive) continue;
        int b1 = w->colliders[p->c1].parent, b2 = w->colliders[p->c2].parent;
        if (b1 == body) wake_request(w, b2, 1);
        if (b2 == body) wake_request(w, b1, 1);
    }
    /* a moved FIXED body wakes its joint partners (user_changes.rs:228-246): it is no island member, so its own wake is a no-op */
    if (b->body_type == RO_BODY_FIXED)
        for (int i = 0; i < w->njoints; ++i) {
            const Joint *j = &w->joints[i];
            if (j->removed) continue;
            if (j->body1 == body) wake_request(w, j->body2, 1);
            if (j->body2 == body) wake_request(w, j->body1, 1);
        }
}
void ro_set_next_kinematic_position(ro_world *w, int32_t body, const float pos7[7]) {
    Body *b = &w->bodies[body];
    if (b->body_type != RO_BODY_KINEMATIC_POSITION && b->body_type != RO_BODY_KINEMATIC_VELOCITY) return;
    pose np; np.t = V3(pos7[0], pos7[1], pos7[2]); np.r = Q(pos7[3], pos7[4], pos7[5], pos7[6]);
    b->next_position = np;
    const pose *p = &b->position;
    if (p->t.x != np.t.x || p->t.y != np.t.y || p->t.z != np.t.z || p->r.x != np.r.x || p->r.y != np.r.y || p->r.z != np.r.z || p->r.w != np.r.w)
        wake_request(w, body, 1);
}
int32_t ro_collision_events_drain(ro_world *w, int32_t cap, int32_t *out5) {
    int n = w->ncol_events;
    for (int i = 0; i < n && i < cap; ++i) memcpy(out5 + 5 * i, w->col_events + 5 * i, sizeof(int32_t) * 5);
    if (out5 || cap == 0) { if (out5) w->ncol_events = 0; }
    return n;
}
int32_t ro_force_events_drain(ro_world *w, int32_t cap, int32_t *meta4, float *vals8) {
    int n = w->nforce_events;
    for (int i = 0; i < n && i < cap; ++i) { memcpy(meta4 + 4 * i, w->force_meta + 4 * i, sizeof(int32_t) * 4); memcpy(vals8 + 8 * i, w->force_vals + 8 * i, sizeof(float) * 8); }
    if (meta4) w->nforce_events = 0;
    return n;
}
/* RigidBodyMassProps::local_mprops: inv_mass, local_com xyz, inv_principal_inertia xyz, principal frame xyzw */
void ro_body_mass_props(const ro_world *w, int32_t body, float out11[11]) {
    const Body *b = &w->bodies[body];
    out11[0] = b->inv_mass; out11[1] = b->local_com.x; out11[2] = b->local_com.y; out11[3] = b->local_com.z;
    out11[4] = b->inv_principal_inertia.x; out11[5] = b->inv_principal_inertia.y; out11[6] = b->inv_principal_inertia.z;
    out11[7] = b->principal_frame.x; out11[8] = b->principal_frame.y; out11[9] = b->principal_frame.z; out11[10] = b->principal_frame.w;
}
void ro_add_force(ro_world *w, int32_t body, const float force[3], const float torque[3], int32_t reset) {
    Body *b = &w->bodies[body];
    if (reset) {
        if (b->user_force.x != 0.0f || b->user_force.y != 0.0f || b->user_force.z != 0.0f) { b->user_force = V3(0, 0, 0); wake_request(w, body, 1); }
        if (b->user_torque.x != 0.0f || b->user_torque.y != 0.0f || b->user_torque.z != 0.0f) { b->user_torque = V3(0, 0, 0); wake_request(w, body, 1); }
    }
    if (b->body_type != RO_BODY_DYNAMIC) return;
    if (force && (force[0] != 0.0f || force[1] != 0.0f || force[2] != 0.0f)) { b->user_force = vadd(b->user_force, V3(force[0], force[1], force[2])); wake_request(w, body, 1); }
    if (torque && (torque[0] != 0.0f || torque[1] != 0.0f || torque[2] != 0.0f)) { b->user_torque = vadd(b->user_torque, V3(torque[0], torque[1], torque[2])); wake_request(w, body, 1); }
}
void ro_apply_impulse(ro_world *w, int32_t body, const float impulse[3], const float torque_impulse[3]) {
    Body *b = &w->bodies[body];
    if (b->body_type != RO_BODY_DYNAMIC) return;
    if (impulse && (impulse[0] != 0.0f || impulse[1] != 0.0f || impulse[2] != 0.0f)) {
        b->linvel = vadd(b->linvel, vcmul(V3(impulse[0], impulse[1], impulse[2]), b->effective_inv_mass));
        wake_request(w, body, 1);
    }
    if (torque_impulse && (torque_impulse[0] != 0.0f || torque_impulse[1] != 0.0f || torque_impulse[2] != 0.0f)) {
        b->angvel = vadd(b->angvel, sym3_mul(b->effective_world_inv_inertia, V3(torque_impulse[0], torque_impulse[1], torque_impulse[2])));
        wake_request(w, body, 1);
    }
}
void ro_wake_up(ro_world *w, int32_t body, int32_t strong) { if (body >= 0 && body < w->nbodies) wake_request(w, body, strong); }
void ro_read_sleeping(const ro_world *w, int32_t *sleeping) {
    for (int i = 0; i < w->nbodies; ++i) sleeping[i] = w->bodies[i].body_type != RO_BODY_FIXED && w->bodies[i].sleeping;
}

/* ------------------------------------------------------------------------------------ */
/* Broad phase.  Contract = the pair SET of the reference's fat-AABB BVH
 * (geometry/broad_phase_bvh/mod.rs:171-263, update.rs:35-602): each leaf keeps an AABB
 * fattened by CHANGE_DETECTION_FACTOR (0.04) that is only rewritten when the tight
 * collision AABB (shape AABB loosened by contact_skin + prediction/2, collider.rs:553-557)
 * leaves it; a pair exists exactly while the two fat AABBs intersect and the pair passes the
 * filters of update.rs:334-396.  The tree itself is free to differ: here a
 * sort-and-sweep over fat AABBs, re-run only when some fat AABB changed. */
static Aabb collider_collision_aabb(const Collider *c, float loosen) {
    Aabb a;
    if (c->shape == RO_SHAPE_CUBOID || c->shape >= RO_SHAPE_CYLINDER) { /* Cylinder / Cone::aabb = local_aabb().transform_by(pos); a polyhedron's local box likewise (ro_polyhedron.h) */
        float m[3][3]; quat_to_mat(c->pos.r, m);
        v3 h = V3(fabsf(m[0][0]) * c->he.x + fabsf(m[0][1]) * c->he.y + fabsf(m[0][2]) * c->he.z,
                  fabsf(m[1][0]) * c->he.x + fabsf(m[1][1]) * c->he.y + fabsf(m[1][2]) * c->he.z,
                  fabsf(m[2][0]) * c->he.x + fabsf(m[2][1]) * c->he.y + fabsf(m[2][2]) * c->he.z);
        a.mins = vsub(c->pos.t, h); a.maxs = vadd(c->pos.t, h);
    } else if (c->shape == RO_SHAPE_CAPSULE) {
        /* Capsule::aabb: the transformed segment's box loosened by the radius */
        v3 e = capsule_axis_dir(c->axis);
        v3 pa = pose_tp(c->pos, vmul(e, -c->he.x)), pb = pose_tp(c->pos, vmul(e, c->he.x));
        v3 r = V3(c->radius, c->radius, c->radius);
        a.mins = vsub(V3(ro_minf(pa.x, pb.x), ro_minf(pa.y, pb.y), ro_minf(pa.z, pb.z)), r);
        a.maxs = vadd(V3(ro_maxf(pa.x, pb.x), ro_maxf(pa.y, pb.y), ro_maxf(pa.z, pb.z)), r);
    } else if (c->shape == RO_SHAPE_HALFSPACE) {
        /* HalfSpace::aabb: half of the float range in every direction, wherever the plane is ("so that we can still loosen it") */
        v3 h = V3(FLT_MAX / 2.0f, FLT_MAX / 2.0f, FLT_MAX / 2.0f);
        a.mins = vneg(h); a.maxs = h;
    } else {
        v3 h = V3(c->radius, c->radius, c->radius);
        a.mins = vsub(c->pos.t, h); a.maxs = vadd(c->pos.t, h);
    }
    if (c->border > 0.0f) { v3 bb = V3(c->border, c->border, c->border); a.mins = vsub(a.mins, bb); a.maxs = vadd(a.maxs, bb); } /* RoundShape::aabb = inner.aabb(pos).loosened(border_radius) */
    v3 l = V3(loosen, loosen, loosen);
    a.mins = vsub(a.mins, l); a.maxs = vadd(a.maxs, l);
    return a;
}
static int aabb_contains(const Aabb *o, const Aabb *i) {
    return o->mins.x <= i->mins.x && o->mins.y <= i->mins.y && o->mins.z <= i->mins.z &&
           o->maxs.x >= i->maxs.x && o->maxs.y >= i->maxs.y && o->maxs.z >= i->maxs.z;
}
static int aabb_intersects(const Aabb *a, const Aabb *b) {
    return a->mins.x <= b->maxs.x && b->mins.x <= a->maxs.x && a->mins.y <= b->maxs.y && b->mins.y <= a->maxs.y &&
           a->mins.z <= b->maxs.z && b->mins.z <= a->maxs.z;
}
/* BroadPhaseBvh::set_aabb / insert_with_change_detection */
static void bp_set_aabb(ro_world *w, Collider *c) {
    float prediction = w->params.normalized_prediction_distance * w->params.length_unit;
    Aabb tight = collider_collision_aabb(c, prediction / 2.0f);
    if (c->has_fat && aabb_contains(&c->fat, &tight)) return;
    float skin = 4.0e-2f * w->params.length_unit;
    v3 l = V3(skin, skin, skin);
    c->fat.mins = vsub(tight.mins, l); c->fat.maxs = vadd(tight.maxs, l);
    c->has_fat = 1;
    w->bp_dirty = 1;
}

static uint64_t hash64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
static int map_find(const ro_world *w, int64_t key) {
    uint64_t h = hash64((uint64_t)key) & (uint64_t)(w->map_cap - 1);
    while (w->map_keys[h] != -1) { if (w->map_keys[h] == key) return w->map_vals[h]; h = (h + 1) & (uint64_t)(w->map_cap - 1); }
    return -1;
}
static void map_insert_raw(int64_t *keys, int *vals, int cap, int64_t key, int val) {
    uint64_t h = hash64((uint64_t)key) & (uint64_t)(cap - 1);
    while (keys[h] != -1) h = (h + 1) & (uint64_t)(cap - 1);
    keys[h] = key; vals[h] = val;
}
static void map_rebuild(ro_world *w) {
    int need = 1 << 12; while (need < 4 * (w->npairs + 16)) need <<= 1;
    if (need != w->map_cap) {
        w->map_cap = need;
        w->map_keys = (int64_t *)realloc(w->map_keys, sizeof(int64_t) * need);
        w->map_vals = (int *)realloc(w->map_vals, sizeof(int) * need);
    }
    for (int i = 0; i < w->map_cap; ++i) w->map_keys[i] = -1;
    for (int i = 0; i < w->npairs; ++i)
        map_insert_raw(w->map_keys, w->map_vals, w->map_cap, ((int64_t)w->pairs[i].c1 << 32) | (uint32_t)w->pairs[i].c2, i);
}

static int body_is_dynamic(const ro_world *w, int parent) { return parent >= 0 && w->bodies[parent].body_type == RO_BODY_DYNAMIC; }
/* member of the active set: an awake dynamic body (IslandManager::active_bodies) */
static int body_is_active(const ro_world *w, int body) { return body >= 0 && w->bodies[body].body_type != RO_BODY_FIXED && !w->bodies[body].sleeping; } /* dynamic or kinematic, awake */
static int body_is_dyn_awake(const ro_world *w, int body) { return body >= 0 && w->bodies[body].body_type == RO_BODY_DYNAMIC && !w->bodies[body].sleeping; }
static int body_is_sleeping_nonfixed(const ro_world *w, int body) { return body >= 0 && w->bodies[body].body_type != RO_BODY_FIXED && w->bodies[body].sleeping; }
/* pair_solver_hints count cleared by clear_asleep_pair_solver_hint_counts_of (solver_graph.rs:21-49): one of the
 * pair's bodies fell asleep after the hint was last computed */
static int pair_hint_cleared(const ro_world *w, const Pair *p) {
    int b1 = w->colliders[p->c1].parent, b2 = w->colliders[p->c2].parent;
    int s1 = b1 >= 0 ? w->bodies[b1].slept_at : 0, s2 = b2 >= 0 ? w->bodies[b2].slept_at : 0;
    return (s1 > s2 ? s1 : s2) >= p->hint_seq;
}
/* for_each_desired_manifold (solver_graph.rs:517-571) + qualify_manifold_bqi (:101-124) */
static int pair_selected(const ro_world *w, const Pair *p) {
    if (!p->alive || p->nsc == 0 || p->color == RO_COLOR_UNCOLORED) return 0;
    if (pair_hint_cleared(w, p)) return 0;
    /* PAIR_HINT_DYN_BIT + qualify_manifold_bqi: at least one awake DYNAMIC body */
    return body_is_dyn_awake(w, w->colliders[p->c1].parent) || body_is_dyn_awake(w, w->colliders[p->c2].parent);
}

/* update.rs:334-396 pair filter (same parent, collision types, groups) */
static int bp_pair_allowed(const ro_world *w, const Collider *a, const Collider *b) {
    if (a->sub != b->sub) return 0; /* batched small worlds (rapier_hip.h rp_world_begin_subworld): as a filter_contact_pair hook would */
    if (a->parent >= 0 && a->parent == b->parent) return 0;
    if (!body_is_dynamic(w, a->parent) && !body_is_dynamic(w, b->parent)) return 0; /* ActiveCollisionTypes::default */
    if (!((a->memberships & b->filter) != 0 && (b->memberships & a->filter) != 0)) return 0;
    return 1;
}

typedef struct { float minx; int idx; } SweepItem;
static int sweep_cmp(const void *a, const void *b) {
    const SweepItem *x = (const SweepItem *)a, *y = (const SweepItem *)b;
    if (x->minx < y->minx) return -1; if (x->minx > y->minx) return 1; return x->idx - y->idx;
}
static void clear_pair_solver_color(ro_world *w, Pair *p);

/* EventHandler::handle_collision_event — event_handler.rs:94-130 (collected, not called back) */
static void push_collision_event(ro_world *w, int c1, int c2, int started, int flags) {
    if (w->ncol_events == w->cap_col_events) {
        w->cap_col_events = w->cap_col_events ? 2 * w->cap_col_events : 256;
        w->col_events = (int32_t *)realloc(w->col_events, sizeof(int32_t) * 5 * (size_t)w->cap_col_events);
    }
    int32_t *e = w->col_events + 5 * w->ncol_events++;
    e[0] = c1; e[1] = c2; e[2] = started; e[3] = flags; e[4] = w->step_seq;
}

/* ---- sleeping: IslandManager::wake_up (island_manager/sleep.rs:31-79) --------------------------------
 * Requests are collected per body (1 = weak, 2 = strong) and applied island-wide by apply_wakes: waking
 * any body of a sleeping island wakes the whole island with a strong timer reset for every member; a
 * strong wake of an awake body only resets its own timer. */
static void wake_request(ro_world *w, int body, int strong) {
    if (body < 0 || w->bodies[body].body_type == RO_BODY_FIXED) return;
    int lvl = strong ? 2 : 1;
    if (w->bodies[body].wake_req < lvl) w->bodies[body].wake_req = lvl;
}
static void apply_wakes(ro_world *w) {
    int any = 0;
    for (int i = 0; i < w->nbodies; ++i) if (w->bodies[i].wake_req) { any = 1; break; }
    if (!any) return;
    for (int i = 0; i < w->nbodies; ++i) {
        Body *b = &w->bodies[i];
        if (!b->wake_req) continue;
        if (b->sleeping && b->island_id >= 0) w->isl[b->island_id].sleeping = 2; /* marked: the whole persistent island wakes (sleep.rs:44-70) */
        else if (b->wake_req == 2) b->time_since_can_sleep = 0.0f; /* RigidBodyActivation::wake_up(strong) */
        b->wake_req = 0;
    }
    for (int i = 0; i < w->nbodies; ++i) {
        Body *b = &w->bodies[i];
        if (b->body_type != RO_BODY_FIXED && b->sleeping && b->island_id >= 0 && w->isl[b->island_id].sleeping == 2) { b->sleeping = 0; b->time_since_can_sleep = 0.0f; }
    }
    for (int i = 0; i < w->isl_next; ++i) if (w->isl[i].used && w->isl[i].sleeping == 2) w->isl[i].sleeping = 0;
}

static void clear_pair_solver_color(ro_world *w, Pair *p);
static void delete_pair_effects(ro_world *w, Pair *p) {
    const Collider *a = &w->colliders[p->c1], *b = &w->colliders[p->c2];
    /* remove_pair wakes the bodies of a touching pair (pair_management.rs:541-552); remove_collider wakes
     * every body that had a pair with the removed collider (:88-99) */
    int gone = (a->memberships == 0 && a->filter == 0) || (b->memberships == 0 && b->filter == 0);
    if (p->nsc > 0 || gone) { wake_request(w, a->parent, 1); wake_request(w, b->parent, 1); }
    if (p->nsc > 0) pi_journal(w, a->parent, b->parent, 1, ((uint64_t)(uint32_t)p->c1 << 32) | (uint32_t)p->c2); /* unlink_contact (pair_management.rs:531) */
    /* Stopped event of a touching pair: remove_pair (pair_management.rs:554-558), remove_collider (:101-110, REMOVED flag) */
    if (p->nsc > 0 && ((a->active_events | b->active_events) & 1u)) push_collision_event(w, p->c1, p->c2, 0, gone ? 2 : 0);
    if (p->intersecting && ((a->active_events | b->active_events) & 1u)) push_collision_event(w, p->c1, p->c2, 0, (gone ? 2 : 0) | 1); /* remove_pair / remove_collider on the intersection graph */
    p->intersecting = 0;
    clear_pair_solver_color(w, p);
}
static void broad_phase_update(ro_world *w) {
    for (int i = 0; i < w->ncolliders; ++i) if (!w->colliders[i].has_fat) bp_set_aabb(w, &w->colliders[i]);
    w->stats.bp_rebuilt = 0;
    if (!w->bp_dirty) return;
    w->bp_dirty = 0; w->stats.bp_rebuilt = 1;
    int n = w->ncolliders;
    SweepItem *items = (SweepItem *)malloc(sizeof(SweepItem) * (n + 1));
    for (int i = 0; i < n; ++i) { items[i].minx = w->colliders[i].fat.mins.x; items[i].idx = i; }
    qsort(items, n, sizeof(SweepItem), sweep_cmp);
    for (int i = 0; i < w->npairs; ++i) w->pairs[i].alive = 0;
    for (int a = 0; a < n; ++a) {
        const Collider *ca = &w->colliders[items[a].idx];
        for (int b = a + 1; b < n && items[b].minx <= ca->fat.maxs.x; ++b) {
            const Collider *cb = &w->colliders[items[b].idx];
            if (!aabb_intersects(&ca->fat, &cb->fat)) continue;
            if (!bp_pair_allowed(w, ca, cb)) continue;
            int c1 = items[a].idx < items[b].idx ? items[a].idx : items[b].idx;
            int c2 = items[a].idx < items[b].idx ? items[b].idx : items[a].idx;
            int64_t key = ((int64_t)c1 << 32) | (uint32_t)c2;
            int pi = map_find(w, key);
            if (pi >= 0) { w->pairs[pi].alive = 1; continue; }
            if (w->npairs == w->cap_pairs) {
                w->cap_pairs = w->cap_pairs ? w->cap_pairs * 2 : 4096;
                w->pairs = (Pair *)realloc(w->pairs, sizeof(Pair) * w->cap_pairs);
            }
            Pair *p = &w->pairs[w->npairs];
            memset(p, 0, sizeof(*p));
            p->c1 = c1; p->c2 = c2; p->alive = 1; p->color = RO_COLOR_UNCOLORED; p->plain_sub[0] = p->plain_sub[1] = -1;
            p->color_bodies[0] = p->color_bodies[1] = RO_NO_BODY;
            p->solver_body_ids[0] = p->solver_body_ids[1] = RO_NO_BODY;
            w->npairs++;
            if (4 * (w->npairs + 16) > w->map_cap) map_rebuild(w);
            else map_insert_raw(w->map_keys, w->map_vals, w->map_cap, key, w->npairs - 1);
        }
    }
    free(items);
    /* DeletePair: NarrowPhase::remove_pair (pair_management.rs:382) frees the colour and drops the edge. */
    int out = 0, removed = 0;
    for (int i = 0; i < w->npairs; ++i) {
        if (!w->pairs[i].alive) { delete_pair_effects(w, &w->pairs[i]); free(w->pairs[i].ex); w->pairs[i].ex = NULL; removed = 1; continue; }
        if (out != i) w->pairs[out] = w->pairs[i];
        out++;
    }
    w->npairs = out;
    if (removed) map_rebuild(w);
    w->dead_pairs = 0;
}
/* NarrowPhase::handle_user_changes for removed colliders (pair_management.rs:24-203) ahead of time: the pairs of every removed
 * collider leave the pair set NOW, with the effects the next step's broad-phase pass would have had (events stamped with that step).
 * Called before an arena slot is handed out again — the reference removes the pairs of a removed collider by HANDLE (index +
 * generation) at the start of the next step; here a pair names its colliders by index alone, so it must not outlive the slot. */
static void purge_dead_pairs(ro_world *w) {
    if (!w->dead_pairs) return;
    w->dead_pairs = 0;
    w->step_seq++;
    int out = 0, removed = 0;
    for (int i = 0; i < w->npairs; ++i) {
        Pair *p = &w->pairs[i];
        const Collider *a = &w->colliders[p->c1], *b = &w->colliders[p->c2];
        if ((a->memberships == 0 && a->filter == 0) || (b->memberships == 0 && b->filter == 0)) { delete_pair_effects(w, p); free(p->ex); p->ex = NULL; removed = 1; continue; }
        if (out != i) w->pairs[out] = w->pairs[i];
        out++;
    }
    w->npairs = out;
    if (removed) map_rebuild(w);
    w->step_seq--;
}

/* ------------------------------------------------------------------------------------ */
/* Pair colouring — geometry/narrow_phase/mod.rs:90-172 */
static void masks_reserve(ro_world *w, int n) {
    if (n <= w->cap_masks) return;
    int nc = w->cap_masks ? w->cap_masks : 1024; while (nc < n) nc *= 2;
    w->color_masks = (u128 *)realloc(w->color_masks, sizeof(u128) * nc);
    memset(w->color_masks + w->cap_masks, 0, sizeof(u128) * (nc - w->cap_masks));
    w->cap_masks = nc;
}
static int u128_test(u128 m, int bit) { return bit < 64 ? (int)((m.lo >> bit) & 1) : (int)((m.hi >> (bit - 64)) & 1); }
static void u128_set(u128 *m, int bit) { if (bit < 64) m->lo |= 1ULL << bit; else m->hi |= 1ULL << (bit - 64); }
static void u128_clear(u128 *m, int bit) { if (bit < 64) m->lo &= ~(1ULL << bit); else m->hi &= ~(1ULL << (bit - 64)); }

static void clear_pair_solver_color(ro_world *w, Pair *p) {
    if (p->color < RO_COLOR_OVERFLOW)
        for (int k = 0; k < 2; ++k)
            if (p->color_bodies[k] != RO_NO_BODY && (int)p->color_bodies[k] < w->cap_masks)
                u128_clear(&w->color_masks[p->color_bodies[k]], p->color);
    p->color = RO_COLOR_UNCOLORED;
    p->color_bodies[0] = p->color_bodies[1] = RO_NO_BODY;
}
static void assign_pair_solver_color(ro_world *w, Pair *p, int body1, int body2) {
    if (p->color != RO_COLOR_UNCOLORED) return;
    uint32_t i1 = (body1 >= 0 && w->bodies[body1].body_type != RO_BODY_FIXED) ? (uint32_t)body1 : RO_NO_BODY;
    uint32_t i2 = (body2 >= 0 && w->bodies[body2].body_type != RO_BODY_FIXED) ? (uint32_t)body2 : RO_NO_BODY;
    masks_reserve(w, w->nbodies + 1);
    int color; uint32_t cb[2];
    if (i1 != RO_NO_BODY && i2 != RO_NO_BODY) {
        u128 a = w->color_masks[i1], b = w->color_masks[i2];
        u128 m = {a.lo | b.lo, a.hi | b.hi};
        color = 128;
        for (int c = 0; c < RO_DYNAMIC_COLOR_COUNT; ++c) if (!u128_test(m, c)) { color = c; break; }
        cb[0] = i1; cb[1] = i2;
    } else if (i1 != RO_NO_BODY || i2 != RO_NO_BODY) {
        uint32_t i = i1 != RO_NO_BODY ? i1 : i2;
        u128 m = w->color_masks[i];
        color = 128;
        for (int c = 127; c >= 0; --c) if (!u128_test(m, c)) { color = c; break; }
        cb[0] = i; cb[1] = RO_NO_BODY;
    } else {
        p->color = RO_COLOR_OVERFLOW; p->color_bodies[0] = p->color_bodies[1] = RO_NO_BODY; return;
    }
    if (color >= 128) { p->color = RO_COLOR_OVERFLOW; p->color_bodies[0] = p->color_bodies[1] = RO_NO_BODY; return; }
    for (int k = 0; k < 2; ++k) if (cb[k] != RO_NO_BODY) u128_set(&w->color_masks[cb[k]], color);
    p->color = (uint8_t)color; p->color_bodies[0] = cb[0]; p->color_bodies[1] = cb[1];
}

/* ------------------------------------------------------------------------------------ */
/* Narrow phase: pair_update::process_pair — geometry/narrow_phase/pair_update.rs:67-680 */
static float relative_rot_cos(quat base, quat cur) { float c = qdot(base, cur); return 2.0f * c * c - 1.0f; }
static float relative_pose_drift(pose base, pose cur, float max_extent) {
    float trans = vlen(vsub(cur.t, base.t));
    quat d = qmul(cur.r, qconj(base.r));
    float chord = 2.0f * vlen(V3(d.x, d.y, d.z)) * max_extent;
    return trans + chord;
}
static float collider_origin_radius(const Collider *c) {
    if (c->border > 0.0f) return vlen(V3(c->he.x + c->border, c->he.y + c->border, c->he.z + c->border)); /* the inner local box loosened by the border */
    if (c->shape == RO_SHAPE_CUBOID || c->shape >= RO_SHAPE_CYLINDER) return vlen(c->he); /* max(|mins|,|maxs|) of the local AABB */
    if (c->shape == RO_SHAPE_CAPSULE) { v3 h = V3(c->radius, c->radius, c->radius); vset(&h, c->axis, c->he.x + c->radius); return vlen(h); }
    if (c->shape == RO_SHAPE_HALFSPACE) return INFINITY; /* |(MAX/2, MAX/2, MAX/2)| overflows: a pair with a half-space never recycles */
    return vlen(V3(c->radius, c->radius, c->radius));
}

/* manifold_reduction::reduce_manifold_naive — geometry/manifold_reduction.rs:4-84 */
static void reduce_points_naive(const TrackedContact *points, int npoints, v3 local_n1, int selected[4], int *num_selected, float prediction) {
    if (npoints <= 4) return;
    selected[0] = selected[1] = selected[2] = selected[3] = -1;
    float deepest = FLT_MAX;
    for (int i = 0; i < npoints; ++i) if (points[i].dist < deepest) { deepest = points[i].dist; selected[0] = i; }
    if (selected[0] < 0) { *num_selected = 0; return; }
    v3 a = points[selected[0]].local_p1;
    float furthest = -FLT_MAX;
    for (int i = 0; i < npoints; ++i) {
        float d = vlen2(vsub(points[i].local_p1, a));
        if (i != selected[0] && points[i].dist <= prediction && d > furthest) { furthest = d; selected[1] = i; }
    }
    if (selected[1] < 0) { *num_selected = 1; return; }
    v3 b = points[selected[1]].local_p1;
    if (a.x == b.x && a.y == b.y && a.z == b.z) { *num_selected = 1; return; }
    v3 tangent = vcross(vsub(b, a), local_n1);
    float min_dot = FLT_MAX, max_dot = -FLT_MAX;
    for (int i = 0; i < npoints; ++i) {
        if (i == selected[0] || i == selected[1] || points[i].dist > prediction) continue;
        float dot = vdot(vsub(points[i].local_p1, a), tangent);
        if (dot < min_dot) { min_dot = dot; selected[2] = i; }
        if (dot > max_dot) { max_dot = dot; selected[3] = i; }
    }
    if (selected[2] < 0) *num_selected = 2;
    else if (selected[2] == selected[3]) *num_selected = 3;
    else *num_selected = 4;
}

typedef struct { int pair; int body1, body2; int touching; int sensor; } Transition;

static int effective_dominance_group(const ro_world *w, int body) {
    /* RigidBodyDominance::effective_group — rigid_body_components.rs:1269-1275 */
    if (body >= 0 && w->bodies[body].body_type != RO_BODY_FIXED) return w->bodies[body].dominance; /* is_dynamic_or_kinematic, rigid_body_components.rs:1269 */
    return 128;
}

static int u64_cmp(const void *a, const void *b) { uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b; return x < y ? -1 : x > y; }
/* ImpulseJointSet::joints_between(b1, b2).any(|j| !j.data.contacts_enabled) — pair_update.rs:191-201 */
static void joints_disable_contacts_prepare(ro_world *w) { /* before the (parallel) pair loop */
    if (w->nc_dirty) {
        w->n_nc = 0;
        w->nc_keys = (uint64_t *)realloc(w->nc_keys, sizeof(uint64_t) * (size_t)(w->njoints + 1));
        for (int i = 0; i < w->njoints; ++i) {
            const Joint *j = &w->joints[i];
            if (j->removed || j->contacts_enabled) continue;
            uint32_t lo = (uint32_t)(j->body1 < j->body2 ? j->body1 : j->body2), hi = (uint32_t)(j->body1 < j->body2 ? j->body2 : j->body1);
            w->nc_keys[w->n_nc++] = ((uint64_t)lo << 32) | hi;
        }
        qsort(w->nc_keys, w->n_nc, sizeof(uint64_t), u64_cmp);
        w->nc_dirty = 0;
    }
}
static int joints_disable_contacts(const ro_world *w, int b1, int b2) {
    if (w->n_nc == 0) return 0;
    uint32_t lo = (uint32_t)(b1 < b2 ? b1 : b2), hi = (uint32_t)(b1 < b2 ? b2 : b1);
    uint64_t key = ((uint64_t)lo << 32) | hi;
    return bsearch(&key, w->nc_keys, w->n_nc, sizeof(uint64_t), u64_cmp) != NULL;
}
/* outcome: 0 = recycled, 1 = full update; *tr_out->pair = -1 when the pair has no transition */
/* parry intersection_test for the three shapes: ball-ball (centre distance), cuboid-cuboid (the SAT of the manifold generator: no
 * separating axis among 3 + 3 face normals and 9 edge cross products), a ball against a convex shape (solid point projection),
 * capsule-capsule (segment distance).  Cuboid-capsule goes through GJK in parry; here the distance from the capsule's segment to
 * the box is minimised over the segment parameter (a convex function: ternary search, 48 fixed iterations). */
static SmShape sm_shape_of(const Collider *c) { SmShape s; s.shape = c->shape; s.he = c->he; s.radius = c->radius; s.axis = c->axis; s.poly = c->poly; s.border = c->border; s.tri[0] = c->tri[0]; s.tri[1] = c->tri[1]; s.tri[2] = c->tri[2]; return s; }
static float point_box_dist2(v3 p, v3 he) {
    float dx = ro_maxf(fabsf(p.x) - he.x, 0.0f), dy = ro_maxf(fabsf(p.y) - he.y, 0.0f), dz = ro_maxf(fabsf(p.z) - he.z, 0.0f);
    return dx * dx + dy * dy + dz * dz;
}
static int shapes_intersect(const Collider *c1, const Collider *c2) {
    pose pos12 = pose_inv_mul(c1->pos, c2->pos);
    int s1 = c1->shape, s2 = c2->shape;
    if (s1 >= RO_SHAPE_CYLINDER || s2 >= RO_SHAPE_CYLINDER || c1->border > 0.0f || c2->border > 0.0f) { /* cylinders, cones, polyhedra, round shapes: GJK (intersection_test_support_map_support_map) */
        SmShape a = sm_shape_of(c1), b = sm_shape_of(c2);
        if (s1 == RO_SHAPE_HALFSPACE) return vdot(c1->he, pose_tp(pos12, sm_support(&b, qrot_inv(pos12.r, vneg(c1->he))))) - b.border <= 0.0f;
        if (s2 == RO_SHAPE_HALFSPACE) { pose pos21 = pose_inv(pos12); return vdot(c2->he, pose_tp(pos21, sm_support(&a, qrot_inv(pos21.r, vneg(c2->he))))) - a.border <= 0.0f; }
        return sm_intersects(&a, &b, pos12);
    }
    if (s1 > s2) { /* order the pair: ball < cuboid < capsule < half-space */
        const Collider *t = c1; c1 = c2; c2 = t; pos12 = pose_inv(pos12); s1 = c1->shape; s2 = c2->shape;
    }
    if (s2 == RO_SHAPE_HALFSPACE) {
        /* intersection_test_support_map_halfspace: the shape's support point toward -normal lies in the solid side.
         * pos12 = the half-space in shape 1's frame here: work in the half-space's frame */
        pose pos21 = pose_inv(pos12);
        v3 n = c2->he, dir = qrot_inv(pos21.r, vneg(n)), sp;
        if (s1 == RO_SHAPE_HALFSPACE) return 0; /* unsupported pair in parry: never intersecting */
        if (s1 == RO_SHAPE_BALL) sp = vmul(dir, c1->radius);
        else if (s1 == RO_SHAPE_CUBOID) sp = cuboid_support_point(c1->he, dir);
        else {
            v3 e = capsule_axis_dir(c1->axis), a = vmul(e, -c1->he.x), b = vmul(e, c1->he.x);
            sp = vadd(vdot(dir, a) > vdot(dir, b) ? a : b, vmul(dir, c1->radius));
        }
        return vdot(n, pose_tp(pos21, sp)) <= 0.0f;
    }
    if (s1 == RO_SHAPE_BALL && s2 == RO_SHAPE_BALL) { float r = c1->radius + c2->radius; return vdot(pos12.t, pos12.t) <= r * r; }
    if (s1 == RO_SHAPE_BALL && s2 == RO_SHAPE_CUBOID) { v3 c = pose_itp(pos12, V3(0, 0, 0)); return point_box_dist2(c, c2->he) <= c1->radius * c1->radius; }
    if (s1 == RO_SHAPE_BALL && s2 == RO_SHAPE_CAPSULE) {
        v3 c = pose_itp(pos12, V3(0, 0, 0)), e = capsule_axis_dir(c2->axis);
        v3 q = segment_project_point(vmul(e, -c2->he.x), vmul(e, c2->he.x), c);
        float r = c1->radius + c2->radius; v3 d = vsub(c, q);
        return vdot(d, d) <= r * r;
    }
    if (s1 == RO_SHAPE_CUBOID && s2 == RO_SHAPE_CUBOID) {
        v3 d;
        if (sat_normal_oneway(c1->he, c2->he, pos12, &d) > 0.0f) return 0;
        if (sat_normal_oneway(c2->he, c1->he, pose_inv(pos12), &d) > 0.0f) return 0;
        if (sat_edge_twoway(c1->he, c2->he, pos12, &d) > 0.0f) return 0;
        return 1;
    }
    if (s1 == RO_SHAPE_CUBOID && s2 == RO_SHAPE_CAPSULE) {
        v3 e = capsule_axis_dir(c2->axis);
        v3 a = pose_tp(pos12, vmul(e, -c2->he.x)), b = pose_tp(pos12, vmul(e, c2->he.x));
        float lo = 0.0f, hi = 1.0f;
        for (int it = 0; it < 48; ++it) {
            float m1 = lo + (hi - lo) / 3.0f, m2 = hi - (hi - lo) / 3.0f;
            float d1 = point_box_dist2(vadd(vmul(a, 1.0f - m1), vmul(b, m1)), c1->he), d2 = point_box_dist2(vadd(vmul(a, 1.0f - m2), vmul(b, m2)), c1->he);
            if (d1 <= d2) hi = m2; else lo = m1;
        }
        float t = 0.5f * (lo + hi);
        return point_box_dist2(vadd(vmul(a, 1.0f - t), vmul(b, t)), c1->he) <= c2->radius * c2->radius;
    }
    { /* capsule - capsule */
        v3 e1 = capsule_axis_dir(c1->axis), e2 = capsule_axis_dir(c2->axis);
        v3 a1 = vmul(e1, -c1->he.x), b1 = vmul(e1, c1->he.x), a2 = pose_tp(pos12, vmul(e2, -c2->he.x)), b2 = pose_tp(pos12, vmul(e2, c2->he.x));
        float s, t; closest_points_segment_segment(a1, b1, a2, b2, &s, &t);
        v3 d = vsub(vadd(vmul(a2, 1.0f - t), vmul(b2, t)), vadd(vmul(a1, 1.0f - s), vmul(b1, s)));
        float r = c1->radius + c2->radius;
        return vdot(d, d) <= r * r;
    }
}
/* pair_update.rs:404-577 for ONE solver manifold (a pair's plain manifold or a cluster): manifold reduction, the lexicographic sort in
 * the contact plane, the solver contacts that pass the distance / approach test, anchors localised and lever arms frozen.
 * pts[npts] / local_n1: the manifold (points local to wp1 / wp2 = collider pose x subshape pose); `normal` = wp1.rotation * local_n1 */
static void build_solver_contacts(const ro_world *w, int rb1, int rb2, int rel_dom, v3 normal, v3 local_n1, TrackedContact *pts, int npts,
                                  pose wp1, pose wp2, float prediction, SolverContact *scs, int *nsc_out) {
    const ro_params *prm = &w->params;
    int nsc = 0;
    if (npts > 0) {
        int selected[4] = {0, 1, 2, 3};
        int num_selected = npts < 4 ? npts : 4;
        reduce_points_naive(pts, npts, local_n1, selected, &num_selected, prediction);
        /* :430-457 lexicographic sort in the contact plane */
        if (num_selected > 1) {
            v3 basis[2]; orthonormal_basis(local_n1, basis);
            float k0[4], k1[4]; int ks[4];
            for (int i = 0; i < num_selected; ++i) {
                v3 lp = pts[selected[i]].local_p1;
                k0[i] = vdot(lp, basis[0]); k1[i] = vdot(lp, basis[1]); ks[i] = selected[i];
            }
            for (int i = 1; i < num_selected; ++i) {
                float a0 = k0[i], a1 = k1[i]; int as = ks[i]; int j = i;
                while (j > 0 && (k0[j - 1] > a0 || (k0[j - 1] == a0 && k1[j - 1] > a1))) {
                    k0[j] = k0[j - 1]; k1[j] = k1[j - 1]; ks[j] = ks[j - 1]; j--;
                }
                k0[j] = a0; k1[j] = a1; ks[j] = as;
            }
            for (int i = 0; i < num_selected; ++i) selected[i] = ks[i];
        }
        /* :459-498 solver contacts */
        for (int s = 0; s < num_selected; ++s) {
            int cid = selected[s];
            TrackedContact *c = &pts[cid];
            float eff_dist = c->dist; /* - skins (0) */
            v3 world_pt1 = pose_tp(wp1, c->local_p1);
            v3 world_pt2 = pose_tp(wp2, c->local_p2);
            int keep = eff_dist < prediction;
            if (!keep) {
                v3 vel1 = V3(0, 0, 0), vel2 = V3(0, 0, 0);
                if (rb1 >= 0) { const Body *b = &w->bodies[rb1]; vel1 = vadd(b->linvel, vcross(b->angvel, vsub(world_pt1, b->world_com))); }
                if (rb2 >= 0) { const Body *b = &w->bodies[rb2]; vel2 = vadd(b->linvel, vcross(b->angvel, vsub(world_pt2, b->world_com))); }
                keep = eff_dist + vdot(vsub(vel2, vel1), normal) * prm->dt < prediction;
            }
            if (keep) {
                SolverContact *sc = &scs[nsc++];
                sc->anchor1 = world_pt1; sc->anchor2 = world_pt2; sc->dist = eff_dist;
                sc->tangent_velocity = V3(0, 0, 0); sc->cid = cid;
            }
        }
        /* :536-577 localise anchors and freeze the solver lever arms */
        {
            int has1 = rb1 >= 0 && rel_dom <= 0, has2 = rb2 >= 0 && rel_dom >= 0;
            pose com1 = pose_ident(), com2 = pose_ident();
            if (has1) { const Body *b = &w->bodies[rb1]; com1.r = b->position.r; com1.t = pose_tp(b->position, b->local_com); }
            if (has2) { const Body *b = &w->bodies[rb2]; com2.r = b->position.r; com2.t = pose_tp(b->position, b->local_com); }
            for (int s = 0; s < nsc; ++s) {
                SolverContact *sc = &scs[s];
                float shift = vdot(vsub(sc->anchor2, sc->anchor1), normal) - sc->dist;
                v3 p1 = vadd(sc->anchor1, vmul(normal, shift));
                v3 point = vmul(vadd(p1, sc->anchor2), 0.5f);
                ContactData *pd = &pts[sc->cid].data;
                pd->solver_dp1 = has1 ? vsub(point, com1.t) : point;
                pd->solver_dp2 = has2 ? vsub(point, com2.t) : point;
                sc->anchor1 = has1 ? pose_itp(com1, p1) : p1;
                if (has2) sc->anchor2 = pose_itp(com2, sc->anchor2);
            }
        }
    }
    *nsc_out = nsc;
}
/* parry DefaultQueryDispatcher::contact_manifold_convex_convex (pair_update.rs:323-330 reaches it through contact_manifolds): co1 / co2
 * name SHAPES here (a collider, a part of a compound, a mesh triangle) — only their shape fields are read */
static void dispatch_manifold(const Collider *co1, const Collider *co2, pose pos12, float prediction, Manifold *m) {
    int s1 = co1->shape, s2 = co2->shape;
    if (s1 >= RO_SHAPE_CYLINDER || s2 >= RO_SHAPE_CYLINDER || co1->border > 0.0f || co2->border > 0.0f) { /* cylinders, cones, polyhedra, round shapes (ro_convex.h): same dispatcher order — ball arms, half-space arms, pfm_pfm */
        SmShape a = sm_shape_of(co1), b = sm_shape_of(co2);
        if (s2 == RO_SHAPE_BALL) manifold_sm_ball(pos12, &a, co2->radius, prediction, m, 0);
        else if (s1 == RO_SHAPE_BALL) manifold_sm_ball(pose_inv(pos12), &b, co1->radius, prediction, m, 1);
        else if (s1 == RO_SHAPE_HALFSPACE) manifold_halfspace_sm(pos12, co1->he, &b, prediction, m, 0);
        else if (s2 == RO_SHAPE_HALFSPACE) manifold_halfspace_sm(pose_inv(pos12), co2->he, &a, prediction, m, 1);
        else manifold_pfm_pfm(pos12, &a, &b, prediction, m);
    }
    else if (s1 == RO_SHAPE_HALFSPACE || s2 == RO_SHAPE_HALFSPACE) {
        /* (_, Ball) | (Ball, _) -> convex_ball comes before (HalfSpace, pfm) | (pfm, HalfSpace) in the dispatcher */
        if (s1 == RO_SHAPE_HALFSPACE && s2 == RO_SHAPE_HALFSPACE) m->npoints = 0; /* Unsupported */
        else if (s1 == RO_SHAPE_HALFSPACE && s2 == RO_SHAPE_BALL) manifold_halfspace_ball(pos12, co1->he, co2->radius, prediction, m, 0);
        else if (s2 == RO_SHAPE_HALFSPACE && s1 == RO_SHAPE_BALL) manifold_halfspace_ball(pose_inv(pos12), co2->he, co1->radius, prediction, m, 1);
        else if (s1 == RO_SHAPE_HALFSPACE) manifold_halfspace_pfm(pos12, co1->he, s2 == RO_SHAPE_CAPSULE, co2->he, co2->he.x, co2->radius, co2->axis, prediction, m, 0);
        else manifold_halfspace_pfm(pose_inv(pos12), co2->he, s1 == RO_SHAPE_CAPSULE, co1->he, co1->he.x, co1->radius, co1->axis, prediction, m, 1);
    }
    else if (s1 == RO_SHAPE_CUBOID && s2 == RO_SHAPE_CUBOID) manifold_cuboid_cuboid(pos12, co1->he, co2->he, prediction, m);
    else if (s1 == RO_SHAPE_BALL && s2 == RO_SHAPE_BALL) manifold_ball_ball(pos12, co1->radius, co2->radius, prediction, m);
    else if (s1 == RO_SHAPE_CAPSULE && s2 == RO_SHAPE_CAPSULE) manifold_capsule_capsule(pos12, co1->he.x, co1->radius, co1->axis, co2->he.x, co2->radius, co2->axis, prediction, m);
    else if (s1 == RO_SHAPE_CUBOID && s2 == RO_SHAPE_CAPSULE) manifold_cuboid_capsule(pos12, pos12, co1->he, co2->he.x, co2->radius, co2->axis, prediction, m, 0);
    else if (s1 == RO_SHAPE_CAPSULE && s2 == RO_SHAPE_CUBOID) manifold_cuboid_capsule(pose_inv(pos12), pos12, co2->he, co1->he.x, co1->radius, co1->axis, prediction, m, 1);
    else if (s1 == RO_SHAPE_CAPSULE && s2 == RO_SHAPE_BALL) manifold_capsule_ball(pos12, co1->he.x, co1->radius, co1->axis, co2->radius, prediction, m, 0);
    else if (s1 == RO_SHAPE_BALL && s2 == RO_SHAPE_CAPSULE) manifold_capsule_ball(pose_inv(pos12), co2->he.x, co2->radius, co2->axis, co1->radius, prediction, m, 1);
    else if (s1 == RO_SHAPE_CUBOID) manifold_cuboid_ball(pos12, co1->he, co2->radius, prediction, m, 0);
    else manifold_cuboid_ball(pose_inv(pos12), co2->he, co1->radius, prediction, m, 1);

}
#include "ro_composite.h"
static int process_composite_pair(ro_world *w, int pair_idx, Transition *tr_out, pose pos12, float prediction, float recycle_dist, int had);
static int composite_shapes_intersect(const Collider *co1, const Collider *co2, float prediction);
static int process_pair(ro_world *w, int pair_idx, Transition *tr_out) {
    tr_out->pair = -1;
    Pair *p = &w->pairs[pair_idx];
    const Collider *co1 = &w->colliders[p->c1], *co2 = &w->colliders[p->c2];
    const ro_params *prm = &w->params;
    float prediction = prm->normalized_prediction_distance * prm->length_unit;
    float recycle_dist = prm->contact_recycling ? prm->normalized_contact_recycle_distance * prm->length_unit : 0.0f;
    /* :98-106 — neither body awake (fixed or asleep): skipped */
    if (!body_is_active(w, co1->parent) && !body_is_active(w, co2->parent)) return 2;

    tr_out->sensor = 0;
    /* sensor pairs live in the intersection graph (narrow_phase/intersections.rs:17-175): no manifold, no solver, no wake-up; the
     * pair is re-tested while one of its bodies may have moved, Started / Stopped (CollisionEventFlags::SENSOR) on a change */
    if (co1->sensor || co2->sensor) {
        int had_i = p->intersecting;
        p->m.npoints = 0; p->nsc = 0; p->has_recycle = 0; pair_clear_clusters(p);
        p->intersecting = co1->parent == co2->parent && co1->parent >= 0 ? 0 : ((co_is_composite(co1) || co_is_composite(co2)) ? composite_shapes_intersect(co1, co2, prediction) : shapes_intersect(co1, co2));
        if (had_i != p->intersecting) { tr_out->pair = pair_idx; tr_out->body1 = co1->parent; tr_out->body2 = co2->parent; tr_out->touching = p->intersecting; tr_out->sensor = 1; }
        return 2;
    }
    /* :111-171 contact recycling */
    if (recycle_dist > 0.0f && p->has_recycle) {
        pose pos12 = pose_inv_mul(co1->pos, co2->pos);
        float drift = relative_pose_drift(p->rec_pos12, pos12, p->rec_max_extent);
        float rot_cos = ro_minf(relative_rot_cos(p->rec_rot1, co1->pos.r), relative_rot_cos(p->rec_rot2, co2->pos.r));
        if (drift <= p->rec_max_drift && rot_cos > 0.98f) {
            /* :141-161 a count-cleared hint (the pair slept) is recomputed: the pair re-enters the selection */
            if (pair_hint_cleared(w, p)) p->hint_seq = w->step_seq;
            return 0;
        }
    }
    int had = p->nsc > 0;
    int rb1 = co1->parent, rb2 = co2->parent;
    /* :191-201 contacts disabled between two bodies attached by a joint: clear_filtered_pair */
    if (rb1 >= 0 && rb2 >= 0 && joints_disable_contacts(w, rb1, rb2)) {
        p->m.npoints = 0; p->nsc = 0; p->has_recycle = 0; pair_clear_clusters(p); /* ContactPair::clear */
        p->hint_seq = w->step_seq;
        if (had) { tr_out->pair = pair_idx; tr_out->body1 = rb1; tr_out->body2 = rb2; tr_out->touching = 0; }
        return 1;
    }
    /* the other filters (:203-252) were applied when the pair was created (static in this scope) */
    pose pos12 = pose_inv_mul(co1->pos, co2->pos);
    float eff_prediction = prediction; /* contact_skin = 0, no soft-ccd */
    if (co_is_composite(co1) || co_is_composite(co2)) return process_composite_pair(w, pair_idx, tr_out, pos12, prediction, recycle_dist, had);

    /* :323-330 parry DefaultQueryDispatcher::contact_manifolds */
    dispatch_manifold(co1, co2, pos12, eff_prediction, &p->m);

    p->friction = ro_combine_coefficient(co1->friction, co2->friction, co1->friction_rule, co2->friction_rule);
    p->restitution = ro_combine_coefficient(co1->restitution, co2->restitution, co1->restitution_rule, co2->restitution_rule);
    p->relative_dominance = effective_dominance_group(w, rb1) - effective_dominance_group(w, rb2);
    p->normal = qrot(co1->pos.r, p->m.local_n1);
    build_solver_contacts(w, rb1, rb2, p->relative_dominance, p->normal, p->m.local_n1, p->m.points, p->m.npoints, co1->pos, co2->pos, prediction, p->sc, &p->nsc);
    /* :582-613 recycle state */
    if (recycle_dist > 0.0f) {
        float max_extent = p->has_recycle ? p->rec_max_extent : ro_maxf(collider_origin_radius(co1), collider_origin_radius(co2));
        p->rec_max_drift = p->nsc > 0 ? recycle_dist : ro_minf(recycle_dist, prediction);
        p->rec_pos12 = pos12; p->rec_rot1 = co1->pos.r; p->rec_rot2 = co2->pos.r; p->rec_max_extent = max_extent;
        p->has_recycle = 1;
    }
    p->hint_seq = w->step_seq; /* :636-650 hint refreshed from the final state */
    /* :622-629 begin/end-touch transition */
    int has = p->nsc > 0;
    if (has != had) {
        Transition *t = tr_out;
        t->pair = pair_idx; t->body1 = rb1; t->body2 = rb2; t->touching = has;
    }
    return 1;
}

/* intersection test of a sensor pair with a composite side: any candidate sub-shape pair intersects (intersection_test_composite_shape_shape) */
static int composite_shapes_intersect(const Collider *co1, const Collider *co2, float prediction) {
    (void)prediction;
    pose pos12 = pose_inv_mul(co1->pos, co2->pos);
    int cand[RO_MAX_SUBPAIRS][2], ovf;
    int n = comp_candidates(co1, co2, pos12, 0.0f, cand, RO_MAX_SUBPAIRS, &ovf);
    for (int q = 0; q < n; ++q) {
        Collider a, b; pose pa, pb; int ha, hb;
        co_sub(co1, cand[q][0] < 0 ? 0 : cand[q][0], &a, &pa, &ha); co_sub(co2, cand[q][1] < 0 ? 0 : cand[q][1], &b, &pb, &hb);
        a.pos = ha ? pose_mul(co1->pos, pa) : co1->pos; b.pos = hb ? pose_mul(co2->pos, pb) : co2->pos;
        if (shapes_intersect(&a, &b)) return 1;
    }
    return 0;
}
/* The full update of a pair with a composite collider (pair_update.rs:323-577 with several manifolds): sub-manifolds of the candidate
 * sub-shape pairs, then either the plain manifold (one candidate) or the solver clusters (contact_clustering.rs), solver contacts per
 * solver manifold, clusters with solver contacts first.  Called from process_pair behind its early outs; `had` = touched before. */
static int process_composite_pair(ro_world *w, int pair_idx, Transition *tr_out, pose pos12, float prediction, float recycle_dist, int had) {
    Pair *p = &w->pairs[pair_idx];
    const Collider *co1 = &w->colliders[p->c1], *co2 = &w->colliders[p->c2];
    const int rb1 = co1->parent, rb2 = co2->parent;
    int cand[RO_MAX_SUBPAIRS][2], ovf = 0;
    const int ncand = comp_candidates(co1, co2, pos12, prediction, cand, RO_MAX_SUBPAIRS, &ovf);
    if (ovf) {
#pragma omp atomic
        w->subpair_overflows += 1;
    }
    p->friction = ro_combine_coefficient(co1->friction, co2->friction, co1->friction_rule, co2->friction_rule);
    p->restitution = ro_combine_coefficient(co1->restitution, co2->restitution, co1->restitution_rule, co2->restitution_rule);
    p->relative_dominance = effective_dominance_group(w, rb1) - effective_dominance_group(w, rb2);
    if (!p->ex && ncand > 1) p->ex = (ExtraCluster *)calloc(RO_MAX_CLUSTERS - 1, sizeof(ExtraCluster));
    /* the solver manifolds of the previous step: the warm-start source of this step's clusters (pair_update.rs:360-369) */
    const int prev_ncl = p->ncl;
    Manifold prev_m[RO_MAX_CLUSTERS]; const Manifold *prev[RO_MAX_CLUSTERS];
    for (int k = 0; k < prev_ncl; ++k) { prev_m[k] = *sm_m(p, k); prev[k] = &prev_m[k]; }

    if (ncand <= 1) {
        /* ---- one manifold: the plain path (pair_update.rs:385-396, :398-403) ---- */
        const int s1 = ncand ? cand[0][0] : -1, s2 = ncand ? cand[0][1] : -1;
        Manifold *m = &p->m;
        if (prev_ncl > 0 || p->plain_sub[0] != s1 || p->plain_sub[1] != s2 || ncand == 0) { m->npoints = 0; m->local_n1 = V3(0, 0, 0); m->local_n2 = V3(0, 0, 0); } /* another sub-shape pair: another (new) manifold */
        p->plain_sub[0] = s1; p->plain_sub[1] = s2; p->ncl = 0;
        for (int k = 1; k < RO_MAX_CLUSTERS && p->ex; ++k) { p->ex[k - 1].nsc = 0; p->ex[k - 1].m.npoints = 0; }
        pose wp1 = co1->pos, wp2 = co2->pos;
        p->nsc = 0;
        if (ncand == 1) {
            Collider a, b; pose pa, pb; int ha, hb;
            co_sub(co1, s1 < 0 ? 0 : s1, &a, &pa, &ha); co_sub(co2, s2 < 0 ? 0 : s2, &b, &pb, &hb);
            pose rel = ha ? pose_inv_mul(pa, pos12) : pos12;
            if (hb) rel = pose_mul(rel, pb);
            dispatch_manifold(&a, &b, rel, prediction, m);
            if (prev_ncl > 0) { /* clustering stopped applying: carry the warm-start data back into the plain manifold once (:385-396) */
                v3 tn1 = m->local_n1; TrackedContact *tp = m->points; int tnp = m->npoints; const pose *tpos = ha ? &pa : NULL;
                carry_warmstart(prev, prev_ncl, &tn1, &tp, &tnp, &tpos, 1, prediction);
            }
            if (ha) wp1 = pose_mul(co1->pos, pa);
            if (hb) wp2 = pose_mul(co2->pos, pb);
            p->normal = qrot(wp1.r, m->local_n1);
            build_solver_contacts(w, rb1, rb2, p->relative_dominance, p->normal, m->local_n1, m->points, m->npoints, wp1, wp2, prediction, p->sc, &p->nsc);
        }
    } else {
        /* ---- several manifolds: solver clusters (contact_clustering.rs:33-122) ---- */
        static _Thread_local ClusterTmp cl[RO_MAX_CLUSTERS];
        int ncl = 0;
        const float dedup_eps = prediction * 0.25f, dedup_eps_sq = dedup_eps * dedup_eps;
        for (int q = 0; q < ncand; ++q) {
            Collider a, b; pose pa, pb; int ha, hb;
            co_sub(co1, cand[q][0] < 0 ? 0 : cand[q][0], &a, &pa, &ha); co_sub(co2, cand[q][1] < 0 ? 0 : cand[q][1], &b, &pb, &hb);
            pose rel = ha ? pose_inv_mul(pa, pos12) : pos12;
            if (hb) rel = pose_mul(rel, pb);
            Manifold sub; memset(&sub, 0, sizeof(sub));
            dispatch_manifold(&a, &b, rel, prediction, &sub);
            if (sub.npoints == 0) continue;
            const v3 n1 = ha ? qrot(pa.r, sub.local_n1) : sub.local_n1, n2 = hb ? qrot(pb.r, sub.local_n2) : sub.local_n2;
            cluster_add_manifold(cl, &ncl, &sub, n1, n2, ha ? &pa : NULL, hb ? &pb : NULL, dedup_eps_sq);
        }
        { /* carry_warmstart_data(prev clusters -> new clusters) :124 */
            v3 tn1[RO_MAX_CLUSTERS]; TrackedContact *tp[RO_MAX_CLUSTERS]; int tnp[RO_MAX_CLUSTERS]; const pose *tpos[RO_MAX_CLUSTERS];
            for (int c = 0; c < ncl; ++c) { tn1[c] = cl[c].n1; tp[c] = cl[c].pts; tnp[c] = cl[c].np; tpos[c] = NULL; }
            carry_warmstart(prev, prev_ncl, tn1, tp, tnp, tpos, ncl, prediction);
        }
        /* solver contacts of every cluster (pair_update.rs:404-577 with subshape_pos = None), then the clusters with solver contacts first */
        SolverContact scs[RO_MAX_CLUSTERS][4]; int nscs[RO_MAX_CLUSTERS]; v3 normals[RO_MAX_CLUSTERS];
        for (int c = 0; c < ncl; ++c) {
            normals[c] = qrot(co1->pos.r, cl[c].n1);
            build_solver_contacts(w, rb1, rb2, p->relative_dominance, normals[c], cl[c].n1, cl[c].pts, cl[c].np, co1->pos, co2->pos, prediction, scs[c], &nscs[c]);
        }
        int order[RO_MAX_CLUSTERS], no = 0;
        for (int c = 0; c < ncl; ++c) if (nscs[c] > 0) order[no++] = c;
        for (int c = 0; c < ncl; ++c) if (nscs[c] == 0) order[no++] = c;
        for (int k = 0; k < RO_MAX_CLUSTERS; ++k) {
            Manifold *m = sm_m(p, k); SolverContact *sc = sm_sc(p, k); int *nsc = sm_nsc(p, k);
            *nsc = 0; m->npoints = 0;
            if (k >= ncl) continue;
            const ClusterTmp *C = &cl[order[k]];
            m->local_n1 = C->n1; m->local_n2 = C->n2; *sm_normal(p, k) = normals[order[k]];
            /* kept: the points the solver contacts name (in that order: contact id = position), then the other points that carry
             * warm-start data (a later step may hand it on), as many as the manifold holds */
            int remap[RO_CLUSTER_PTS]; for (int i = 0; i < C->np; ++i) remap[i] = -1;
            for (int j = 0; j < nscs[order[k]]; ++j) { const int cid = scs[order[k]][j].cid; remap[cid] = m->npoints; m->points[m->npoints++] = C->pts[cid]; }
            for (int i = 0; i < C->np && m->npoints < RO_MAX_MANIFOLD_PTS; ++i) if (remap[i] < 0 && data_has_warmstart(&C->pts[i].data)) m->points[m->npoints++] = C->pts[i];
            for (int j = 0; j < nscs[order[k]]; ++j) { sc[j] = scs[order[k]][j]; sc[j].cid = remap[sc[j].cid]; }
            *nsc = nscs[order[k]];
        }
        p->ncl = ncl; p->plain_sub[0] = p->plain_sub[1] = -2;
    }
    /* :582-613 recycle state */
    if (recycle_dist > 0.0f) {
        float max_extent = p->has_recycle ? p->rec_max_extent : ro_maxf(collider_origin_radius(co1), collider_origin_radius(co2));
        p->rec_max_drift = p->nsc > 0 ? recycle_dist : ro_minf(recycle_dist, prediction);
        p->rec_pos12 = pos12; p->rec_rot1 = co1->pos.r; p->rec_rot2 = co2->pos.r; p->rec_max_extent = max_extent;
        p->has_recycle = 1;
    }
    p->hint_seq = w->step_seq;
    const int has = p->nsc > 0;
    if (has != had) { tr_out->pair = pair_idx; tr_out->body1 = rb1; tr_out->body2 = rb2; tr_out->touching = has; }
    return 1;
}

/* Canonical order of the deferred colouring: (min body, max body) like contacts.rs:369-385; the reference breaks ties
 * (several collider pairs between the same two bodies: compound bodies) by contact-graph edge id, i.e. by the creation
 * order of its BVH traversal, which no other broad phase reproduces — ties are broken here by the colliders' attachment
 * ordinals on the (min body, max body) sides, a total order that does not depend on pair creation order. */
typedef struct { uint64_t key; uint32_t tie; int pair; int b1, b2; } ColorTodo;
static int todo_cmp(const void *a, const void *b) {
    const ColorTodo *x = (const ColorTodo *)a, *y = (const ColorTodo *)b;
    if (x->key < y->key) return -1; if (x->key > y->key) return 1;
    if (x->tie < y->tie) return -1; if (x->tie > y->tie) return 1; return 0;
}
/* NarrowPhase::compute_contacts + apply_pair_transitions — contacts.rs:22-385 */
static void narrow_phase_compute_contacts(ro_world *w) {
    Transition *tr = (Transition *)malloc(sizeof(Transition) * (w->npairs + 1));
    int ntr = 0;
    w->stats.num_full_updates = 0; w->stats.num_recycled = 0;
    /* pairs are independent (contacts.rs:22-251 hands them to a rayon broadcast); transitions are
     * collected per pair and compacted in edge order afterwards */
    int nfull = 0, nrec = 0;
    joints_disable_contacts_prepare(w);
    RO_PARALLEL_FOR_RED(nfull, nrec)
    for (int i = 0; i < w->npairs; ++i) { int o = process_pair(w, i, &tr[i]); nfull += o == 1; nrec += o == 0; }
    w->stats.num_full_updates = nfull; w->stats.num_recycled = nrec;
    for (int i = 0; i < w->npairs; ++i) if (tr[i].pair >= 0) tr[ntr++] = tr[i];
    /* transitions are visited in edge order (already ascending); end-touch frees its colour first */
    ColorTodo *todo = (ColorTodo *)malloc(sizeof(ColorTodo) * (ntr + 1));
    int ntodo = 0;
    for (int i = 0; i < ntr; ++i) {
        Pair *p = &w->pairs[tr[i].pair];
        /* contacts.rs:316-323: Started / Stopped for pairs with ActiveEvents::COLLISION_EVENTS */
        if ((w->colliders[p->c1].active_events | w->colliders[p->c2].active_events) & 1u) push_collision_event(w, p->c1, p->c2, tr[i].touching, tr[i].sensor ? 1 : 0);
        if (tr[i].sensor) continue; /* intersections neither colour nor wake anything */
        if (!tr[i].touching) { /* end touch: the colour is freed, the contact link unlinked (contacts.rs:328-330, :359) */
            clear_pair_solver_color(w, p);
            pi_journal(w, tr[i].body1, tr[i].body2, 2, ((uint64_t)(uint32_t)p->c1 << 32) | (uint32_t)p->c2);
            continue;
        }
        /* wake rule (contacts.rs:333-351): starts wake the sleeping side strongly (whole island), stops never wake */
        if (body_is_sleeping_nonfixed(w, tr[i].body1)) wake_request(w, tr[i].body1, 1);
        if (body_is_sleeping_nonfixed(w, tr[i].body2)) wake_request(w, tr[i].body2, 1);
        uint32_t a = tr[i].body1 >= 0 ? (uint32_t)tr[i].body1 : RO_NO_BODY;
        uint32_t b = tr[i].body2 >= 0 ? (uint32_t)tr[i].body2 : RO_NO_BODY;
        uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
        todo[ntodo].key = ((uint64_t)lo << 32) | hi; todo[ntodo].pair = tr[i].pair;
        { uint32_t o1 = (uint32_t)w->colliders[p->c1].ord, o2 = (uint32_t)w->colliders[p->c2].ord; todo[ntodo].tie = a < b ? (o1 << 20) | o2 : (o2 << 20) | o1; } /* (lower body id: < 4,096 colliders; the higher one may be "no body": up to 2^20 parentless colliders) */
        todo[ntodo].b1 = tr[i].body1; todo[ntodo].b2 = tr[i].body2; ntodo++;
    }
    /* apply_deferred_solver_coloring — contacts.rs:369-385 */
    qsort(todo, ntodo, sizeof(ColorTodo), todo_cmp);
    for (int i = 0; i < ntodo; ++i) assign_pair_solver_color(w, &w->pairs[todo[i].pair], todo[i].b1, todo[i].b2);
    free(todo); free(tr);
}

/* ------------------------------------------------------------------------------------ */
/* Solver */
static void solver_reserve(ro_world *w, int nb, int nc) {
    static int cap_b = 0; (void)cap_b;
    w->vels = (SolverVel *)realloc(w->vels, sizeof(SolverVel) * (nb + 1));
    w->incr = (SolverVel *)realloc(w->incr, sizeof(SolverVel) * (nb + 1));
    w->poses = (SolverPose *)realloc(w->poses, sizeof(SolverPose) * (nb + 1));
    w->gyro = (Gyro *)realloc(w->gyro, sizeof(Gyro) * (nb + 1));
    w->flags = (uint8_t *)realloc(w->flags, nb + 1);
    w->dyn_bodies = (int *)realloc(w->dyn_bodies, sizeof(int) * (nb + 1));
    if (nc > w->cap_cons) { w->cap_cons = nc * 2 + 64; w->cons = (Constraint *)realloc(w->cons, sizeof(Constraint) * w->cap_cons); }
}

static void gather_vel(const ro_world *w, uint32_t id, SolverVel *v) {
    if (id == RO_NO_BODY) { v->linear = V3(0, 0, 0); v->angular = V3(0, 0, 0); } else *v = w->vels[id];
}
static void gather_pose(const ro_world *w, uint32_t id, SolverPose *p) {
    if (id == RO_NO_BODY) { memset(p, 0, sizeof(*p)); p->rotation = qident(); } else *p = w->poses[id];
}
static void scatter_vel(ro_world *w, uint32_t id, const SolverVel *v) { if (id != RO_NO_BODY) w->vels[id] = *v; }
static v3 spose_tp(const SolverPose *p, v3 x) { return vadd(qrot(p->rotation, x), p->translation); }
static v3 spose_itp(const SolverPose *p, v3 x) { return qrot_inv(p->rotation, vsub(x, p->translation)); }

/* ContactWithTwistFrictionBuilder::generate — contact_with_twist_friction.rs:58-424 (one lane) */
static void constraint_generate(ro_world *w, int pair_idx, Constraint *c) {
    const int sm_k = RO_SM_K(pair_idx); /* solver-manifold reference: pair | k << 28 */
    Pair *p = &w->pairs[RO_SM_PAIR(pair_idx)];
    const v3 sm_normal_v = *sm_normal(p, sm_k); const int sm_nsc_v = *sm_nsc(p, sm_k); const SolverContact *sm_sc_v = sm_sc(p, sm_k); const Manifold *sm_m_v = sm_m(p, sm_k);
    memset(c, 0, sizeof(*c));
    uint32_t ids1 = p->relative_dominance <= 0 ? p->solver_body_ids[0] : RO_NO_BODY;
    uint32_t ids2 = p->relative_dominance >= 0 ? p->solver_body_ids[1] : RO_NO_BODY;
    SolverVel vels1, vels2; SolverPose poses1, poses2;
    gather_vel(w, ids1, &vels1); gather_vel(w, ids2, &vels2);
    gather_pose(w, ids1, &poses1); gather_pose(w, ids2, &poses2);
    v3 world_com1 = poses1.translation, world_com2 = poses2.translation;
    v3 force_dir1 = vneg(sm_normal_v);
    int count = sm_nsc_v < 4 ? sm_nsc_v : 4;
    v3 tangents1[2];
    tangents1[0] = orthonormal_vector(force_dir1);           /* contact_constraint/mod.rs:27-46 */
    tangents1[1] = vcross(force_dir1, tangents1[0]);
    float inv_num_points = 1.0f / (float)count;

    c->dir1 = force_dir1; c->im1 = poses1.im; c->im2 = poses2.im; c->ii1 = poses1.ii; c->ii2 = poses2.ii;
    c->local_n1 = qrot_inv(poses1.rotation, force_dir1);
    c->restitution = p->restitution;
    c->solver_vel1 = ids1; c->solver_vel2 = ids2; c->pair = pair_idx; c->num_contacts = count;
    c->tangent1 = tangents1[0];
    c->limit = p->friction;

    v3 friction_center = V3(0, 0, 0), friction_center2 = V3(0, 0, 0), tangent_vel = V3(0, 0, 0);
    float twist_warmstart = 0.0f, tangent_warmstart[2] = {0, 0};
    v3 points[4];
    for (int k = 0; k < count; ++k) {
        float weight = inv_num_points;
        const SolverContact *sc = &sm_sc_v[k];
        const ContactData *pd = &sm_m_v->points[sc->cid].data;
        float warmstart_impulse = pd->warmstart_impulse;
        v3 wt = pd->warmstart_tangent_world;
        float wti[2] = {vdot(wt, tangents1[0]), vdot(wt, tangents1[1])};
        float warmstart_twist_impulse = pd->warmstart_twist_impulse;
        int is_new = pd->impulse == 0.0f;
        float is_bouncy = is_new ? (p->restitution > 0.0f ? 1.0f : 0.0f) : (p->restitution >= 1.0f ? 1.0f : 0.0f);

        v3 p1 = spose_tp(&poses1, sc->anchor1);
        v3 p2 = spose_tp(&poses2, sc->anchor2);
        float dist = vdot(vsub(p1, p2), force_dir1);
        v3 dp1 = pd->solver_dp1, dp2 = pd->solver_dp2;
        v3 point = vadd(world_com1, dp1);
        points[k] = point;
        friction_center = vadd(friction_center, vmul(point, weight));
        friction_center2 = vadd(friction_center2, vmul(vadd(world_com2, dp2), weight));
        v3 vel1 = vadd(vels1.linear, vcross(vels1.angular, dp1));
        v3 vel2 = vadd(vels2.linear, vcross(vels2.angular, dp2));
        twist_warmstart += warmstart_twist_impulse * weight;
        tangent_warmstart[0] += wti[0] * weight; tangent_warmstart[1] += wti[1] * weight;
        tangent_vel = vadd(tangent_vel, vmul(sc->tangent_velocity, weight));
        c->contact_id[k] = sc->cid;
        {
            v3 torque_dir1 = vcross(dp1, force_dir1);
            v3 torque_dir2 = vcross(dp2, vneg(force_dir1));
            v3 ii_torque_dir1 = sym3_mul(poses1.ii, torque_dir1);
            v3 ii_torque_dir2 = sym3_mul(poses2.ii, torque_dir2);
            v3 imsum = vadd(poses1.im, poses2.im);
            float projected_mass = ro_inv(vdot(force_dir1, vcmul(imsum, force_dir1)) + vdot(ii_torque_dir1, torque_dir1) +
                                          vdot(ii_torque_dir2, torque_dir2));
            float projected_velocity = vdot(vsub(vel1, vel2), force_dir1);
            float restitution_seed = is_bouncy * p->restitution * projected_velocity;
            NormalPart *n = &c->normal_part[k];
            n->torque_dir1 = torque_dir1; n->torque_dir2 = torque_dir2;
            n->ii_torque_dir1 = ii_torque_dir1; n->ii_torque_dir2 = ii_torque_dir2;
            n->impulse = warmstart_impulse; n->impulse_accumulator = -n->impulse; n->r = projected_mass;
            c->infos[k].local_p1 = spose_itp(&poses1, point);
            c->infos[k].local_p2 = spose_itp(&poses2, vadd(world_com2, dp2));
            c->infos[k].dist = dist - vdot(vsub(point, vadd(world_com2, dp2)), force_dir1);
            c->infos[k].restitution_seed = restitution_seed;
        }
    }
    c->tangent_part.impulse[0] = tangent_warmstart[0]; c->tangent_part.impulse[1] = tangent_warmstart[1];
    c->tangent_part.impulse_accumulator[0] = -tangent_warmstart[0]; c->tangent_part.impulse_accumulator[1] = -tangent_warmstart[1];
    c->twist_part.impulse = count > 1 ? twist_warmstart : 0.0f;
    c->twist_part.impulse_accumulator = -c->twist_part.impulse;
    c->local_friction_center1 = spose_itp(&poses1, friction_center);
    c->local_friction_center2 = spose_itp(&poses2, friction_center2);
    c->tangent_vel = tangent_vel;
    v3 dp1 = vsub(friction_center, world_com1), dp2 = vsub(friction_center2, world_com2);
    if (count > 1) {
        for (int k = 0; k < count; ++k) c->twist_dists[k] = vlen(vsub(friction_center, points[k]));
        v3 ii_twist_dir1 = sym3_mul(poses1.ii, force_dir1);
        v3 ii_twist_dir2 = sym3_mul(poses2.ii, vneg(force_dir1));
        c->twist_part.rhs = 0.0f;
        c->twist_part.r = ro_inv(vdot(ii_twist_dir1, force_dir1) + vdot(ii_twist_dir2, vneg(force_dir1)));
    }
    c->tangent_part.dp1 = dp1; c->tangent_part.dp2 = dp2;
    for (int j = 0; j < 2; ++j) {
        v3 torque_dir1 = vcross(dp1, tangents1[j]);
        v3 torque_dir2 = vcross(dp2, vneg(tangents1[j]));
        v3 ii_torque_dir1 = sym3_mul(poses1.ii, torque_dir1);
        v3 ii_torque_dir2 = sym3_mul(poses2.ii, torque_dir2);
        v3 imsum = vadd(poses1.im, poses2.im);
        float r = vdot(tangents1[j], vcmul(imsum, tangents1[j])) + vdot(ii_torque_dir1, torque_dir1) + vdot(ii_torque_dir2, torque_dir2);
        float rhs_wo_bias = vdot(tangent_vel, tangents1[j]);
        c->tangent_part.torque_dir1[j] = torque_dir1; c->tangent_part.torque_dir2[j] = torque_dir2;
        c->tangent_part.ii_torque_dir1[j] = ii_torque_dir1; c->tangent_part.ii_torque_dir2[j] = ii_torque_dir2;
        c->tangent_part.rhs_wo_bias[j] = rhs_wo_bias; c->tangent_part.rhs[j] = rhs_wo_bias; c->tangent_part.r[j] = r;
    }
    c->tangent_part.r[2] = 2.0f * (vdot(c->tangent_part.ii_torque_dir1[0], c->tangent_part.torque_dir1[1]) +
                                   vdot(c->tangent_part.ii_torque_dir2[0], c->tangent_part.torque_dir2[1]));
}

typedef struct { quat rotation; v3 translation; } Xform;
static void gather_xform(const ro_world *w, uint32_t id, Xform *x) {
    if (id == RO_NO_BODY) { x->rotation = qident(); x->translation = V3(0, 0, 0); }
    else { x->rotation = w->poses[id].rotation; x->translation = w->poses[id].translation; }
}
static v3 xform_tp(const Xform *x, v3 p) { return vadd(qrot(x->rotation, p), x->translation); }

/* ContactWithTwistFrictionBuilder::update — contact_with_twist_friction.rs:426-522; `dt` = substep dt */
static void constraint_update(const ro_world *w, Constraint *c, float dt, float solved_dt) {
    const ro_params *prm = &w->params;
    int is_static = c->solver_vel1 == RO_NO_BODY || c->solver_vel2 == RO_NO_BODY;
    float dyn_cfm = spring_cfm_factor(prm->contact_natural_frequency, prm->contact_damping_ratio, dt);
    float static_cfm = spring_cfm_factor(prm->static_contact_natural_frequency, prm->static_contact_damping_ratio, dt);
    float dyn_erp = spring_erp_inv_dt(prm->contact_natural_frequency, prm->contact_damping_ratio, dt);
    float static_erp = spring_erp_inv_dt(prm->static_contact_natural_frequency, prm->static_contact_damping_ratio, dt);
    float fstatic = is_static ? 1.0f : 0.0f;
    float cfm_factor = dyn_cfm + fstatic * (static_cfm - dyn_cfm);
    float inv_dt = dt == 0.0f ? 0.0f : 1.0f / dt;
    float erp_inv_dt = dyn_erp + fstatic * (static_erp - dyn_erp);
    float max_corrective_velocity = prm->normalized_max_corrective_velocity * prm->length_unit;
    float warmstart_coeff = prm->warmstart_coefficient;
    Xform poses1, poses2; gather_xform(w, c->solver_vel1, &poses1); gather_xform(w, c->solver_vel2, &poses2);
    v3 tangents1[2] = {c->tangent1, vcross(c->dir1, c->tangent1)};
    v3 tangent_delta = vmul(c->tangent_vel, solved_dt);
    for (int k = 0; k < c->num_contacts; ++k) {
        NormalPart *n = &c->normal_part[k];
        v3 p1 = vadd(xform_tp(&poses1, c->infos[k].local_p1), tangent_delta);
        v3 p2 = xform_tp(&poses2, c->infos[k].local_p2);
        float dist = c->infos[k].dist + vdot(vsub(p1, p2), c->dir1);
        float rhs_wo_bias = ro_maxf(dist, 0.0f) * inv_dt;
        float rhs_bias = ro_clampf(dist * erp_inv_dt, -max_corrective_velocity, 0.0f);
        n->rhs_wo_bias = rhs_wo_bias; n->rhs = rhs_wo_bias + rhs_bias;
        n->cfm_factor = dist <= 0.0f ? cfm_factor : 1.0f;
        n->impulse_accumulator += n->impulse;
        n->impulse *= warmstart_coeff;
    }
    {
        v3 p1 = vadd(xform_tp(&poses1, c->local_friction_center1), tangent_delta);
        v3 p2 = xform_tp(&poses2, c->local_friction_center2);
        for (int j = 0; j < 2; ++j) {
            float bias = vdot(vsub(p1, p2), tangents1[j]) * inv_dt;
            c->tangent_part.rhs[j] = c->tangent_part.rhs_wo_bias[j] + bias;
        }
        for (int j = 0; j < 2; ++j) { c->tangent_part.impulse_accumulator[j] += c->tangent_part.impulse[j]; c->tangent_part.impulse[j] *= warmstart_coeff; }
        c->twist_part.impulse_accumulator += c->twist_part.impulse;
        c->twist_part.impulse *= warmstart_coeff;
    }
    c->cfm_factor = cfm_factor;
}

/* refresh_rhs_wo_bias — contact_with_twist_friction.rs:529-554 */
static void constraint_refresh_rhs_wo_bias(const ro_world *w, Constraint *c, float dt, float solved_dt) {
    float inv_dt = dt == 0.0f ? 0.0f : 1.0f / dt;
    Xform poses1, poses2; gather_xform(w, c->solver_vel1, &poses1); gather_xform(w, c->solver_vel2, &poses2);
    v3 tangent_delta = vmul(c->tangent_vel, solved_dt);
    for (int k = 0; k < c->num_contacts; ++k) {
        v3 p1 = vadd(xform_tp(&poses1, c->infos[k].local_p1), tangent_delta);
        v3 p2 = xform_tp(&poses2, c->infos[k].local_p2);
        float dist = c->infos[k].dist + vdot(vsub(p1, p2), c->dir1);
        c->normal_part[k].rhs = ro_maxf(dist, 0.0f) * inv_dt;
        c->normal_part[k].cfm_factor = 1.0f;
    }
    c->cfm_factor = 1.0f;
    c->tangent_part.rhs[0] = c->tangent_part.rhs_wo_bias[0];
    c->tangent_part.rhs[1] = c->tangent_part.rhs_wo_bias[1];
}

/* warmstart — contact_with_twist_friction.rs:633-678; elements contact_constraint_element.rs:465-478,627-647,720-732 */
static void constraint_warmstart(ro_world *w, Constraint *c) {
    SolverVel v1, v2; gather_vel(w, c->solver_vel1, &v1); gather_vel(w, c->solver_vel2, &v2);
    for (int k = 0; k < c->num_contacts; ++k) {
        NormalPart *n = &c->normal_part[k];
        v1.linear = vadd(v1.linear, vmul(vcmul(c->dir1, c->im1), n->impulse));
        v1.angular = vadd(v1.angular, vmul(n->ii_torque_dir1, n->impulse));
        v2.linear = vadd(v2.linear, vmul(vcmul(c->dir1, c->im2), -n->impulse));
        v2.angular = vadd(v2.angular, vmul(n->ii_torque_dir2, n->impulse));
    }
    v3 t0 = c->tangent1, t1 = vcross(c->dir1, c->tangent1);
    float i0 = c->tangent_part.impulse[0], i1 = c->tangent_part.impulse[1];
    v1.linear = vadd(v1.linear, vcmul(vadd(vmul(t0, i0), vmul(t1, i1)), c->im1));
    v1.angular = vadd(v1.angular, vadd(vmul(c->tangent_part.ii_torque_dir1[0], i0), vmul(c->tangent_part.ii_torque_dir1[1], i1)));
    v2.linear = vadd(v2.linear, vcmul(vadd(vmul(t0, -i0), vmul(t1, -i1)), c->im2));
    v2.angular = vadd(v2.angular, vadd(vmul(c->tangent_part.ii_torque_dir2[0], i0), vmul(c->tangent_part.ii_torque_dir2[1], i1)));
    if (c->num_contacts > 1) {
        v3 a = sym3_mul(c->ii1, c->dir1), b = sym3_mul(c->ii2, c->dir1);
        v1.angular = vadd(v1.angular, vmul(a, c->twist_part.impulse));
        v2.angular = vsub(v2.angular, vmul(b, c->twist_part.impulse));
    }
    scatter_vel(w, c->solver_vel1, &v1); scatter_vel(w, c->solver_vel2, &v2);
}

/* solve — contact_with_twist_friction.rs:680-781; elements contact_constraint_element.rs:481-504,650-705,735-755 */
static void constraint_solve(ro_world *w, Constraint *c, int solve_friction) {
    SolverVel v1, v2; gather_vel(w, c->solver_vel1, &v1); gather_vel(w, c->solver_vel2, &v2);
    for (int k = 0; k < c->num_contacts; ++k) {
        NormalPart *n = &c->normal_part[k];
        float dvel = vdot(c->dir1, v1.linear) + vdot(n->torque_dir1, v1.angular) - vdot(c->dir1, v2.linear) +
                     vdot(n->torque_dir2, v2.angular) + n->rhs;
        float new_impulse = n->cfm_factor * ro_maxf(n->impulse - n->r * dvel, 0.0f);
        float dlambda = new_impulse - n->impulse;
        n->impulse = new_impulse;
        v1.linear = vadd(v1.linear, vmul(vcmul(c->dir1, c->im1), dlambda));
        v1.angular = vadd(v1.angular, vmul(n->ii_torque_dir1, dlambda));
        v2.linear = vadd(v2.linear, vmul(vcmul(c->dir1, c->im2), -dlambda));
        v2.angular = vadd(v2.angular, vmul(n->ii_torque_dir2, dlambda));
    }
    if (solve_friction) {
        v3 t0 = c->tangent1, t1 = vcross(c->dir1, c->tangent1);
        float tangent_limit = 0.0f, twist_limit = 0.0f;
        for (int k = 0; k < c->num_contacts; ++k) {
            tangent_limit += c->normal_part[k].impulse;
            twist_limit += c->normal_part[k].impulse * c->twist_dists[k];
        }
        tangent_limit *= c->limit; twist_limit *= c->limit;
        if (c->num_contacts > 1) {
            v3 a = sym3_mul(c->ii1, c->dir1), b = sym3_mul(c->ii2, c->dir1);
            float dvel = vdot(c->dir1, vsub(v1.angular, v2.angular)) + c->twist_part.rhs;
            float new_impulse = ro_clampf(c->twist_part.impulse - c->twist_part.r * dvel, -twist_limit, twist_limit);
            float dlambda = new_impulse - c->twist_part.impulse;
            c->twist_part.impulse = new_impulse;
            v1.angular = vadd(v1.angular, vmul(a, dlambda));
            v2.angular = vsub(v2.angular, vmul(b, dlambda));
        }
        {
            float dvel_0 = vdot(t0, v1.linear) + vdot(c->tangent_part.torque_dir1[0], v1.angular) - vdot(t0, v2.linear) +
                           vdot(c->tangent_part.torque_dir2[0], v2.angular) + c->tangent_part.rhs[0];
            float dvel_1 = vdot(t1, v1.linear) + vdot(c->tangent_part.torque_dir1[1], v1.angular) - vdot(t1, v2.linear) +
                           vdot(c->tangent_part.torque_dir2[1], v2.angular) + c->tangent_part.rhs[1];
            float k11 = c->tangent_part.r[0], k22 = c->tangent_part.r[1], k12 = c->tangent_part.r[2] * 0.5f;
            float inv_det = ro_inv(k11 * k22 - k12 * k12);
            float d0 = (k22 * dvel_0 - k12 * dvel_1) * inv_det;
            float d1 = (k11 * dvel_1 - k12 * dvel_0) * inv_det;
            float n0 = c->tangent_part.impulse[0] - d0, n1 = c->tangent_part.impulse[1] - d1;
            /* nalgebra simd_cap_magnitude(limit) */
            float len = sqrtf(n0 * n0 + n1 * n1);
            if (len > tangent_limit) { float s = tangent_limit / len; n0 *= s; n1 *= s; }
            float dl0 = n0 - c->tangent_part.impulse[0], dl1 = n1 - c->tangent_part.impulse[1];
            c->tangent_part.impulse[0] = n0; c->tangent_part.impulse[1] = n1;
            v1.linear = vadd(v1.linear, vcmul(vadd(vmul(t0, dl0), vmul(t1, dl1)), c->im1));
            v1.angular = vadd(v1.angular, vadd(vmul(c->tangent_part.ii_torque_dir1[0], dl0), vmul(c->tangent_part.ii_torque_dir1[1], dl1)));
            v2.linear = vadd(v2.linear, vcmul(vadd(vmul(t0, -dl0), vmul(t1, -dl1)), c->im2));
            v2.angular = vadd(v2.angular, vadd(vmul(c->tangent_part.ii_torque_dir2[0], dl0), vmul(c->tangent_part.ii_torque_dir2[1], dl1)));
        }
    }
    scatter_vel(w, c->solver_vel1, &v1); scatter_vel(w, c->solver_vel2, &v2);
}

/* apply_restitution — contact_with_twist_friction.rs:568-597; solve_restitution contact_constraint_element.rs:508-534 */
static void constraint_apply_restitution(ro_world *w, Constraint *c) {
    int any = 0;
    for (int k = 0; k < c->num_contacts; ++k) any |= c->infos[k].restitution_seed < 0.0f;
    if (!any) return;
    SolverVel v1, v2; gather_vel(w, c->solver_vel1, &v1); gather_vel(w, c->solver_vel2, &v2);
    for (int k = 0; k < c->num_contacts; ++k) {
        NormalPart *n = &c->normal_part[k];
        float seed = c->infos[k].restitution_seed;
        float dvel = vdot(c->dir1, v1.linear) + vdot(n->torque_dir1, v1.angular) - vdot(c->dir1, v2.linear) +
                     vdot(n->torque_dir2, v2.angular) + seed;
        int gate = seed < 0.0f && (n->impulse_accumulator + n->impulse) > 0.0f;
        float new_impulse = gate ? ro_maxf(n->impulse - n->r * dvel, 0.0f) : n->impulse;
        float dlambda = new_impulse - n->impulse;
        n->impulse = new_impulse;
        v1.linear = vadd(v1.linear, vmul(vcmul(c->dir1, c->im1), dlambda));
        v1.angular = vadd(v1.angular, vmul(n->ii_torque_dir1, dlambda));
        v2.linear = vadd(v2.linear, vmul(vcmul(c->dir1, c->im2), -dlambda));
        v2.angular = vadd(v2.angular, vmul(n->ii_torque_dir2, dlambda));
    }
    scatter_vel(w, c->solver_vel1, &v1); scatter_vel(w, c->solver_vel2, &v2);
}

/* writeback_impulses — contact_with_twist_friction.rs:783-829 */
static void constraint_writeback(ro_world *w, const Constraint *c) {
    Pair *p = &w->pairs[RO_SM_PAIR(c->pair)];
    Manifold *sm_m_v = sm_m(p, RO_SM_K(c->pair));
    v3 tangent2 = vcross(c->dir1, c->tangent1);
    /* the stored impulses go through utils::canonicalize_zero (x + 0.0: -0.0 becomes +0.0; utils/mod.rs:80-102, enabled by the
     * reference's enhanced-determinism feature, a bit-level no-op for every other value) */
    v3 wtw = vcanon(vadd(vmul(c->tangent1, canon0(c->tangent_part.impulse[0])), vmul(tangent2, canon0(c->tangent_part.impulse[1]))));
    for (int k = 0; k < c->num_contacts; ++k) {
        ContactData *pd = &sm_m_v->points[c->contact_id[k]].data;
        pd->warmstart_impulse = canon0(c->normal_part[k].impulse);
        pd->impulse = canon0(c->normal_part[k].impulse_accumulator + c->normal_part[k].impulse);
        pd->warmstart_tangent_world = wtw;
        pd->warmstart_twist_impulse = canon0(c->twist_part.impulse);
    }
}


/* ---- FrictionModel::Coulomb: ContactWithCoulombFriction(+Builder) — contact_with_coulomb_friction.rs:52-760 --------
 * One Coulomb friction constraint per contact point (exact coupled 2x2 solve, limit mu * lambda_k) instead of the
 * friction-centre tangent + twist pair.  The normal parts are the twist model's. */
static void coulomb_generate(ro_world *w, int pair_idx, Constraint *c) {
    const int sm_k = RO_SM_K(pair_idx); /* solver-manifold reference: pair | k << 28 */
    Pair *p = &w->pairs[RO_SM_PAIR(pair_idx)];
    const v3 sm_normal_v = *sm_normal(p, sm_k); const int sm_nsc_v = *sm_nsc(p, sm_k); const SolverContact *sm_sc_v = sm_sc(p, sm_k); const Manifold *sm_m_v = sm_m(p, sm_k);
    memset(c, 0, sizeof(*c));
    uint32_t ids1 = p->relative_dominance <= 0 ? p->solver_body_ids[0] : RO_NO_BODY;
    uint32_t ids2 = p->relative_dominance >= 0 ? p->solver_body_ids[1] : RO_NO_BODY;
    SolverVel vels1, vels2; SolverPose poses1, poses2;
    gather_vel(w, ids1, &vels1); gather_vel(w, ids2, &vels2);
    gather_pose(w, ids1, &poses1); gather_pose(w, ids2, &poses2);
    v3 world_com1 = poses1.translation, world_com2 = poses2.translation;
    v3 force_dir1 = vneg(sm_normal_v);
    int count = sm_nsc_v < 4 ? sm_nsc_v : 4;
    v3 tangents1[2];
    tangents1[0] = orthonormal_vector(force_dir1);           /* compute_tangent_contact_directions, mod.rs:27-46 */
    tangents1[1] = vcross(force_dir1, tangents1[0]);
    c->dir1 = force_dir1; c->im1 = poses1.im; c->im2 = poses2.im; c->ii1 = poses1.ii; c->ii2 = poses2.ii;
    c->local_n1 = qrot_inv(poses1.rotation, force_dir1);
    c->restitution = p->restitution;
    c->solver_vel1 = ids1; c->solver_vel2 = ids2; c->pair = pair_idx; c->num_contacts = count;
    c->tangent1 = tangents1[0];
    c->limit = p->friction;
    v3 imsum = vadd(poses1.im, poses2.im);
    for (int k = 0; k < count; ++k) {
        const SolverContact *sc = &sm_sc_v[k];
        const ContactData *pd = &sm_m_v->points[sc->cid].data;
        float warmstart_impulse = pd->warmstart_impulse;
        v3 wt = pd->warmstart_tangent_world;
        float wti[2] = {vdot(wt, tangents1[0]), vdot(wt, tangents1[1])};
        int is_new = pd->impulse == 0.0f;
        float is_bouncy = is_new ? (p->restitution > 0.0f ? 1.0f : 0.0f) : (p->restitution >= 1.0f ? 1.0f : 0.0f);
        v3 p1 = spose_tp(&poses1, sc->anchor1);
        v3 p2 = spose_tp(&poses2, sc->anchor2);
        float dist = vdot(vsub(p1, p2), force_dir1);
        v3 dp1 = pd->solver_dp1, dp2 = pd->solver_dp2;
        v3 vel1 = vadd(vels1.linear, vcross(vels1.angular, dp1));
        v3 vel2 = vadd(vels2.linear, vcross(vels2.angular, dp2));
        c->contact_id[k] = sc->cid;
        {
            v3 torque_dir1 = vcross(dp1, force_dir1);
            v3 torque_dir2 = vcross(dp2, vneg(force_dir1));
            v3 ii_torque_dir1 = sym3_mul(poses1.ii, torque_dir1);
            v3 ii_torque_dir2 = sym3_mul(poses2.ii, torque_dir2);
            float projected_mass = ro_inv(vdot(force_dir1, vcmul(imsum, force_dir1)) + vdot(ii_torque_dir1, torque_dir1) +
                                          vdot(ii_torque_dir2, torque_dir2));
            float projected_velocity = vdot(vsub(vel1, vel2), force_dir1);
            NormalPart *n = &c->normal_part[k];
            n->torque_dir1 = torque_dir1; n->torque_dir2 = torque_dir2;
            n->ii_torque_dir1 = ii_torque_dir1; n->ii_torque_dir2 = ii_torque_dir2;
            n->impulse = warmstart_impulse; n->impulse_accumulator = -warmstart_impulse; n->r = projected_mass;
            c->infos[k].restitution_seed = is_bouncy * p->restitution * projected_velocity;
        }
        c->ctangent[k].impulse[0] = wti[0]; c->ctangent[k].impulse[1] = wti[1];
        c->ctangent[k].impulse_accumulator[0] = -wti[0]; c->ctangent[k].impulse_accumulator[1] = -wti[1];
        for (int j = 0; j < 2; ++j) {
            v3 torque_dir1 = vcross(dp1, tangents1[j]);
            v3 torque_dir2 = vcross(dp2, vneg(tangents1[j]));
            v3 ii_torque_dir1 = sym3_mul(poses1.ii, torque_dir1);
            v3 ii_torque_dir2 = sym3_mul(poses2.ii, torque_dir2);
            float r = vdot(tangents1[j], vcmul(imsum, tangents1[j])) + vdot(ii_torque_dir1, torque_dir1) + vdot(ii_torque_dir2, torque_dir2);
            float rhs_wo_bias = vdot(sc->tangent_velocity, tangents1[j]);
            c->ctangent[k].torque_dir1[j] = torque_dir1; c->ctangent[k].torque_dir2[j] = torque_dir2;
            c->ctangent[k].ii_torque_dir1[j] = ii_torque_dir1; c->ctangent[k].ii_torque_dir2[j] = ii_torque_dir2;
            c->ctangent[k].rhs_wo_bias[j] = rhs_wo_bias; c->ctangent[k].rhs[j] = rhs_wo_bias; c->ctangent[k].r[j] = r;
        }
        c->ctangent[k].r[2] = 2.0f * (vdot(c->ctangent[k].ii_torque_dir1[0], c->ctangent[k].torque_dir1[1]) +
                                      vdot(c->ctangent[k].ii_torque_dir2[0], c->ctangent[k].torque_dir2[1]));
        c->infos[k].local_p1 = spose_itp(&poses1, vadd(world_com1, dp1));
        c->infos[k].local_p2 = spose_itp(&poses2, vadd(world_com2, dp2));
        c->infos[k].dist = dist - vdot(vsub(vadd(world_com1, dp1), vadd(world_com2, dp2)), force_dir1);
    }
}
/* update :362-455 (tangent_velocity is zero without contact-modification hooks, so the p1 shift vanishes) */
static void coulomb_update(const ro_world *w, Constraint *c, float dt, float solved_dt) {
    const ro_params *prm = &w->params;
    (void)solved_dt;
    int is_static = c->solver_vel1 == RO_NO_BODY || c->solver_vel2 == RO_NO_BODY;
    float dyn_cfm = spring_cfm_factor(prm->contact_natural_frequency, prm->contact_damping_ratio, dt);
    float static_cfm = spring_cfm_factor(prm->static_contact_natural_frequency, prm->static_contact_damping_ratio, dt);
    float dyn_erp = spring_erp_inv_dt(prm->contact_natural_frequency, prm->contact_damping_ratio, dt);
    float static_erp = spring_erp_inv_dt(prm->static_contact_natural_frequency, prm->static_contact_damping_ratio, dt);
    float fstatic = is_static ? 1.0f : 0.0f;
    float cfm_factor = dyn_cfm + fstatic * (static_cfm - dyn_cfm);
    float inv_dt = dt == 0.0f ? 0.0f : 1.0f / dt;
    float erp_inv_dt = dyn_erp + fstatic * (static_erp - dyn_erp);
    float max_corrective_velocity = prm->normalized_max_corrective_velocity * prm->length_unit;
    float warmstart_coeff = prm->warmstart_coefficient;
    Xform poses1, poses2; gather_xform(w, c->solver_vel1, &poses1); gather_xform(w, c->solver_vel2, &poses2);
    v3 tangents1[2] = {c->tangent1, vcross(c->dir1, c->tangent1)};
    for (int k = 0; k < c->num_contacts; ++k) {
        NormalPart *n = &c->normal_part[k];
        v3 p1 = xform_tp(&poses1, c->infos[k].local_p1);
        v3 p2 = xform_tp(&poses2, c->infos[k].local_p2);
        float dist = c->infos[k].dist + vdot(vsub(p1, p2), c->dir1);
        float rhs_wo_bias = ro_maxf(dist, 0.0f) * inv_dt;
        float rhs_bias = ro_clampf(dist * erp_inv_dt, -max_corrective_velocity, 0.0f);
        n->rhs_wo_bias = rhs_wo_bias; n->rhs = rhs_wo_bias + rhs_bias;
        n->cfm_factor = dist <= 0.0f ? cfm_factor : 1.0f;
        n->impulse_accumulator += n->impulse;
        n->impulse *= warmstart_coeff;
        for (int j = 0; j < 2; ++j) { c->ctangent[k].impulse_accumulator[j] += c->ctangent[k].impulse[j]; c->ctangent[k].impulse[j] *= warmstart_coeff; }
        for (int j = 0; j < 2; ++j) {
            float bias = vdot(vsub(p1, p2), tangents1[j]) * inv_dt;
            c->ctangent[k].rhs[j] = c->ctangent[k].rhs_wo_bias[j] + bias;
        }
    }
    c->cfm_factor = cfm_factor;
}
/* refresh_rhs_wo_bias :460-489 */
static void coulomb_refresh_rhs_wo_bias(const ro_world *w, Constraint *c, float dt, float solved_dt) {
    (void)solved_dt;
    float inv_dt = dt == 0.0f ? 0.0f : 1.0f / dt;
    Xform poses1, poses2; gather_xform(w, c->solver_vel1, &poses1); gather_xform(w, c->solver_vel2, &poses2);
    for (int k = 0; k < c->num_contacts; ++k) {
        v3 p1 = xform_tp(&poses1, c->infos[k].local_p1);
        v3 p2 = xform_tp(&poses2, c->infos[k].local_p2);
        float dist = c->infos[k].dist + vdot(vsub(p1, p2), c->dir1);
        c->normal_part[k].rhs = ro_maxf(dist, 0.0f) * inv_dt;
        c->normal_part[k].cfm_factor = 1.0f;
        c->ctangent[k].rhs[0] = c->ctangent[k].rhs_wo_bias[0]; c->ctangent[k].rhs[1] = c->ctangent[k].rhs_wo_bias[1];
    }
    c->cfm_factor = 1.0f;
}
/* warmstart :561-603; elements contact_constraint_element.rs:64-97,226-240 */
static void coulomb_warmstart(ro_world *w, Constraint *c) {
    SolverVel v1, v2; gather_vel(w, c->solver_vel1, &v1); gather_vel(w, c->solver_vel2, &v2);
    for (int k = 0; k < c->num_contacts; ++k) {
        NormalPart *n = &c->normal_part[k];
        v1.linear = vadd(v1.linear, vmul(vcmul(c->dir1, c->im1), n->impulse));
        v1.angular = vadd(v1.angular, vmul(n->ii_torque_dir1, n->impulse));
        v2.linear = vadd(v2.linear, vmul(vcmul(c->dir1, c->im2), -n->impulse));
        v2.angular = vadd(v2.angular, vmul(n->ii_torque_dir2, n->impulse));
    }
    v3 t0 = c->tangent1, t1 = vcross(c->dir1, c->tangent1);
    for (int k = 0; k < c->num_contacts; ++k) {
        float i0 = c->ctangent[k].impulse[0], i1 = c->ctangent[k].impulse[1];
        v1.linear = vadd(v1.linear, vcmul(vadd(vmul(t0, i0), vmul(t1, i1)), c->im1));
        v1.angular = vadd(v1.angular, vadd(vmul(c->ctangent[k].ii_torque_dir1[0], i0), vmul(c->ctangent[k].ii_torque_dir1[1], i1)));
        v2.linear = vadd(v2.linear, vcmul(vadd(vmul(t0, -i0), vmul(t1, -i1)), c->im2));
        v2.angular = vadd(v2.angular, vadd(vmul(c->ctangent[k].ii_torque_dir2[0], i0), vmul(c->ctangent[k].ii_torque_dir2[1], i1)));
    }
    scatter_vel(w, c->solver_vel1, &v1); scatter_vel(w, c->solver_vel2, &v2);
}
/* solve :605-690; elements contact_constraint_element.rs:100-176,242-270 */
static void coulomb_solve(ro_world *w, Constraint *c, int solve_friction) {
    SolverVel v1, v2; gather_vel(w, c->solver_vel1, &v1); gather_vel(w, c->solver_vel2, &v2);
    for (int k = 0; k < c->num_contacts; ++k) {
        NormalPart *n = &c->normal_part[k];
        float dvel = vdot(c->dir1, v1.linear) + vdot(n->torque_dir1, v1.angular) - vdot(c->dir1, v2.linear) +
                     vdot(n->torque_dir2, v2.angular) + n->rhs;
        float new_impulse = n->cfm_factor * ro_maxf(n->impulse - n->r * dvel, 0.0f);
        float dlambda = new_impulse - n->impulse;
        n->impulse = new_impulse;
        v1.linear = vadd(v1.linear, vmul(vcmul(c->dir1, c->im1), dlambda));
        v1.angular = vadd(v1.angular, vmul(n->ii_torque_dir1, dlambda));
        v2.linear = vadd(v2.linear, vmul(vcmul(c->dir1, c->im2), -dlambda));
        v2.angular = vadd(v2.angular, vmul(n->ii_torque_dir2, dlambda));
    }
    if (solve_friction) {
        v3 t0 = c->tangent1, t1 = vcross(c->dir1, c->tangent1);
        for (int k = 0; k < c->num_contacts; ++k) {
            float limit = c->limit * c->normal_part[k].impulse;
            float dvel_0 = vdot(t0, v1.linear) + vdot(c->ctangent[k].torque_dir1[0], v1.angular) - vdot(t0, v2.linear) +
                           vdot(c->ctangent[k].torque_dir2[0], v2.angular) + c->ctangent[k].rhs[0];
            float dvel_1 = vdot(t1, v1.linear) + vdot(c->ctangent[k].torque_dir1[1], v1.angular) - vdot(t1, v2.linear) +
                           vdot(c->ctangent[k].torque_dir2[1], v2.angular) + c->ctangent[k].rhs[1];
            float k11 = c->ctangent[k].r[0], k22 = c->ctangent[k].r[1], k12 = c->ctangent[k].r[2] * 0.5f;
            float inv_det = ro_inv(k11 * k22 - k12 * k12);
            float d0 = (k22 * dvel_0 - k12 * dvel_1) * inv_det;
            float d1 = (k11 * dvel_1 - k12 * dvel_0) * inv_det;
            float n0 = c->ctangent[k].impulse[0] - d0, n1 = c->ctangent[k].impulse[1] - d1;
            float len = sqrtf(n0 * n0 + n1 * n1);            /* nalgebra simd_cap_magnitude(limit) */
            if (len > limit) { float s = limit / len; n0 *= s; n1 *= s; }
            float dl0 = n0 - c->ctangent[k].impulse[0], dl1 = n1 - c->ctangent[k].impulse[1];
            c->ctangent[k].impulse[0] = n0; c->ctangent[k].impulse[1] = n1;
            v1.linear = vadd(v1.linear, vcmul(vadd(vmul(t0, dl0), vmul(t1, dl1)), c->im1));
            v1.angular = vadd(v1.angular, vadd(vmul(c->ctangent[k].ii_torque_dir1[0], dl0), vmul(c->ctangent[k].ii_torque_dir1[1], dl1)));
            v2.linear = vadd(v2.linear, vcmul(vadd(vmul(t0, -dl0), vmul(t1, -dl1)), c->im2));
            v2.angular = vadd(v2.angular, vadd(vmul(c->ctangent[k].ii_torque_dir2[0], dl0), vmul(c->ctangent[k].ii_torque_dir2[1], dl1)));
        }
    }
    scatter_vel(w, c->solver_vel1, &v1); scatter_vel(w, c->solver_vel2, &v2);
}
/* writeback_impulses :692-760: per-point world-space friction impulse; the twist warm start is left untouched */
static void coulomb_writeback(ro_world *w, const Constraint *c) {
    Pair *p = &w->pairs[RO_SM_PAIR(c->pair)];
    Manifold *sm_m_v = sm_m(p, RO_SM_K(c->pair));
    v3 tangent2 = vcross(c->dir1, c->tangent1);
    for (int k = 0; k < c->num_contacts; ++k) {
        ContactData *pd = &sm_m_v->points[c->contact_id[k]].data;
        pd->warmstart_impulse = canon0(c->normal_part[k].impulse);
        pd->impulse = canon0(c->normal_part[k].impulse_accumulator + c->normal_part[k].impulse);
        pd->warmstart_tangent_world = vcanon(vadd(vmul(c->tangent1, canon0(c->ctangent[k].impulse[0])), vmul(tangent2, canon0(c->ctangent[k].impulse[1]))));
    }
}

/* gyroscopic_corrected_angvel — dynamics/rigid_body.rs:2023-2046 */
static v3 gyroscopic_corrected_angvel(v3 angvel, quat principal_axes, v3 pi, v3 inv_pi, float dt) {
    v3 wl = qrot_inv(principal_axes, angvel);
    v3 curr = vcmul(pi, wl);
    v3 explicit_gyro = vmul(vneg(vcross(wl, curr)), dt);
    v3 total = vadd(curr, explicit_gyro);
    float sq = vlen2(total);
    if (sq != 0.0f) {
        v3 capped = vmul(total, sqrtf(vlen2(curr) / sq));
        return qrot(principal_axes, vcmul(inv_pi, capped));
    }
    return angvel;
}

/* StagedIslandSolver::init_and_solve + run_worker — staged_island_solver/init.rs:30-545, worker.rs:32-898 */
/* ------------------------------------------------------------------------------------ */
/* Impulse joints */

/* ImpulseJointSet::select_active_interactions — impulse_joint_set.rs:504-572:
 * enabled joints with at least one dynamic body, in edge (insertion) order; stamps solver-body ids. */
static void joints_select_active(ro_world *w) {
    w->nactive_joints = 0;
    for (int i = 0; i < w->njoints; ++i) {
        Joint *j = &w->joints[i];
        if (j->removed) continue;
        const Body *rb1 = &w->bodies[j->body1], *rb2 = &w->bodies[j->body2];
        int d1 = rb1->body_type != RO_BODY_FIXED, d2 = rb2->body_type != RO_BODY_FIXED; /* is_dynamic_or_kinematic */
        if (!d1 && !d2) continue;
        if ((d1 && rb1->sleeping) || (d2 && rb2->sleeping)) continue; /* :553-556 */
        j->solver_body_ids[0] = d1 ? rb1->solver_id : RO_NO_BODY;
        j->solver_body_ids[1] = d2 ? rb2->solver_id : RO_NO_BODY;
        w->active_joints[w->nactive_joints++] = i;
    }
}
static u128 u128_or(u128 a, u128 b) { u128 r = {a.lo | b.lo, a.hi | b.hi}; return r; }
/* ParallelInteractionGroups::group_interactions — interaction_groups.rs:59-197: greedy colouring of
 * the joints in the contacts' colour space (external masks = persistent contact colours per body). */
static void joints_color(ro_world *w) {
    memset(w->joint_body_colors, 0, sizeof(u128) * (size_t)(w->ndyn + 1));
    for (int a = 0; a < w->nactive_joints; ++a) {
        Joint *j = &w->joints[w->active_joints[a]];
        uint32_t id1 = j->solver_body_ids[0], id2 = j->solver_body_ids[1];
        u128 ext = {0, 0};
        if (j->body1 < w->cap_masks) ext = u128_or(ext, w->color_masks[j->body1]);
        if (j->body2 < w->cap_masks) ext = u128_or(ext, w->color_masks[j->body2]);
        int color;
        if (id1 != RO_NO_BODY && id2 != RO_NO_BODY) {
            u128 m = u128_or(u128_or(w->joint_body_colors[id1], w->joint_body_colors[id2]), ext);
            if (j->solver_color < 128 && !u128_test(m, j->solver_color)) color = j->solver_color;
            else { color = 128; for (int c = 0; c < RO_DYNAMIC_COLOR_COUNT; ++c) if (!u128_test(m, c)) { color = c; break; } }
            if (color < 128) { u128_set(&w->joint_body_colors[id1], color); u128_set(&w->joint_body_colors[id2], color); }
        } else {
            uint32_t id = id1 != RO_NO_BODY ? id1 : id2;
            u128 m = u128_or(w->joint_body_colors[id], ext);
            if (j->solver_color < 128 && !u128_test(m, j->solver_color)) color = j->solver_color;
            else { color = 128; for (int c = 127; c >= 0; --c) if (!u128_test(m, c)) { color = c; break; } }
            if (color < 128) u128_set(&w->joint_body_colors[id], color);
        }
        j->solver_color = (uint8_t)color;
    }
    /* single_group_joint_layout — staged_island_solver/joints.rs:331-395: colours with at least
     * JOINT_BATCH * LAYOUT_REF_WORKERS / 2 = 64 joints are parallel stages (ascending colour), every
     * other colour (and the overflow colour 128) is solved serially afterwards, colour-major. */
    int counts[RO_NUM_COLORS]; memset(counts, 0, sizeof(counts));
    for (int a = 0; a < w->nactive_joints; ++a) counts[w->joints[w->active_joints[a]].solver_color]++;
    int n = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int c = 0; c < RO_NUM_COLORS; ++c) {
            int parallel = c < 128 && counts[c] >= 64;
            if (counts[c] == 0 || (pass == 0) != parallel) continue;
            for (int a = 0; a < w->nactive_joints; ++a)
                if (w->joints[w->active_joints[a]].solver_color == c) w->joint_order[n++] = w->active_joints[a];
            if (pass == 0) w->njoint_parallel = n;
        }
    if (n == 0) w->njoint_parallel = 0;
}
/* JointConstraintBuilder::generate — joint_constraint_builder.rs:34-60 +
 * GenericJoint::transform_to_solver_body_space — generic_joint.rs:624-636 */
static int ro_ctz(uint32_t x) { int n = 0; while (!(x & 1u)) { x >>= 1; ++n; } return n; } /* trailing_zeros of a non-zero mask */
static int joint_num_rows(const Joint *j) {
    /* joint_velocity_constraint.rs:159-352: per-axis rows skip the coupled axes; the coupled linear axes add one motor row and one
     * limit row (carried by the first coupled axis), two coupled angular axes one limit row; a coupled angular motor is a no-op */
    const uint32_t locked = j->locked_axes & 0x3fu, coupled = j->coupled_axes, motor = j->motor_axes & ~locked, limit = j->limit_axes & ~locked;
    int n = 0;
    for (int i = 0; i < 6; ++i) { if (locked & (1u << i)) n++; if ((limit & ~coupled) & (1u << i)) n++; if ((motor & ~coupled) & (1u << i)) n++; }
    if ((motor & coupled) & 7u) n++;
    if ((coupled & 0x38u) && (limit & (1u << ro_ctz(coupled & 0x38u)))) n++;
    if ((coupled & 7u) && (limit & (1u << ro_ctz(coupled & 7u)))) n++;
    return n;
}
static void joint_builder_generate(ro_world *w, Joint *j, int *num_rows) {
    const Body *rb1 = &w->bodies[j->body1], *rb2 = &w->bodies[j->body2];
    j->sb_frame1 = j->local_frame1; j->sb_frame2 = j->local_frame2;
    if (rb1->body_type == RO_BODY_FIXED) j->sb_frame1 = pose_mul(rb1->position, j->local_frame1);
    else j->sb_frame1.t = vsub(j->sb_frame1.t, rb1->local_com);
    if (rb2->body_type == RO_BODY_FIXED) j->sb_frame2 = pose_mul(rb2->position, j->local_frame2);
    else j->sb_frame2.t = vsub(j->sb_frame2.t, rb2->local_com);
    j->first_row = *num_rows;
    *num_rows += joint_num_rows(j);
}
/* JointConstraintHelper::finalize_constraints (joint_constraint_helper.rs:676-720): modified Gram-Schmidt over one block of
 * rows; rows with bounded impulses (limits, motors) are not removed from the others */
static void joint_finalize_rows(JointRow *out, int len) {
    if (len == 0) return;
    v3 imsum = vadd(out[0].im1, out[0].im2);
    for (int a = 0; a < len; ++a) {
        JointRow *cj = &out[a];
        float dot_jj = vdot(cj->lin_jac, vcmul(imsum, cj->lin_jac)) + vdot(cj->ii_ang_jac1, cj->ang_jac1) + vdot(cj->ii_ang_jac2, cj->ang_jac2);
        float cfm_gain = dot_jj * cj->cfm_coeff + cj->cfm_gain;
        float inv_dot_jj = ro_inv(dot_jj);
        cj->inv_lhs = ro_inv(dot_jj + cfm_gain);
        cj->cfm_gain = cfm_gain;
        if (cj->impulse_bounds[0] != -FLT_MAX || cj->impulse_bounds[1] != FLT_MAX) continue;
        for (int b = a + 1; b < len; ++b) {
            JointRow *ci = &out[b];
            float dot_ij = vdot(ci->lin_jac, vcmul(imsum, cj->lin_jac)) + vdot(ci->ii_ang_jac1, cj->ang_jac1) + vdot(ci->ii_ang_jac2, cj->ang_jac2);
            float coeff = dot_ij * inv_dot_jj;
            ci->lin_jac = vsub(ci->lin_jac, vmul(cj->lin_jac, coeff));
            ci->ang_jac1 = vsub(ci->ang_jac1, vmul(cj->ang_jac1, coeff));
            ci->ang_jac2 = vsub(ci->ang_jac2, vmul(cj->ang_jac2, coeff));
            ci->ii_ang_jac1 = vsub(ci->ii_ang_jac1, vmul(cj->ii_ang_jac1, coeff));
            ci->ii_ang_jac2 = vsub(ci->ii_ang_jac2, vmul(cj->ii_ang_jac2, coeff));
            ci->rhs_wo_bias = ci->rhs_wo_bias - cj->rhs_wo_bias * coeff;
            ci->rhs = ci->rhs - cj->rhs * coeff;
        }
    }
}
/* JointMotor::motor_params (generic_joint.rs:236-250) + MotorModel::combine_coefficients (motor_model.rs:39-58) */
typedef struct { float erp_inv_dt, cfm_coeff, cfm_gain, target_pos, target_vel, max_impulse; } MotorParams;
static MotorParams motor_params(const ro_joint_motor *m, float dt) {
    MotorParams p;
    p.erp_inv_dt = m->stiffness * ro_inv(dt * m->stiffness + m->damping);
    float c = ro_inv(dt * dt * m->stiffness + dt * m->damping);
    p.cfm_coeff = m->model == 0 ? c : 0.0f;
    p.cfm_gain = m->model == 0 ? 0.0f : c;
    p.target_pos = m->target_pos; p.target_vel = m->target_vel; p.max_impulse = m->max_force * dt;
    return p;
}
/* utils::smallest_abs_diff_between_angles (utils/mod.rs:217-224); simd_signum = copysign(1, x) */
static float smallest_abs_diff_between_angles(float a, float b) {
    float s_err = a - b;
    float sgn = copysignf(1.0f, s_err);
    float s_err_complement = s_err - sgn * 6.28318530717958647692f;
    return fabsf(s_err) < fabsf(s_err_complement) ? s_err : s_err_complement;
}
/* JointConstraint::<Real,1>::update (joint_velocity_constraint.rs:144-353) for locked linear axes:
 * JointConstraintHelper::new (joint_constraint_helper.rs:95-164), lock_linear (:411-458),
 * finalize_constraints (:676-720). */
static int joint_update_rows(const ro_world *w, const Joint *j, float dt, JointRow *out) {
    SolverPose rb1, rb2;
    gather_pose(w, j->solver_body_ids[0], &rb1); gather_pose(w, j->solver_body_ids[1], &rb2);
    pose p1, p2; p1.r = rb1.rotation; p1.t = rb1.translation; p2.r = rb2.rotation; p2.t = rb2.translation;
    pose frame1 = pose_mul(p1, j->sb_frame1), frame2 = pose_mul(p2, j->sb_frame2);
    v3 world_com1 = rb1.translation, world_com2 = rb2.translation;
    float erp_inv_dt = spring_erp_inv_dt(w->params.joint_natural_frequency, w->params.joint_damping_ratio, dt);
    float cfm_coeff = spring_cfm_coeff(w->params.joint_natural_frequency, w->params.joint_damping_ratio, dt);
    float m[3][3]; quat_to_mat(frame1.r, m); /* basis: column i = (m[0][i], m[1][i], m[2][i]) */
    v3 col[3]; for (int i = 0; i < 3; ++i) col[i] = V3(m[0][i], m[1][i], m[2][i]);
    v3 lin_err = vsub(frame2.t, frame1.t);
    v3 new_center1 = frame2.t;
    for (int i = 0; i < 3; ++i) if (j->locked_axes & (1u << i)) new_center1 = vsub(new_center1, vmul(col[i], vdot(lin_err, col[i])));
    frame1.t = new_center1;
    v3 r1 = vsub(frame1.t, world_com1), r2 = vsub(frame2.t, world_com2);
    /* cmat * basis: column i = gcross_matrix(r) * col[i] with glam Mat3 * Vec3 = x_axis*v.x + y_axis*v.y + z_axis*v.z */
    v3 c1x = V3(0.0f, r1.z, -r1.y), c1y = V3(-r1.z, 0.0f, r1.x), c1z = V3(r1.y, -r1.x, 0.0f);
    v3 c2x = V3(0.0f, r2.z, -r2.y), c2y = V3(-r2.z, 0.0f, r2.x), c2z = V3(r2.y, -r2.x, 0.0f);
    int len = 0, start = 0;
    /* motor rows first, finalised as a block of their own (joint_velocity_constraint.rs:186-246): motor_angular
     * (joint_constraint_helper.rs:566-625) for the angular axes, then motor_linear (:285-331) */
    const uint32_t coupled = j->coupled_axes;
    const uint32_t motor_all = j->motor_axes & ~j->locked_axes, motor_axes = motor_all & ~coupled; /* (motor_axes & !coupled_axes, :190-221) */
    if (motor_all) {
        quat q1m = frame1.r, q2m = frame2.r;
        float sgnm = copysignf(1.0f, qdot(q1m, q2m));
        quat ang_errm = qmul(qconj(q1m), q2m);
        float imagm[3] = {ang_errm.x * sgnm, ang_errm.y * sgnm, ang_errm.z * sgnm};
        for (int a = 0; a < 3; ++a) {
            if (!(motor_axes & (8u << a))) continue;
            MotorParams mp = motor_params(&j->motors[3 + a], dt);
            v3 ang_jac = col[a];
            float rhs_wo_bias = 0.0f;
            if (mp.erp_inv_dt != 0.0f) {
                float ang_dist = ro_asin_portable(ro_clampf(imagm[a], -1.0f, 1.0f)) * 2.0f;
                rhs_wo_bias += smallest_abs_diff_between_angles(ang_dist, mp.target_pos) * mp.erp_inv_dt;
            }
            rhs_wo_bias += -mp.target_vel;
            JointRow *c = &out[len++];
            c->solver_vel1 = j->solver_body_ids[0]; c->solver_vel2 = j->solver_body_ids[1];
            c->im1 = rb1.im; c->im2 = rb2.im;
            c->impulse = 0.0f; c->impulse_bounds[0] = -mp.max_impulse; c->impulse_bounds[1] = mp.max_impulse;
            c->lin_jac = V3(0, 0, 0); c->ang_jac1 = ang_jac; c->ang_jac2 = ang_jac;
            c->ii_ang_jac1 = sym3_mul(rb1.ii, ang_jac);
            c->ii_ang_jac2 = sym3_mul(rb2.ii, ang_jac);
            c->inv_lhs = 0.0f; c->cfm_coeff = mp.cfm_coeff; c->cfm_gain = mp.cfm_gain;
            c->rhs = rhs_wo_bias; c->rhs_wo_bias = rhs_wo_bias; c->dof = 12 + 3 + a;
        }
        for (int i = 0; i < 3; ++i) {
            if (!(motor_axes & (1u << i))) continue;
            MotorParams mp = motor_params(&j->motors[i], dt);
            JointRow *c = &out[len++];
            c->solver_vel1 = j->solver_body_ids[0]; c->solver_vel2 = j->solver_body_ids[1];
            c->im1 = rb1.im; c->im2 = rb2.im;
            c->impulse = 0.0f; c->impulse_bounds[0] = -mp.max_impulse; c->impulse_bounds[1] = mp.max_impulse;
            c->lin_jac = col[i];
            c->ang_jac1 = vadd(vadd(vmul(c1x, col[i].x), vmul(c1y, col[i].y)), vmul(c1z, col[i].z));
            c->ang_jac2 = vadd(vadd(vmul(c2x, col[i].x), vmul(c2y, col[i].y)), vmul(c2z, col[i].z));
            c->ii_ang_jac1 = sym3_mul(rb1.ii, c->ang_jac1);
            c->ii_ang_jac2 = sym3_mul(rb2.ii, c->ang_jac2);
            float rhs_wo_bias = 0.0f;
            float dist = vdot(lin_err, c->lin_jac);
            if (mp.erp_inv_dt != 0.0f) rhs_wo_bias += (dist - mp.target_pos) * mp.erp_inv_dt;
            float target_vel = mp.target_vel;
            if ((j->limit_axes & ~j->locked_axes) & (1u << i)) { float inv_dt = ro_inv(dt); target_vel = ro_clampf(target_vel, (j->limits[i][0] - dist) * inv_dt, (j->limits[i][1] - dist) * inv_dt); }
            rhs_wo_bias += -target_vel;
            c->inv_lhs = 0.0f; c->cfm_coeff = mp.cfm_coeff; c->cfm_gain = mp.cfm_gain;
            c->rhs = rhs_wo_bias; c->rhs_wo_bias = rhs_wo_bias; c->dof = 12 + i;
        }
        /* (a coupled ANGULAR motor is a no-op: "TODO: coupled angular motor constraint", :223-225) */
        if ((motor_all & coupled) & 7u) {
            /* motor_linear_coupled (joint_constraint_helper.rs:333-408): ONE row along the combined error of the coupled linear axes;
             * motor and limits are those of the first coupled linear axis (SpringJoint: LinX) */
            const int fa = ro_ctz(coupled & 7u);
            MotorParams mp = motor_params(&j->motors[fa], dt);
            v3 lin_jac = V3(0, 0, 0), aj1 = V3(0, 0, 0), aj2 = V3(0, 0, 0);
            for (int i = 0; i < 3; ++i) {
                if (!(coupled & (1u << i))) continue;
                float coeff = vdot(col[i], lin_err);
                lin_jac = vadd(lin_jac, vmul(col[i], coeff));
                aj1 = vadd(aj1, vmul(vadd(vadd(vmul(c1x, col[i].x), vmul(c1y, col[i].y)), vmul(c1z, col[i].z)), coeff));
                aj2 = vadd(aj2, vmul(vadd(vadd(vmul(c2x, col[i].x), vmul(c2y, col[i].y)), vmul(c2z, col[i].z)), coeff));
            }
            float dist = sqrtf(vdot(lin_jac, lin_jac)), inv_dist = ro_inv(dist);
            lin_jac = vmul(lin_jac, inv_dist); aj1 = vmul(aj1, inv_dist); aj2 = vmul(aj2, inv_dist);
            float rhs_wo_bias = 0.0f;
            if (mp.erp_inv_dt != 0.0f) rhs_wo_bias += (dist - mp.target_pos) * mp.erp_inv_dt;
            float target_vel = mp.target_vel;
            if ((j->limit_axes & ~j->locked_axes) & (1u << fa)) { float inv_dt = ro_inv(dt); target_vel = ro_clampf(target_vel, (j->limits[fa][0] - dist) * inv_dt, (j->limits[fa][1] - dist) * inv_dt); }
            rhs_wo_bias += -target_vel;
            JointRow *c = &out[len++];
            c->solver_vel1 = j->solver_body_ids[0]; c->solver_vel2 = j->solver_body_ids[1];
            c->im1 = rb1.im; c->im2 = rb2.im;
            c->impulse = 0.0f; c->impulse_bounds[0] = -mp.max_impulse; c->impulse_bounds[1] = mp.max_impulse;
            c->lin_jac = lin_jac; c->ang_jac1 = aj1; c->ang_jac2 = aj2;
            c->ii_ang_jac1 = sym3_mul(rb1.ii, aj1); c->ii_ang_jac2 = sym3_mul(rb2.ii, aj2);
            c->inv_lhs = 0.0f; c->cfm_coeff = mp.cfm_coeff; c->cfm_gain = mp.cfm_gain;
            c->rhs = rhs_wo_bias; c->rhs_wo_bias = rhs_wo_bias; c->dof = 12 + fa;
        }
        joint_finalize_rows(out, len);
        start = len;
    }
    if (j->locked_axes & 0x38u) {
        /* locked angular axes — JointConstraintHelper::new (:129-139): ang_basis = diff_conj1_2_tr(q1, q2) * sgn,
         * ang_err = (q1^-1 q2) * sgn, sgn = copysign(1, q1 . q2); lock_angular (:628-673).  The scalar update emits the
         * angular lock rows BEFORE the linear ones (joint_velocity_constraint.rs:253-283). */
        quat q1 = frame1.r, q2 = frame2.r;
        v3 v1 = V3(q1.x, q1.y, q1.z), v2 = V3(q2.x, q2.y, q2.z);
        float w1 = q1.w, w2 = q2.w;
        /* RotationOps::diff_conj1_2 (utils/rotation_ops.rs:121-135), matrices as columns */
        v3 u = vadd(vmul(v1, w2), vmul(v2, w1));
        v3 cu[3] = {V3(0.0f, u.z, -u.y), V3(-u.z, 0.0f, u.x), V3(u.y, -u.x, 0.0f)};
        v3 ca[3] = {V3(0.0f, v1.z, -v1.y), V3(-v1.z, 0.0f, v1.x), V3(v1.y, -v1.x, 0.0f)};
        v3 cb[3] = {V3(0.0f, v2.z, -v2.y), V3(-v2.z, 0.0f, v2.x), V3(v2.y, -v2.x, 0.0f)};
        float d = w1 * w2;
        v3 dg[3] = {V3(d, 0.0f, 0.0f), V3(0.0f, d, 0.0f), V3(0.0f, 0.0f, d)};
        float v2c[3] = {v2.x, v2.y, v2.z};
        v3 M[3];
        for (int c = 0; c < 3; ++c) {
            v3 kron = vmul(v1, v2c[c]);
            v3 prod = vadd(vadd(vmul(ca[0], cb[c].x), vmul(ca[1], cb[c].y)), vmul(ca[2], cb[c].z));
            M[c] = vmul(vadd(vsub(vadd(kron, dg[c]), cu[c]), prod), 0.5f);
        }
        float sgn = copysignf(1.0f, qdot(q1, q2));
        quat ang_err = qmul(qconj(q1), q2);
        float ang_err_imag[3] = {ang_err.x * sgn, ang_err.y * sgn, ang_err.z * sgn};
        for (int a = 0; a < 3; ++a) {
            if (!(j->locked_axes & (8u << a))) continue;
            /* column a of the transpose = row a of M */
            v3 ang_jac = a == 0 ? V3(M[0].x, M[1].x, M[2].x) : a == 1 ? V3(M[0].y, M[1].y, M[2].y) : V3(M[0].z, M[1].z, M[2].z);
            ang_jac = vmul(ang_jac, sgn);
            JointRow *c = &out[len++];
            c->solver_vel1 = j->solver_body_ids[0]; c->solver_vel2 = j->solver_body_ids[1];
            c->im1 = rb1.im; c->im2 = rb2.im;
            c->impulse = 0.0f; c->impulse_bounds[0] = -FLT_MAX; c->impulse_bounds[1] = FLT_MAX;
            c->lin_jac = V3(0, 0, 0); c->ang_jac1 = ang_jac; c->ang_jac2 = ang_jac;
            float rhs_wo_bias = 0.0f;
            float rhs_bias = ang_err_imag[a] * erp_inv_dt;
            c->ii_ang_jac1 = sym3_mul(rb1.ii, ang_jac);
            c->ii_ang_jac2 = sym3_mul(rb2.ii, ang_jac);
            c->inv_lhs = 0.0f; c->cfm_coeff = cfm_coeff; c->cfm_gain = 0.0f;
            c->rhs = rhs_wo_bias + rhs_bias; c->rhs_wo_bias = rhs_wo_bias; c->dof = 3 + a;
        }
    }
    for (int i = 0; i < 3; ++i) {
        if (!(j->locked_axes & (1u << i))) continue;
        JointRow *c = &out[len++];
        c->solver_vel1 = j->solver_body_ids[0]; c->solver_vel2 = j->solver_body_ids[1];
        c->im1 = rb1.im; c->im2 = rb2.im;
        c->impulse = 0.0f; c->impulse_bounds[0] = -FLT_MAX; c->impulse_bounds[1] = FLT_MAX;
        c->lin_jac = col[i];
        c->ang_jac1 = vadd(vadd(vmul(c1x, col[i].x), vmul(c1y, col[i].y)), vmul(c1z, col[i].z));
        c->ang_jac2 = vadd(vadd(vmul(c2x, col[i].x), vmul(c2y, col[i].y)), vmul(c2z, col[i].z));
        float rhs_wo_bias = 0.0f;
        float rhs_bias = vdot(c->lin_jac, lin_err) * erp_inv_dt;
        c->ii_ang_jac1 = sym3_mul(rb1.ii, c->ang_jac1);
        c->ii_ang_jac2 = sym3_mul(rb2.ii, c->ang_jac2);
        c->inv_lhs = 0.0f; c->cfm_coeff = cfm_coeff; c->cfm_gain = 0.0f;
        c->rhs = rhs_wo_bias + rhs_bias; c->rhs_wo_bias = rhs_wo_bias; c->dof = i;
    }
    /* limited (free) axes — scalar update order: limit_angular rows, then limit_linear rows (joint_velocity_constraint.rs:285-314) */
    const uint32_t limit_all = j->limit_axes & ~j->locked_axes, limit_axes = limit_all & ~coupled; /* (limit_axes & !coupled_axes, :285-314) */
    float max_bias = w->params.normalized_max_corrective_velocity * w->params.length_unit;
    if (limit_axes & 0x38u) {
        /* limit_angular (joint_constraint_helper.rs:503-564) with recentered_angle (:468-501) */
        quat q1 = frame1.r, q2 = frame2.r;
        float sgn = copysignf(1.0f, qdot(q1, q2));
        quat ang_err = qmul(qconj(q1), q2);
        float imag[3] = {ang_err.x * sgn, ang_err.y * sgn, ang_err.z * sgn}, real = ang_err.w * sgn;
        for (int a = 0; a < 3; ++a) {
            if (!(limit_axes & (8u << a))) continue;
            float c_cos = j->ang_limit_center[a][0], c_sin = j->ang_limit_center[a][1], half_range = j->ang_limit_half_range[a];
            float x = imag[a];
            float sin_half = c_cos * x - c_sin * real;
            float cos_half = c_cos * real + c_sin * x;
            float half = ro_atan2_portable(sin_half, cos_half);
            float shift = copysignf(3.14159265358979323846f, half);
            float wrapped_half = fabsf(half) > 1.5707963267948966f ? half - shift : half;
            float ang = wrapped_half * 2.0f;
            int min_enabled = ang <= -half_range, max_enabled = half_range <= ang;
            v3 ang_jac = col[a];
            JointRow *c = &out[len++];
            c->solver_vel1 = j->solver_body_ids[0]; c->solver_vel2 = j->solver_body_ids[1];
            c->im1 = rb1.im; c->im2 = rb2.im;
            c->impulse = 0.0f; c->impulse_bounds[0] = min_enabled ? -INFINITY : 0.0f; c->impulse_bounds[1] = max_enabled ? INFINITY : 0.0f;
            c->lin_jac = V3(0, 0, 0); c->ang_jac1 = ang_jac; c->ang_jac2 = ang_jac;
            float rhs_wo_bias = 0.0f;
            float rhs_bias = ro_clampf((ro_maxf(ang - half_range, 0.0f) - ro_maxf(-half_range - ang, 0.0f)) * erp_inv_dt, -max_bias, max_bias);
            c->ii_ang_jac1 = sym3_mul(rb1.ii, ang_jac);
            c->ii_ang_jac2 = sym3_mul(rb2.ii, ang_jac);
            c->inv_lhs = 0.0f; c->cfm_coeff = cfm_coeff; c->cfm_gain = 0.0f;
            c->rhs = rhs_wo_bias + rhs_bias; c->rhs_wo_bias = rhs_wo_bias; c->dof = 6 + 3 + a;
        }
    }
    for (int i = 0; i < 3; ++i) {
        if (!(limit_axes & (1u << i))) continue;
        /* limit_linear (:166-208) = lock_linear row with one-sided impulse bounds */
        JointRow *c = &out[len++];
        c->solver_vel1 = j->solver_body_ids[0]; c->solver_vel2 = j->solver_body_ids[1];
        c->im1 = rb1.im; c->im2 = rb2.im;
        c->impulse = 0.0f;
        c->lin_jac = col[i];
        c->ang_jac1 = vadd(vadd(vmul(c1x, col[i].x), vmul(c1y, col[i].y)), vmul(c1z, col[i].z));
        c->ang_jac2 = vadd(vadd(vmul(c2x, col[i].x), vmul(c2y, col[i].y)), vmul(c2z, col[i].z));
        c->ii_ang_jac1 = sym3_mul(rb1.ii, c->ang_jac1);
        c->ii_ang_jac2 = sym3_mul(rb2.ii, c->ang_jac2);
        float dist = vdot(lin_err, c->lin_jac);
        float lmin = j->limits[i][0], lmax = j->limits[i][1];
        int min_enabled = dist <= lmin, max_enabled = lmax <= dist;
        float rhs_wo_bias = 0.0f;
        float rhs_bias = ro_clampf((ro_maxf(dist - lmax, 0.0f) - ro_maxf(lmin - dist, 0.0f)) * erp_inv_dt, -max_bias, max_bias);
        c->inv_lhs = 0.0f; c->cfm_coeff = cfm_coeff; c->cfm_gain = 0.0f;
        c->rhs = rhs_wo_bias + rhs_bias; c->rhs_wo_bias = rhs_wo_bias; c->dof = 6 + i;
        c->impulse_bounds[0] = min_enabled ? -INFINITY : 0.0f; c->impulse_bounds[1] = max_enabled ? INFINITY : 0.0f;
    }
    if ((coupled & 0x38u) && (limit_all & (1u << ro_ctz(coupled & 0x38u)))) {
        /* limit_angular_coupled (joint_constraint_helper.rs:725-790): exactly two coupled angular axes; the angle between the two
         * frames' copies of the THIRD axis is limited — glam 0.33 Quat::from_rotation_arc + to_axis_angle restated (the crate is not
         * under /root/reference: its published algorithm) */
        const int fa = ro_ctz(coupled & 0x38u);
        const uint32_t ca = (coupled >> 3) & 7u;
        int nc = 0; while (ca & (1u << nc)) ++nc; /* trailing_ones: the index of the angular axis that is NOT coupled */
        float m2[3][3]; quat_to_mat(frame2.r, m2);
        v3 axis1 = col[nc], axis2 = V3(m2[0][nc], m2[1][nc], m2[2][nc]);
        quat rot; float d = vdot(axis1, axis2);
        const float one_minus_eps = 1.0f - 2.0f * 1.1920929e-7f;
        if (d > one_minus_eps) rot = Q(0.0f, 0.0f, 0.0f, 1.0f);
        else if (d < -one_minus_eps) { /* from_axis_angle(from.any_orthonormal_vector(), PI) */
            float sign = copysignf(1.0f, axis1.z), a = -1.0f / (sign + axis1.z), b = axis1.x * axis1.y * a;
            v3 o = V3(b, sign + axis1.y * axis1.y * a, -axis1.y);
            rot = Q(o.x * 1.0f, o.y * 1.0f, o.z * 1.0f, -4.371139e-08f); /* sin, cos of PI / 2 in f32 */
        } else {
            v3 cr = vcross(axis1, axis2);
            quat q = Q(cr.x, cr.y, cr.z, 1.0f + d);
            float inv = 1.0f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
            rot = Q(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
        }
        v3 rv = V3(rot.x, rot.y, rot.z), ang_jac; float angle;
        float rl = sqrtf(vdot(rv, rv));
        if (rl >= 1.0e-8f) { angle = 2.0f * ro_atan2_portable(rl, rot.w); ang_jac = vmul(rv, 1.0f / rl); } else { ang_jac = V3(1.0f, 0.0f, 0.0f); angle = 0.0f; }
        if (angle == 0.0f) { /* axis1.orthonormal_basis()[0] (utils/orthonormal_basis.rs:37-50) */
            float sign = copysignf(1.0f, axis1.z), a = -1.0f / (sign + axis1.z), b = axis1.x * axis1.y * a;
            ang_jac = V3(1.0f + sign * axis1.x * axis1.x * a, sign * b, -sign * axis1.x);
        }
        float lmin = j->limits[fa][0], lmax = j->limits[fa][1];
        int min_enabled = angle <= lmin, max_enabled = lmax <= angle;
        JointRow *c = &out[len++];
        c->solver_vel1 = j->solver_body_ids[0]; c->solver_vel2 = j->solver_body_ids[1];
        c->im1 = rb1.im; c->im2 = rb2.im;
        c->impulse = 0.0f; c->impulse_bounds[0] = min_enabled ? -INFINITY : 0.0f; c->impulse_bounds[1] = max_enabled ? INFINITY : 0.0f;
        c->lin_jac = V3(0, 0, 0); c->ang_jac1 = ang_jac; c->ang_jac2 = ang_jac;
        float rhs_bias = ro_clampf((ro_maxf(angle - lmax, 0.0f) - ro_maxf(lmin - angle, 0.0f)) * erp_inv_dt, -max_bias, max_bias);
        c->ii_ang_jac1 = sym3_mul(rb1.ii, ang_jac); c->ii_ang_jac2 = sym3_mul(rb2.ii, ang_jac);
        c->inv_lhs = 0.0f; c->cfm_coeff = cfm_coeff; c->cfm_gain = 0.0f;
        c->rhs = 0.0f + rhs_bias; c->rhs_wo_bias = 0.0f; c->dof = 6 + fa;
    }
    if ((coupled & 7u) && (limit_all & (1u << ro_ctz(coupled & 7u)))) {
        /* limit_linear_coupled (joint_constraint_helper.rs:210-283): the distance along the combined error of the coupled linear
         * axes against the MAX limit of the first coupled axis (RopeJoint; "FIXME: handle min limit too") */
        const int fa = ro_ctz(coupled & 7u);
        v3 lin_jac = V3(0, 0, 0), aj1 = V3(0, 0, 0), aj2 = V3(0, 0, 0);
        for (int i = 0; i < 3; ++i) {
            if (!(coupled & (1u << i))) continue;
            float coeff = vdot(col[i], lin_err);
            lin_jac = vadd(lin_jac, vmul(col[i], coeff));
            aj1 = vadd(aj1, vmul(vadd(vadd(vmul(c1x, col[i].x), vmul(c1y, col[i].y)), vmul(c1z, col[i].z)), coeff));
            aj2 = vadd(aj2, vmul(vadd(vadd(vmul(c2x, col[i].x), vmul(c2y, col[i].y)), vmul(c2z, col[i].z)), coeff));
        }
        float dist = sqrtf(vdot(lin_jac, lin_jac)), inv_dist = ro_inv(dist);
        lin_jac = vmul(lin_jac, inv_dist); aj1 = vmul(aj1, inv_dist); aj2 = vmul(aj2, inv_dist);
        float lmax = j->limits[fa][1];
        float rhs_wo_bias = ro_minf(dist - lmax, 0.0f) * ro_inv(dt);
        float rhs_bias = ro_clampf(ro_maxf(dist - lmax, 0.0f) * erp_inv_dt, -max_bias, max_bias);
        JointRow *c = &out[len++];
        c->solver_vel1 = j->solver_body_ids[0]; c->solver_vel2 = j->solver_body_ids[1];
        c->im1 = rb1.im; c->im2 = rb2.im;
        c->impulse = 0.0f; c->impulse_bounds[0] = 0.0f; c->impulse_bounds[1] = INFINITY;
        c->lin_jac = lin_jac; c->ang_jac1 = aj1; c->ang_jac2 = aj2;
        c->ii_ang_jac1 = sym3_mul(rb1.ii, aj1); c->ii_ang_jac2 = sym3_mul(rb2.ii, aj2);
        c->inv_lhs = 0.0f; c->cfm_coeff = cfm_coeff; c->cfm_gain = 0.0f;
        c->rhs = rhs_wo_bias + rhs_bias; c->rhs_wo_bias = rhs_wo_bias; c->dof = 6 + fa;
    }
    joint_finalize_rows(out + start, len - start);
    return len;
}
/* JointConstraintBuilder::update — joint_constraint_builder.rs:76-150 (row rebuild + warm-start carry) */
static void joint_builder_update(ro_world *w, const Joint *j, float dt, int substep_id) {
    JointRow *rows = &w->joint_rows[j->first_row];
    float prev[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int ws = w->params.warmstart_joints;
    int count = joint_num_rows(j);
    if (ws && substep_id > 0) for (int k = 0; k < count; ++k) prev[k] = rows[k].impulse;
    int len = joint_update_rows(w, j, dt, rows);
    if (ws) {
        float coeff = w->params.warmstart_coefficient;
        for (int k = 0; k < len; ++k) rows[k].impulse = (substep_id == 0 ? (rows[k].dof >= 12 ? j->motor_impulses[rows[k].dof - 12] : rows[k].dof >= 6 ? j->limit_impulses[rows[k].dof - 6] : j->impulses[rows[k].dof]) : prev[k]) * coeff;
    }
}
/* JointConstraint::solve_generic / warmstart_generic — joint_velocity_constraint.rs:97-142 */
static void joint_row_warmstart(ro_world *w, JointRow *c) {
    SolverVel v1, v2; gather_vel(w, c->solver_vel1, &v1); gather_vel(w, c->solver_vel2, &v2);
    v3 lin_impulse = vmul(c->lin_jac, c->impulse);
    v3 ii1 = vmul(c->ii_ang_jac1, c->impulse), ii2 = vmul(c->ii_ang_jac2, c->impulse);
    v1.linear = vadd(v1.linear, vcmul(lin_impulse, c->im1)); v1.angular = vadd(v1.angular, ii1);
    v2.linear = vsub(v2.linear, vcmul(lin_impulse, c->im2)); v2.angular = vsub(v2.angular, ii2);
    scatter_vel(w, c->solver_vel1, &v1); scatter_vel(w, c->solver_vel2, &v2);
}
static void joint_row_solve(ro_world *w, JointRow *c) {
    SolverVel v1, v2; gather_vel(w, c->solver_vel1, &v1); gather_vel(w, c->solver_vel2, &v2);
    float dlinvel = vdot(c->lin_jac, vsub(v2.linear, v1.linear));
    float dangvel = vdot(c->ang_jac2, v2.angular) - vdot(c->ang_jac1, v1.angular);
    float rhs = dlinvel + dangvel + c->rhs;
    float total = ro_clampf(c->impulse + c->inv_lhs * (rhs - c->cfm_gain * c->impulse), c->impulse_bounds[0], c->impulse_bounds[1]);
    float delta = total - c->impulse;
    c->impulse = total;
    v3 lin_impulse = vmul(c->lin_jac, delta);
    v3 ii1 = vmul(c->ii_ang_jac1, delta), ii2 = vmul(c->ii_ang_jac2, delta);
    v1.linear = vadd(v1.linear, vcmul(lin_impulse, c->im1)); v1.angular = vadd(v1.angular, ii1);
    v2.linear = vsub(v2.linear, vcmul(lin_impulse, c->im2)); v2.angular = vsub(v2.angular, ii2);
    scatter_vel(w, c->solver_vel1, &v1); scatter_vel(w, c->solver_vel2, &v2);
}
/* The joint part of solve_pass — staged_island_solver/solve.rs:31-150: every joint (parallel colours
 * ascending, then the serial overflow) solves BEFORE any contact in every pass. */
static void joint_solve_all_rows(ro_world *w, const Joint *j, int wo_bias, int warmstart_joints) {
    int count = joint_num_rows(j);
    for (int k = 0; k < count; ++k) {
        JointRow *c = &w->joint_rows[j->first_row + k];
        if (wo_bias) c->rhs = c->rhs_wo_bias;
        if (warmstart_joints) joint_row_warmstart(w, c);
        joint_row_solve(w, c);
    }
}
/* `order` / `nparallel` / `ntotal`: the group's own layout (single_group_joint_layout applied to the joints of one solve group) */
static void joints_solve_pass(ro_world *w, const int *order, int nparallel, int ntotal, int wo_bias, int warmstart_joints) {
    int a = 0;
    while (a < nparallel) { /* one parallel colour = one body-disjoint stage */
        int c = w->joints[order[a]].solver_color, e = a;
        while (e < nparallel && w->joints[order[e]].solver_color == c) ++e;
        RO_PARALLEL_FOR
        for (int i = a; i < e; ++i) joint_solve_all_rows(w, &w->joints[order[i]], wo_bias, warmstart_joints);
        a = e;
    }
    for (; a < ntotal; ++a) joint_solve_all_rows(w, &w->joints[order[a]], wo_bias, warmstart_joints);
}

/* Substep solve-groups — island_manager/substep_groups.rs:44-229: the awake set is partitioned by the effective
 * RigidBody::additional_solver_iterations: connected components of awake DYNAMIC bodies over pairs with an active contact and over
 * joints take the maximum extra count of their members; a kinematic body is lifted to the largest count among the dynamic bodies it
 * touches; one group per distinct count, in descending order.  Returns the number of groups; key_of_body[solver id] = group index. */
#define RO_MAX_GROUPS 16
/* union-find over small index sets; the root of a set is its smallest member */
static int uf_find(int *uf, int x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; }
static void uf_union(int *uf, int a, int b) { a = uf_find(uf, a); b = uf_find(uf, b); if (a == b) return; if (a < b) uf[b] = a; else uf[a] = b; }
static int compute_solve_groups(ro_world *w, int nd, int *group_of_body, int *group_extra) {
    int any_extra = 0;
    for (int i = 0; i < nd; ++i) any_extra |= w->bodies[w->dyn_bodies[i]].additional_solver_iterations > 0;
    if (!any_extra) { for (int i = 0; i < nd; ++i) group_of_body[i] = 0; group_extra[0] = 0; return 1; }
    int *uf = (int *)malloc(sizeof(int) * (size_t)(nd + 1)), *key = (int *)calloc((size_t)nd + 1, sizeof(int));
    for (int i = 0; i < nd; ++i) uf[i] = i;
#define RO_DYN_SLOT(b) ((b) >= 0 && w->bodies[b].body_type == RO_BODY_DYNAMIC && w->bodies[b].solver_id != RO_NO_BODY ? (int)w->bodies[b].solver_id : -1)
#define RO_KIN_SLOT(b) ((b) >= 0 && w->bodies[b].body_type != RO_BODY_DYNAMIC && w->bodies[b].body_type != RO_BODY_FIXED && w->bodies[b].solver_id != RO_NO_BODY ? (int)w->bodies[b].solver_id : -1)
    for (int i = 0; i < w->npairs; ++i) {
        const Pair *p = &w->pairs[i];
        if (!p->alive || p->nsc == 0) continue;
        int s1 = RO_DYN_SLOT(w->colliders[p->c1].parent), s2 = RO_DYN_SLOT(w->colliders[p->c2].parent);
        if (s1 >= 0 && s2 >= 0) uf_union(uf, s1, s2);
    }
    for (int i = 0; i < w->njoints; ++i) {
        const Joint *j = &w->joints[i];
        if (j->removed) continue;
        int s1 = RO_DYN_SLOT(j->body1), s2 = RO_DYN_SLOT(j->body2);
        if (s1 >= 0 && s2 >= 0) uf_union(uf, s1, s2);
    }
    for (int i = 0; i < nd; ++i) {
        int extra = w->bodies[w->dyn_bodies[i]].additional_solver_iterations;
        if (extra > 0) { int r = uf_find(uf, i); if (key[r] < extra) key[r] = extra; }
    }
    for (int i = 0; i < nd; ++i) if (uf_find(uf, i) != i) key[i] = key[uf_find(uf, i)];
    for (int i = 0; i < w->npairs; ++i) { /* lift the kinematic bodies */
        const Pair *p = &w->pairs[i];
        if (!p->alive || p->nsc == 0) continue;
        int b1 = w->colliders[p->c1].parent, b2 = w->colliders[p->c2].parent;
        int k1 = RO_KIN_SLOT(b1), d2 = RO_DYN_SLOT(b2), k2 = RO_KIN_SLOT(b2), d1 = RO_DYN_SLOT(b1);
        if (k1 >= 0 && d2 >= 0 && key[k1] < key[d2]) key[k1] = key[d2];
        if (k2 >= 0 && d1 >= 0 && key[k2] < key[d1]) key[k2] = key[d1];
    }
    for (int i = 0; i < w->njoints; ++i) {
        const Joint *j = &w->joints[i];
        if (j->removed) continue;
        int k1 = RO_KIN_SLOT(j->body1), d2 = RO_DYN_SLOT(j->body2), k2 = RO_KIN_SLOT(j->body2), d1 = RO_DYN_SLOT(j->body1);
        if (k1 >= 0 && d2 >= 0 && key[k1] < key[d2]) key[k1] = key[d2];
        if (k2 >= 0 && d1 >= 0 && key[k2] < key[d1]) key[k2] = key[d1];
    }
#undef RO_DYN_SLOT
#undef RO_KIN_SLOT
    int ng = 0;
    for (int i = 0; i < nd; ++i) { /* distinct keys, descending */
        int seen = 0;
        for (int g = 0; g < ng; ++g) seen |= group_extra[g] == key[i];
        if (!seen && ng < RO_MAX_GROUPS) group_extra[ng++] = key[i];
    }
    for (int a = 1; a < ng; ++a) { int v = group_extra[a], b = a - 1; while (b >= 0 && group_extra[b] < v) { group_extra[b + 1] = group_extra[b]; --b; } group_extra[b + 1] = v; }
    for (int i = 0; i < nd; ++i) { group_of_body[i] = ng - 1; for (int g = 0; g < ng; ++g) if (group_extra[g] == key[i]) { group_of_body[i] = g; break; } }
    free(uf); free(key);
    return ng;
}

static void solve_velocity_constraints(ro_world *w) {
    const ro_params *prm = &w->params;
    const int base_substeps = prm->num_solver_iterations;

    /* active set = awake dynamic bodies in arena order, manager.rs:20-39 */
    int nd = 0;
    for (int i = 0; i < w->nbodies; ++i) if (body_is_active(w, i)) nd++;
    /* maintain_solver_contact_graph — solver_graph.rs:129-361: buckets of active manifolds per colour,
     * full-rebuild order = ascending (edge, manifold) */
    int counts[RO_NUM_COLORS]; memset(counts, 0, sizeof(counts));
    int M = 0, nsc = 0;
    for (int i = 0; i < w->npairs; ++i) {
        Pair *p = &w->pairs[i];
        if (!pair_selected(w, p)) continue;
        counts[p->color]++; M++; nsc += p->nsc;
        /* the pair's further solver manifolds (clusters 2+ with solver contacts) go to the overflow colour (solver_graph.rs:534-547) */
        for (int k = 1; k < p->ncl; ++k) if (p->ex[k - 1].nsc > 0) { counts[RO_COLOR_OVERFLOW]++; M++; nsc += p->ex[k - 1].nsc; }
    }
    solver_reserve(w, nd, M);
    nd = 0;
    for (int i = 0; i < w->nbodies; ++i) {
        Body *b = &w->bodies[i];
        if (body_is_active(w, i)) { b->solver_id = (uint32_t)nd; w->dyn_bodies[nd++] = i; } else b->solver_id = RO_NO_BODY;
    }
    w->ndyn = nd;
    w->bucket_begin[0] = 0;
    for (int c = 0; c < RO_NUM_COLORS; ++c) w->bucket_begin[c + 1] = w->bucket_begin[c] + counts[c];
    int cursor[RO_NUM_COLORS]; memcpy(cursor, w->bucket_begin, sizeof(cursor));
    int *order = (int *)malloc(sizeof(int) * (M + 1));
    for (int i = 0; i < w->npairs; ++i) {
        Pair *p = &w->pairs[i];
        if (!pair_selected(w, p)) continue;
        /* qualify_manifold_bqi — solver_contact_graph.rs:101-124 */
        int b1 = w->colliders[p->c1].parent, b2 = w->colliders[p->c2].parent;
        p->solver_body_ids[0] = b1 >= 0 ? w->bodies[b1].solver_id : RO_NO_BODY;
        p->solver_body_ids[1] = b2 >= 0 ? w->bodies[b2].solver_id : RO_NO_BODY;
        order[cursor[p->color]++] = i;
        for (int k = 1; k < p->ncl; ++k) if (p->ex[k - 1].nsc > 0) order[cursor[RO_COLOR_OVERFLOW]++] = i | (k << RO_SM_SHIFT);
    }
    /* The overflow colour is swept serially and is not body-disjoint, so its order is part of the result.  The reference orders it
     * by its body-mask regrouping of contact-graph edge ids (interaction_groups.rs:240-355), which depend on the BVH's pair creation
     * order; here — and on the device — it is swept in ascending (collider1, collider2) order, a total order that no creation order
     * can change (deliberate deviations). */
    if (counts[RO_COLOR_OVERFLOW] > 1) {
        int ob = w->bucket_begin[RO_COLOR_OVERFLOW], on = counts[RO_COLOR_OVERFLOW];
        for (int a = 1; a < on; ++a) { /* insertion sort: the bucket is nearly sorted (pairs are created in sweep order); key (collider1, collider2, cluster) */
            int v = order[ob + a]; const Pair *pv = &w->pairs[RO_SM_PAIR(v)];
            uint64_t kv = ((uint64_t)(uint32_t)pv->c1 << 32) | (uint32_t)pv->c2; int cv = RO_SM_K(v);
            int b = a - 1;
            while (b >= 0) {
                const Pair *pb = &w->pairs[RO_SM_PAIR(order[ob + b])];
                uint64_t kb = ((uint64_t)(uint32_t)pb->c1 << 32) | (uint32_t)pb->c2; int cb = RO_SM_K(order[ob + b]);
                if (kb < kv || (kb == kv && cb <= cv)) break;
                order[ob + b + 1] = order[ob + b]; --b;
            }
            order[ob + b + 1] = v;
        }
    }
    /* chunk layout — init.rs:163-254: colours with >= 32 four-lane chunks first (ascending),
     * then the smaller colours (ascending), then the overflow colour. */
    w->nstages = 0; int nparallel = 0, nused = 0;
    for (int c = 0; c < RO_NUM_COLORS - 1; ++c) if ((counts[c] + 3) / 4 >= 32) { w->stage_color[w->nstages++] = c; nparallel++; }
    for (int c = 0; c < RO_NUM_COLORS - 1; ++c) if (counts[c] > 0 && (counts[c] + 3) / 4 < 32) w->stage_color[w->nstages++] = c;
    if (counts[RO_COLOR_OVERFLOW] > 0) w->stage_color[w->nstages++] = RO_COLOR_OVERFLOW;
    for (int c = 0; c < RO_NUM_COLORS; ++c) if (counts[c] > 0) nused++;
    w->stats.num_active_manifolds = M; w->stats.num_solver_contacts = nsc;
    w->stats.num_colors_used = nused; w->stats.num_parallel_colors = nparallel; w->stats.num_pairs = w->npairs;
    w->ncons = M;

    /* substep solve-groups (substep_groups.rs; init.rs:52-100, 163-420): group g runs base + extra[g] substeps at its own dt, over
     * its own bodies, joints and constraints; a constraint belongs to the highest group index (= lowest cadence) among its solver
     * bodies, i.e. to its dynamic side.  Without any elevated body there is one group and everything below is the plain loop. */
    int *grp_body = (int *)malloc(sizeof(int) * (size_t)(nd + 1)), *grp_cons = (int *)malloc(sizeof(int) * (size_t)(M + 1));
    int g_extra[RO_MAX_GROUPS];
    const int ngroups = compute_solve_groups(w, nd, grp_body, g_extra);
    for (int i = 0; i < w->nbodies; ++i) w->bodies[i].last_group_extra = -1;
    for (int i = 0; i < nd; ++i) w->bodies[w->dyn_bodies[i]].last_group_extra = g_extra[grp_body[i]];
    for (int i = 0; i < M; ++i) {
        const Pair *p = &w->pairs[RO_SM_PAIR(order[i])];
        int g = 0;
        for (int k = 0; k < 2; ++k) if (p->solver_body_ids[k] != RO_NO_BODY && grp_body[p->solver_body_ids[k]] > g) g = grp_body[p->solver_body_ids[k]];
        grp_cons[i] = g;
    }
    /* per-group chunk layout — init.rs:163-254 applied to the group's share of every colour bucket */
    int (*g_stage_color)[RO_NUM_COLORS + 1] = (int (*)[RO_NUM_COLORS + 1])malloc(sizeof(int[RO_NUM_COLORS + 1]) * (size_t)ngroups);
    int g_nstages[RO_MAX_GROUPS];
    for (int g = 0; g < ngroups; ++g) {
        int cg[RO_NUM_COLORS]; memset(cg, 0, sizeof(cg));
        if (ngroups == 1) memcpy(cg, counts, sizeof(cg));
        else for (int c = 0; c < RO_NUM_COLORS; ++c) for (int i = w->bucket_begin[c]; i < w->bucket_begin[c + 1]; ++i) cg[c] += grp_cons[i] == g;
        int ns = 0;
        for (int c = 0; c < RO_NUM_COLORS - 1; ++c) if ((cg[c] + 3) / 4 >= 32) g_stage_color[g][ns++] = c;
        for (int c = 0; c < RO_NUM_COLORS - 1; ++c) if (cg[c] > 0 && (cg[c] + 3) / 4 < 32) g_stage_color[g][ns++] = c;
        if (cg[RO_COLOR_OVERFLOW] > 0) g_stage_color[g][ns++] = RO_COLOR_OVERFLOW;
        g_nstages[g] = ns;
    }

    /* S0: solver bodies + increments — worker.rs:46-104, solver_body.rs:82-121 */
    RO_PARALLEL_FOR
    for (int i = 0; i < nd; ++i) {
        Body *rb = &w->bodies[w->dyn_bodies[i]];
        w->flags[i] = rb->allow_fast_rotation ? 1 : 0;
        w->vels[i].angular = rb->angvel; w->vels[i].linear = rb->linvel;
        w->poses[i].rotation = rb->position.r;
        w->poses[i].translation = pose_tp(rb->position, rb->local_com);
        w->poses[i].ii = rb->effective_world_inv_inertia;
        w->poses[i].im = rb->effective_inv_mass;
        const float dt_b = prm->dt / (float)(base_substeps + g_extra[grp_body[i]]); /* the substep length of the body's group */
        w->incr[i].angular = vmul(sym3_mul(rb->effective_world_inv_inertia, rb->torque), dt_b);
        w->incr[i].linear = vmul(vcmul(rb->force, rb->effective_inv_mass), dt_b);
        Gyro *g = &w->gyro[i];
        if (rb->gyroscopic && rb->body_type == RO_BODY_DYNAMIC) { /* worker.rs:86 */
            g->inv_principal_inertia = rb->inv_principal_inertia;
            g->principal_inertia = V3(ro_inv(rb->inv_principal_inertia.x), ro_inv(rb->inv_principal_inertia.y), ro_inv(rb->inv_principal_inertia.z));
            g->principal_frame = rb->principal_frame; g->enabled = 1;
        } else g->enabled = 0;
    }
    /* S1: generate — worker.rs:109-190.  Constraint i lives at bucket position i. */
    int any_bouncy = 0;
    const int coulomb = prm->friction_model == RO_FRICTION_COULOMB; /* init.rs:419 */
    RO_PARALLEL_FOR
    for (int i = 0; i < M; ++i) { if (coulomb) coulomb_generate(w, order[i], &w->cons[i]); else constraint_generate(w, order[i], &w->cons[i]); }
    for (int i = 0; i < M; ++i)
        for (int k = 0; k < w->cons[i].num_contacts; ++k) any_bouncy |= w->cons[i].infos[k].restitution_seed < 0.0f;
    free(order);
    /* joints: selection, colouring in the contacts' colour space, builders — init_joints, joints.rs:25-329 */
    int num_joint_rows = 0;
    if (w->njoints > 0) {
        w->joint_body_colors = (u128 *)realloc(w->joint_body_colors, sizeof(u128) * (size_t)(nd + 1));
        masks_reserve(w, w->nbodies + 1);
        joints_select_active(w);
        joints_color(w);
        for (int a = 0; a < w->nactive_joints; ++a) joint_builder_generate(w, &w->joints[w->active_joints[a]], &num_joint_rows);
        w->joint_rows = (JointRow *)realloc(w->joint_rows, sizeof(JointRow) * (size_t)(num_joint_rows + 1));
    } else w->nactive_joints = 0;

    /* joints per group (a joint follows its dynamic side like a contact does) and the group's own single_group_joint_layout */
    int *grp_joint = (int *)malloc(sizeof(int) * (size_t)(w->nactive_joints + 1));
    int *g_joint_order = (int *)malloc(sizeof(int) * (size_t)(w->nactive_joints + 1));
    int g_joint_begin[RO_MAX_GROUPS + 1], g_joint_parallel[RO_MAX_GROUPS];
    for (int a = 0; a < w->nactive_joints; ++a) {
        const Joint *j = &w->joints[w->active_joints[a]];
        int g = 0;
        for (int k = 0; k < 2; ++k) if (j->solver_body_ids[k] != RO_NO_BODY && grp_body[j->solver_body_ids[k]] > g) g = grp_body[j->solver_body_ids[k]];
        grp_joint[a] = g;
    }
    {
        int n = 0;
        for (int g = 0; g < ngroups; ++g) {
            g_joint_begin[g] = n; g_joint_parallel[g] = 0;
            if (ngroups == 1) { /* the layout joints_color() already built */
                for (int a = 0; a < w->nactive_joints; ++a) g_joint_order[n++] = w->joint_order[a];
                g_joint_parallel[g] = w->njoint_parallel;
                continue;
            }
            int cj[RO_NUM_COLORS]; memset(cj, 0, sizeof(cj));
            for (int a = 0; a < w->nactive_joints; ++a) if (grp_joint[a] == g) cj[w->joints[w->active_joints[a]].solver_color]++;
            for (int pass = 0; pass < 2; ++pass)
                for (int c = 0; c < RO_NUM_COLORS; ++c) {
                    int parallel = c < 128 && cj[c] >= 64;
                    if (cj[c] == 0 || (pass == 0) != parallel) continue;
                    for (int a = 0; a < w->nactive_joints; ++a)
                        if (grp_joint[a] == g && w->joints[w->active_joints[a]].solver_color == c) g_joint_order[n++] = w->active_joints[a];
                    if (pass == 0) g_joint_parallel[g] = n - g_joint_begin[g];
                }
        }
        g_joint_begin[ngroups] = n;
    }

    int solve_friction_in_bias = prm->friction_in_bias_pass || prm->num_internal_stabilization_iterations == 0;
    float max_lin = prm->normalized_max_linear_velocity * prm->length_unit;
    float max_ang = 0.78539816339744830962f * (prm->dt == 0.0f ? 0.0f : 1.0f / prm->dt);
    /* groups in descending cadence, each with its whole substep loop (a kinematic body is integrated with the highest-cadence group
     * it touches, before any lower-cadence group solves against it) */
    for (int grp = 0; grp < ngroups; ++grp) {
    const int num_substeps = base_substeps + g_extra[grp];
    const float dt_s = prm->dt / (float)num_substeps;
    const int *jorder = g_joint_order + g_joint_begin[grp];
    const int jpar = g_joint_parallel[grp], jtot = g_joint_begin[grp + 1] - g_joint_begin[grp];
    const int one_group = ngroups == 1;
    for (int s = 0; s < num_substeps; ++s) {
        float solved_dt = (float)s * dt_s;
        /* S2 increments + gyroscopic — worker.rs:235-284 */
        RO_PARALLEL_FOR
        for (int i = 0; i < nd; ++i) {
            if (!one_group && grp_body[i] != grp) continue;
            w->vels[i].linear = vadd(w->vels[i].linear, w->incr[i].linear);
            w->vels[i].angular = vadd(w->vels[i].angular, w->incr[i].angular);
            if (w->gyro[i].enabled) {
                quat axes = qmul(w->poses[i].rotation, w->gyro[i].principal_frame);
                w->vels[i].angular = gyroscopic_corrected_angvel(w->vels[i].angular, axes, w->gyro[i].principal_inertia,
                                                                 w->gyro[i].inv_principal_inertia, dt_s);
            }
        }
        /* S3 joint rows rebuilt from the current poses — worker.rs:287-357 */
        RO_PARALLEL_FOR
        for (int a = 0; a < w->nactive_joints; ++a) { if (one_group || grp_joint[a] == grp) joint_builder_update(w, &w->joints[w->active_joints[a]], dt_s, s); }
        /* S4 fused update + warmstart per colour — worker.rs:438-538 (non-fused when coefficient == 0) */
        for (int st = 0; st < g_nstages[grp]; ++st) {
            int c = g_stage_color[grp][st];
            int serial = c == RO_COLOR_OVERFLOW; /* the overflow colour is not body-disjoint */
            RO_PRAGMA_IF_PAR(serial)
            for (int i = w->bucket_begin[c]; i < w->bucket_begin[c + 1]; ++i) {
                if (!one_group && grp_cons[i] != grp) continue;
                if (coulomb) { coulomb_update(w, &w->cons[i], dt_s, solved_dt); if (prm->warmstart_coefficient != 0.0f) coulomb_warmstart(w, &w->cons[i]); continue; }
                constraint_update(w, &w->cons[i], dt_s, solved_dt);
                if (prm->warmstart_coefficient != 0.0f) constraint_warmstart(w, &w->cons[i]);
            }
        }
        /* S5 biased pass — worker.rs:544-561, staged_island_solver/solve.rs:12-209 */
        for (int it = 0; it < prm->num_internal_pgs_iterations; ++it) {
            joints_solve_pass(w, jorder, jpar, jtot, 0, prm->warmstart_joints && it == 0);
            for (int st = 0; st < g_nstages[grp]; ++st) {
                int c = g_stage_color[grp][st];
                int serial = c == RO_COLOR_OVERFLOW;
                RO_PRAGMA_IF_PAR(serial)
                for (int i = w->bucket_begin[c]; i < w->bucket_begin[c + 1]; ++i) {
                    if (!one_group && grp_cons[i] != grp) continue;
                    if (coulomb) coulomb_solve(w, &w->cons[i], solve_friction_in_bias); else constraint_solve(w, &w->cons[i], solve_friction_in_bias);
                }
            }
        }
        /* S6 integrate — worker.rs:568-631, rigid_body_components.rs:884-898 */
        RO_PARALLEL_FOR
        for (int i = 0; i < nd; ++i) {
            if (!one_group && grp_body[i] != grp) continue;
            SolverVel *v = &w->vels[i];
            if (max_lin != FLT_MAX) { float n = vlen(v->linear); if (n > max_lin) v->linear = vmul(v->linear, max_lin / n); }
            if (!(w->flags[i] & 1)) { float n = vlen(v->angular); if (n > max_ang) v->angular = vmul(v->angular, max_ang / n); }
            v3 hang = vmul(v->angular, dt_s * 0.5f);
            quat q = qmul(Q(hang.x, hang.y, hang.z, 1.0f), w->poses[i].rotation);
            w->poses[i].rotation = qnormalize(q);
            w->poses[i].translation = vadd(w->poses[i].translation, vmul(v->linear, dt_s));
        }
        /* S7 unbiased pass with refreshed rhs — worker.rs:636-649 */
        for (int it = 0; it < prm->num_internal_stabilization_iterations; ++it) {
            joints_solve_pass(w, jorder, jpar, jtot, 1, 0);
            for (int st = 0; st < g_nstages[grp]; ++st) {
                int c = g_stage_color[grp][st];
                int serial = c == RO_COLOR_OVERFLOW;
                RO_PRAGMA_IF_PAR(serial)
                for (int i = w->bucket_begin[c]; i < w->bucket_begin[c + 1]; ++i) {
                    if (!one_group && grp_cons[i] != grp) continue;
                    if (coulomb) { coulomb_refresh_rhs_wo_bias(w, &w->cons[i], dt_s, solved_dt + dt_s); coulomb_solve(w, &w->cons[i], 1); continue; }
                    constraint_refresh_rhs_wo_bias(w, &w->cons[i], dt_s, solved_dt + dt_s);
                    constraint_solve(w, &w->cons[i], 1);
                }
            }
        }
    }
    }
    /* S8 restitution — worker.rs:657-734 (group by group, in the group's own stage order) */
    if (any_bouncy)
        for (int grp = 0; grp < ngroups; ++grp)
            for (int st = 0; st < g_nstages[grp]; ++st) {
                int c = g_stage_color[grp][st];
                for (int i = w->bucket_begin[c]; i < w->bucket_begin[c + 1]; ++i) if (ngroups == 1 || grp_cons[i] == grp) constraint_apply_restitution(w, &w->cons[i]);
            }
    free(grp_body); free(grp_cons); free(grp_joint); free(g_joint_order); free(g_stage_color);
    /* S9 impulse writeback — worker.rs:742-802 */
    RO_PARALLEL_FOR
    for (int i = 0; i < M; ++i) { if (coulomb) coulomb_writeback(w, &w->cons[i]); else constraint_writeback(w, &w->cons[i]); }
    /* JointConstraintsSet::writeback_impulses — joint_velocity_constraint.rs:346-353 */
    for (int a = 0; a < w->nactive_joints; ++a) {
        Joint *j = &w->joints[w->active_joints[a]];
        int nrows = joint_num_rows(j);
        for (int k = 0; k < nrows; ++k) { /* WritebackId::Dof(i) / WritebackId::Limit(i) */
            const JointRow *r = &w->joint_rows[j->first_row + k];
            if (r->dof >= 12) j->motor_impulses[r->dof - 12] = r->impulse; else if (r->dof >= 6) j->limit_impulses[r->dof - 6] = r->impulse; else j->impulses[r->dof] = r->impulse;
        }
    }
    /* S10 body writeback — worker.rs:809-897 */
    for (int i = 0; i < nd; ++i) {
        Body *rb = &w->bodies[w->dyn_bodies[i]];
        rb->linvel = vmul(w->vels[i].linear, 1.0f / (1.0f + prm->dt * rb->linear_damping));
        rb->angvel = vmul(w->vels[i].angular, 1.0f / (1.0f + prm->dt * rb->angular_damping));
        pose sp; sp.r = w->poses[i].rotation; sp.t = w->poses[i].translation;
        if (rb->body_type == RO_BODY_KINEMATIC_POSITION) continue; /* :836-842 keep exactly the pose the user asked for */
        /* pose.prepend_translation(-local_com) */
        rb->next_position.r = sp.r;
        rb->next_position.t = vadd(sp.t, qrot(sp.r, vneg(rb->local_com)));
        /* CCD activation (worker.rs:845-865): RigidBodyCcd::is_moving_fast_with_next_position (rigid_body_components.rs:1131-1157) with
         * ccd_vels = interpolate_velocity(inv_dt) of the solved motion; dynamic bodies only (ccd_solver.rs:66) */
        rb->ccd_active = 0;
        if (prm->max_ccd_substeps != 0 && rb->body_type == RO_BODY_DYNAMIC && rb->ccd_thickness < 3.0e38f) {
            v3 dcom = vsub(sp.t, rb->world_com);
            quat dq = qmul(sp.r, qconj(rb->position.r));
            v3 dv = V3(dq.x, dq.y, dq.z);
            float inv_dt = prm->dt == 0.0f ? 0.0f : 1.0f / prm->dt;
            float max_delta = vlen(dcom) + 2.0f * vlen(dv) * rb->max_extent;
            float max_vel = vlen(vmul(dcom, inv_dt)) + vlen(vmul(quat_to_scaled_axis(dq), inv_dt)) * rb->max_extent;
            float max_motion = ro_maxf(max_delta, max_vel * prm->dt);
            if (max_motion > 0.5f * rb->ccd_thickness) { rb->ccd_active = 1; w->ccd_active_count++; }
        }
    }
}

/* ---- CCDSolver::solve_continuous — dynamics/ccd/ccd_solver.rs:158-340, sweeps.rs:470-640 (the time-of-impact query itself: ro_ccd.h) ----
 * Fast non-bullet bodies sweep the FIXED colliders (parentless or on a fixed body), then bullets (ccd_enabled) sweep every collider
 * that is not on a bullet, targets standing at their — possibly just clamped — next_position; the earliest solid impact clamps the
 * body's next_position (apply_clamps :325-339), velocities are untouched.  Sensors and colliders whose groups do not match never
 * stop a body (the paired intersection events of sensor crossings, :265-320, are not raised). */
static CcdShape ccd_shape_of(const Collider *c) { return sm_shape_of(c); }
static int ccd_is_bullet(const Body *b) { return b->body_type == RO_BODY_DYNAMIC && b->ccd_enabled; } /* sweeps.rs:29-31 */
static void ccd_sweep_tier(ro_world *w, int bullets) {
    const float slop = w->params.normalized_allowed_linear_error * w->params.length_unit; /* IntegrationParameters::allowed_linear_error */
    for (int bi = 0; bi < w->nbodies; ++bi) {
        Body *rb1 = &w->bodies[bi];
        if (!rb1->ccd_active || rb1->sleeping || ccd_is_bullet(rb1) != bullets) continue;
        CcdSweep sw = ccd_sweep_from_poses(rb1->position, rb1->next_position, rb1->local_com);
        float fraction = 1.0f;
        for (int f = 0; f < w->ncolliders; ++f) {
            const Collider *co1 = &w->colliders[f];
            if (co1->parent != bi || !collider_enabled(co1) || co1->sensor) continue;
            if (co1->shape == RO_SHAPE_TRIMESH) continue; /* a mesh is never the fast shape (sweeps.rs:86-97) */
            /* a compound is swept child by child (FastShapeKind::Compound, sweeps.rs:337-345): each part with the part's own pose on the body */
            const int nparts1 = co_is_composite(co1) ? co_num_subs(co1) : 1;
            for (int part = 0; part < nparts1; ++part) {
            Collider prim1 = *co1; pose pwp1 = co1->pos_wrt_parent;
            if (co_is_composite(co1)) { pose pp; int hp; co_sub(co1, part, &prim1, &pp, &hp); pwp1 = pose_mul(co1->pos_wrt_parent, pp); }
            CcdShape s2 = ccd_shape_of(&prim1);
            const float rot_radius = ccd_rot_radius(&s2, pwp1, rb1->local_com);
            for (int t = 0; t < w->ncolliders; ++t) {
                const Collider *co2 = &w->colliders[t];
                if (t == f || co2->parent == bi || !collider_enabled(co2) || co2->sensor || co2->sub != co1->sub) continue;
                const Body *rb2 = co2->parent >= 0 ? &w->bodies[co2->parent] : NULL;
                /* tier_allows (sweeps.rs:35-41): a non-bullet only meets fixed targets, a bullet everything but bullets */
                if (bullets) { if (rb2 && ccd_is_bullet(rb2)) continue; } else if (rb2 && rb2->body_type != RO_BODY_FIXED) continue;
                if (!((co1->memberships & co2->filter) != 0 && (co2->memberships & co1->filter) != 0)) continue; /* collision_groups.test */
                /* target_collider_pose (:97-102): stationary at its end-of-step pose */
                pose tp = (rb2 && rb2->body_type != RO_BODY_FIXED) ? pose_mul(rb2->next_position, co2->pos_wrt_parent) : co2->pos;
                if (co_is_composite(co2)) {
                    /* a composite target (sweeps.rs:255-262, :384-400: sweep_time_of_impact_composite): every sub-shape whose box meets the
                     * swept volume's box — the centre-of-mass segment inflated by max_extent + 2 slop, taken into the composite's frame —
                     * is a target of its own; the earliest fraction wins (round 5: composites used to be skipped) */
                    const float reach = rb1->max_extent + 2.0f * slop;
                    v3 qmn = V3(fminf(sw.c0.x, sw.c1.x) - reach, fminf(sw.c0.y, sw.c1.y) - reach, fminf(sw.c0.z, sw.c1.z) - reach);
                    v3 qmx = V3(fmaxf(sw.c0.x, sw.c1.x) + reach, fmaxf(sw.c0.y, sw.c1.y) + reach, fmaxf(sw.c0.z, sw.c1.z) + reach);
                    v3 qc = vmul(vadd(qmn, qmx), 0.5f), qh = vmul(vsub(qmx, qmn), 0.5f);
                    v3 lc = pose_itp(tp, qc);
                    float m[3][3]; quat_to_mat(tp.r, m);
                    v3 lh = V3(fabsf(m[0][0]) * qh.x + fabsf(m[1][0]) * qh.y + fabsf(m[2][0]) * qh.z,
                               fabsf(m[0][1]) * qh.x + fabsf(m[1][1]) * qh.y + fabsf(m[2][1]) * qh.z,
                               fabsf(m[0][2]) * qh.x + fabsf(m[1][2]) * qh.y + fabsf(m[2][2]) * qh.z);
                    const int nsub = co_num_subs(co2);
                    for (int i = 0; i < nsub; ++i) {
                        const Aabb a = co_sub_aabb(co2, i);
                        if (a.mins.x > lc.x + lh.x || a.maxs.x < lc.x - lh.x || a.mins.y > lc.y + lh.y || a.maxs.y < lc.y - lh.y || a.mins.z > lc.z + lh.z || a.maxs.z < lc.z - lh.z) continue;
                        Collider prim; pose pp; int hp; co_sub(co2, i, &prim, &pp, &hp);
                        pose tpose = hp ? pose_mul(tp, pp) : tp;
                        if (hp && !ccd_may_reach(sw.c0, sw.c1, rb1->max_extent, tpose.t, shape_bounding_radius(&prim), 2.0f * slop)) continue;
                        CcdShape s1 = ccd_shape_of(&prim);
                        float hit = ccd_cast_pair(&s1, tpose, &s2, pwp1, &sw, rot_radius, fraction, slop);
                        if (hit == -2.0f) { /* starts on this sub-shape: the core ball's turn (ro_ccd.h: ccd_core_of) */
                            CcdShape core = ccd_core_of(&s2);
                            hit = ccd_cast_pair(&s1, tpose, &core, pwp1, &sw, ccd_rot_radius(&core, pwp1, rb1->local_com), fraction, slop);
                        }
                        if (hit > 0.0f && hit < fraction) fraction = hit;
                    }
                    continue;
                }
                if (co2->shape != RO_SHAPE_HALFSPACE && !ccd_may_reach(sw.c0, sw.c1, rb1->max_extent, tp.t, shape_bounding_radius(co2), 2.0f * slop)) continue;
                CcdShape s1 = ccd_shape_of(co2);
                float hit = ccd_cast_pair(&s1, tp, &s2, pwp1, &sw, rot_radius, fraction, slop);
                if (hit > 0.0f && hit < fraction) fraction = hit;
            }
            }
        }
        if (fraction < 1.0f) { rb1->next_position = ccd_sweep_transform_at(&sw, fraction); w->ccd_clamp_count++; }
    }
}
static void ccd_solve_continuous(ro_world *w) {
    int any = 0;
    for (int i = 0; i < w->nbodies && !any; ++i) any = w->bodies[i].ccd_active;
    if (!any) return;
    ccd_sweep_tier(w, 0);
    ccd_sweep_tier(w, 1);
}

/* ---- Persistent islands: connectivity maintenance — island_manager/{persistent,local_split,global_split}.rs ------------------
 *
 * The reference keeps one PersistentIsland per connected component of the touching-contact / joint graph over the non-fixed
 * bodies, merged EAGERLY (link_contact / link_joint, union by size) and split LAZILY: an unlinked edge is journaled and resolved
 * at the top of the next solve by a bounded local search (still connected: nothing happens; detached: the smaller piece moves out
 * at once; both endpoints moving fast, budget exceeded, sleeping island or a removed body: constraint_remove_count > 0), and an
 * island with constraint_remove_count > 0 may NOT sleep (finish_sleep_scan, persistent.rs:498-516) until the deferred global
 * union-find split — one island per step, chosen by the sleepiest eligible body's bid, then SPLIT_RETRY_COOLDOWN steps of rest —
 * has cleared it.  All of that is restated here; what the decisions READ is the same: the partition, the body counts, the flag,
 * the cooldown stamp.
 *
 * What cannot be taken from /root/reference is the ORDER in which the reference visits links: its link vectors and its
 * transition sort are keyed by contact-graph edge ids, which are handed out in the pair-creation order of parry's BVH traversal
 * (not in the tree).  Wherever that order can change an outcome the rule below is canonical (independent of any creation order,
 * the same on the device), and ro_read_island_stats counts how often a scene exercised it:
 *   - a merge group of more than two islands in one step keeps the identity (id, cooldown stamp, pending-split status) of its
 *     LARGEST member, the smaller id on equal size (reference: a tournament of pairwise union-by-size merges in edge order; the
 *     island of collider1's parent wins equal sizes) — RO_IS_MULTIWAY_GROUPS;
 *   - absorbed ids are freed in ascending order (reference: in merge order);
 *   - journal order: joint unlinks (joint index), broad-phase deletions, end-touch transitions (collider pair key) — it matters
 *     only when two detaching removals hit one island in one step — RO_IS_ORDER_DEPENDENT;
 *   - a detached component of equal size: body1 = the parent of the pair's smaller collider handle (reference: of collider1, whose
 *     order is the BVH's) — RO_IS_DETACH_SIZE_TIES;
 *   - a still-connected verdict is never over budget (the reference's expansion count up to the meeting point depends on its
 *     adjacency order; a detached verdict costs exactly 2·min(|C0|, |C1|) (+1) expansions and IS budgeted like the reference) —
 *     differs only inside components of >= 1024 bodies;
 *   - the global split keeps the largest component in the base island, the one with the smallest body on ties (reference: the first
 *     root in island-vector order), and creates the others in ascending order of their smallest body (reference: island-vector
 *     order) — RO_IS_SPLIT_KEEP_TIES;
 *   - a bid tie goes to the larger island id like the reference (solve.rs:206-211); the ids themselves are canonical as above —
 *     RO_IS_BID_TIES. */
static int pi_member(const ro_world *w, int body) { return body >= 0 && w->bodies[body].body_type != RO_BODY_FIXED && w->bodies[body].island_id >= 0; }

/* connected components of the CURRENT touching / joint graph over the awake non-fixed bodies (IslandGraph::for_each_neighbor,
 * local_split.rs:80-118): uf[i] = smallest body of i's component */
static void pi_components(ro_world *w, int *uf) {
    for (int i = 0; i < w->nbodies; ++i) uf[i] = i;
    for (int i = 0; i < w->npairs; ++i) {
        const Pair *p = &w->pairs[i];
        if (!p->alive || p->nsc == 0) continue; /* has_any_active_contact (queries.rs:230) */
        int b1 = w->colliders[p->c1].parent, b2 = w->colliders[p->c2].parent;
        if (body_is_active(w, b1) && body_is_active(w, b2)) uf_union(uf, b1, b2);
    }
    for (int i = 0; i < w->njoints; ++i) {
        const Joint *j = &w->joints[i];
        if (!j->removed && body_is_active(w, j->body1) && body_is_active(w, j->body2)) uf_union(uf, j->body1, j->body2);
    }
    for (int i = 0; i < w->nbodies; ++i) uf[i] = uf_find(uf, i);
}

/* ImpulseJointIslandEvent::Link events, drained in insertion order at the top of the step (substep.rs:357-362) -> link_joint ->
 * merge_islands (persistent.rs:361-393, :420-461): pairwise union by size, the island of body1 survives equal sizes, the absorbed
 * id is freed at once.  Joint events are ordered by in-tree code, so this is the reference's own sequence. */
static void pi_link_joints(ro_world *w) {
    for (int k = 0; k < w->njoints; ++k) {
        Joint *j = &w->joints[k];
        if (j->linked || j->removed) continue;
        j->linked = 1;
        if (!pi_member(w, j->body1) || !pi_member(w, j->body2)) continue; /* a fixed side does not connect */
        int a = w->bodies[j->body1].island_id, b = w->bodies[j->body2].island_id;
        if (a == b) continue;
        int big = w->isl[a].nbodies >= w->isl[b].nbodies ? a : b, small = big == a ? b : a;
        for (int i = 0; i < w->nbodies; ++i) if (w->bodies[i].island_id == small) w->bodies[i].island_id = big;
        w->isl[big].nbodies += w->isl[small].nbodies; w->isl[big].dirty |= w->isl[small].dirty; w->isl[big].sleeping &= w->isl[small].sleeping;
        w->isl[small].nbodies = 0; pi_free(w, small);
        w->pi_stats[RO_IS_MERGED]++;
    }
}

/* link_contact -> merge_islands (persistent.rs:293-329, :420-461), for every touching pair whose endpoints sit in different
 * islands after this step's transitions and wake-ups.  Already-linked edges join nothing, so walking all touching pairs equals
 * walking the step's begin-touch transitions. */
static void pi_merge_links(ro_world *w) {
    int any = 0;
    for (int i = 0; i < w->npairs && !any; ++i) {
        const Pair *p = &w->pairs[i];
        if (!p->alive || p->nsc == 0) continue;
        int b1 = w->colliders[p->c1].parent, b2 = w->colliders[p->c2].parent;
        any = pi_member(w, b1) && pi_member(w, b2) && w->bodies[b1].island_id != w->bodies[b2].island_id;
    }
    if (!any) return;
    int n = w->isl_next;
    int *uf = (int *)malloc(sizeof(int) * (size_t)(4 * n + 4)), *win = uf + n + 1, *cnt = uf + 2 * n + 2, *start_size = uf + 3 * n + 3;
    for (int i = 0; i < n; ++i) { uf[i] = i; win[i] = -1; cnt[i] = 0; start_size[i] = w->isl[i].used ? w->isl[i].nbodies : 0; }
    for (int i = 0; i < w->npairs; ++i) {
        const Pair *p = &w->pairs[i];
        if (!p->alive || p->nsc == 0) continue;
        int b1 = w->colliders[p->c1].parent, b2 = w->colliders[p->c2].parent;
        if (pi_member(w, b1) && pi_member(w, b2)) uf_union(uf, w->bodies[b1].island_id, w->bodies[b2].island_id);
    }
    /* the identity that survives a group: its largest island at the start of the step, the smaller id on equal size */
    for (int i = 0; i < n; ++i) {
        if (!w->isl[i].used) continue;
        int r = uf_find(uf, i); cnt[r]++;
        if (win[r] < 0 || start_size[i] > start_size[win[r]]) win[r] = i; /* ascending i: ties keep the smaller id */
    }
    for (int i = 0; i < n; ++i) {
        if (!w->isl[i].used) continue;
        int r = uf_find(uf, i), k = win[r];
        if (cnt[r] > 2 && k == i) w->pi_stats[RO_IS_MULTIWAY_GROUPS]++;
        if (k == i) continue;
        /* merge_islands: counts add (the flag ORs), sleeping ANDs, the absorbed island is freed (dropping a pending split of it) */
        w->isl[k].nbodies += w->isl[i].nbodies; w->isl[k].dirty |= w->isl[i].dirty; w->isl[k].sleeping &= w->isl[i].sleeping;
        w->pi_stats[RO_IS_MERGED]++;
    }
    for (int i = 0; i < w->nbodies; ++i) { Body *b = &w->bodies[i]; if (b->island_id >= 0) b->island_id = win[uf_find(uf, b->island_id)]; }
    for (int i = 0; i < n; ++i) if (w->isl[i].used && win[uf_find(uf, i)] != i) { w->isl[i].nbodies = 0; pi_free(w, i); } /* ascending id order */
    free(uf);
}

static int removal_cmp(const void *a, const void *b) {
    const Removal *x = (const Removal *)a, *y = (const Removal *)b;
    if (x->phase != y->phase) return x->phase - y->phase;
    return x->key < y->key ? -1 : x->key > y->key;
}
/* resolve_removals — local_split.rs:164-255 */
static void pi_resolve_removals(ro_world *w) {
    if (w->njournal == 0) return;
    qsort(w->journal, (size_t)w->njournal, sizeof(Removal), removal_cmp);
    int n = w->nbodies;
    int *comp = (int *)malloc(sizeof(int) * (size_t)(4 * n + 4)), *size = comp + n + 1, *comp_isl = comp + 2 * n + 2, *detaches = comp + 3 * n + 3;
    pi_components(w, comp);
    for (int i = 0; i < n; ++i) { size[i] = 0; comp_isl[i] = -1; }
    for (int i = 0; i < w->isl_next; ++i) if (i < n) detaches[i] = 0;
    int ndet_isl = w->isl_next < n ? w->isl_next : n;
    for (int i = 0; i < n; ++i) if (body_is_active(w, i) && w->bodies[i].island_id >= 0) { size[comp[i]]++; comp_isl[comp[i]] = w->bodies[i].island_id; }
    const float length_unit = w->params.length_unit;
    for (int k = 0; k < w->njournal; ++k) {
        const Removal *r = &w->journal[k];
        w->pi_stats[RO_IS_REMOVALS]++;
        /* :186-196 an endpoint that is fixed or gone carried no connectivity */
        if (!pi_member(w, r->body1) || !pi_member(w, r->body2)) continue;
        const Body *ba = &w->bodies[r->body1], *bb = &w->bodies[r->body2];
        /* the island of an endpoint: awake bodies follow their component (an earlier removal of this batch may have moved it out) */
        int i1 = ba->sleeping ? ba->island_id : comp_isl[comp[r->body1]], i2 = bb->sleeping ? bb->island_id : comp_isl[comp[r->body2]];
        if (i1 != i2) continue;                                   /* :198-202 */
        PIsland *isl = &w->isl[i1];
        if (isl->sleeping) { isl->dirty = 1; w->pi_stats[RO_IS_SLEEPING_DEFERRED]++; continue; } /* :207-210 */
        /* :212-231 both endpoints above the sleep speed (the farthest-point metric of the sleep energy): defer to the global split */
        int hot = 1;
        for (int e = 0; e < 2; ++e) {
            const Body *b = e ? bb : ba;
            float lin_threshold = b->normalized_linear_threshold * length_unit;
            if (lin_threshold < 0.0f) continue; /* never sleeps: always hot */
            float max_point_vel = sqrtf(vdot(b->linvel, b->linvel)) + sqrtf(vdot(b->angvel, b->angvel)) * b->max_extent;
            if (!(max_point_vel > lin_threshold)) hot = 0;
        }
        if (hot) { isl->dirty = 1; w->pi_stats[RO_IS_HOT]++; continue; }
        /* search (:371-413): a lockstep dual flood from both endpoints.  Meeting = still connected; otherwise side 0 runs dry
         * after |C0| expansions of either side if |C0| <= |C1|, side 1 after |C1| + 1 and |C1| expansions if |C1| < |C0| */
        int c1 = comp[r->body1], c2 = comp[r->body2];
        if (c1 == c2) { w->pi_stats[RO_IS_CONNECTED]++; continue; }
        int side = size[c1] <= size[c2] ? 0 : 1;
        int expansions = side == 0 ? 2 * size[c1] : 2 * size[c2] + 1;
        if (expansions >= RO_SEARCH_BUDGET) { isl->dirty = 1; w->pi_stats[RO_IS_OVER_BUDGET]++; continue; }
        if (size[c1] == size[c2]) w->pi_stats[RO_IS_DETACH_SIZE_TIES]++;
        /* move_component_out (:260-345): a fresh island takes the detached component and its links */
        int c = side == 0 ? c1 : c2;
        int id = pi_alloc(w);
        isl = &w->isl[i1]; /* pi_alloc may have moved the table */
        w->isl[id].nbodies = size[c]; w->isl[id].sleeping = isl->sleeping;
        isl->nbodies -= size[c];
        comp_isl[c] = id;
        w->pi_stats[RO_IS_DETACHED]++;
        if (i1 < ndet_isl && ++detaches[i1] == 2) w->pi_stats[RO_IS_ORDER_DEPENDENT]++;
    }
    for (int i = 0; i < n; ++i) if (body_is_active(w, i) && w->bodies[i].island_id >= 0) w->bodies[i].island_id = comp_isl[comp[i]];
    w->njournal = 0;
    free(comp);
}

/* run_pending_split -> split_island_now — global_split.rs:44-308 */
static void pi_run_pending_split(ro_world *w) {
    int id = w->pending_split;
    w->pending_split = -1;                                        /* take() */
    if (id < 0 || !w->isl[id].used || w->isl[id].sleeping) return; /* :46-53 splits only run on awake islands */
    PIsland *isl = &w->isl[id];
    w->pi_stats[RO_IS_GLOBAL_SPLITS]++;
    int n = w->nbodies, ncomp = 0, keep = -1, nkeep = 0;
    int *comp = NULL, *size = NULL;
    if (isl->nbodies > 1) {
        comp = (int *)malloc(sizeof(int) * (size_t)(2 * n + 2)); size = comp + n + 1;
        pi_components(w, comp);
        for (int i = 0; i < n; ++i) size[i] = 0;
        for (int i = 0; i < n; ++i) if (w->bodies[i].island_id == id) size[body_is_active(w, i) ? comp[i] : i]++;
        for (int i = 0; i < n; ++i) { /* ascending smallest body: the largest component keeps the base island */
            if (!size[i]) continue;
            ncomp++;
            if (keep < 0 || size[i] > size[keep]) { keep = i; nkeep = 1; } else if (size[i] == size[keep]) nkeep++;
        }
    }
    if (ncomp > 1) {
        if (nkeep > 1) w->pi_stats[RO_IS_SPLIT_KEEP_TIES]++;
        int base_sleeping = isl->sleeping;
        for (int c = 0; c < n; ++c) {
            if (!size[c] || c == keep) continue;
            int nid = pi_alloc(w);
            w->isl[nid].nbodies = size[c]; w->isl[nid].sleeping = base_sleeping;
            for (int i = 0; i < n; ++i) if (w->bodies[i].island_id == id && (body_is_active(w, i) ? comp[i] : i) == c) w->bodies[i].island_id = nid;
            w->pi_stats[RO_IS_GLOBAL_SPLIT_PIECES]++;
        }
        isl = &w->isl[id];
        isl->nbodies = size[keep];
    }
    /* every outcome re-arms the cooldown and clears the count (:69-74, :156-162, :302-305) */
    isl->dirty = 0; isl->denied = w->scan_stamp + RO_SPLIT_RETRY_COOLDOWN;
    free(comp);
}

/* The island part of build_islands_and_solve_velocity_constraints (solve.rs:159-300) and IslandManager::update_islands
 * (manager.rs:335-388): removals resolved, the pending split run, then the fused pass over the awake bodies — sleep timers
 * (update_body_energy), the split bid of the sleepiest eligible body of an island that may bid (split_allowed), the per-island
 * observation — and the whole-island sleep decision behind the constraint_remove_count gate. */
static void update_sleep(ro_world *w) {
    const float dt = w->params.dt, length_unit = w->params.length_unit;
    int n = w->nbodies, any_can_sleep = 0;
    pi_resolve_removals(w);
    pi_run_pending_split(w);
    /* update_body_energy (manager.rs:320-333) -> RigidBodyActivation::update_energy (rigid_body_components.rs:1412-1478) */
    for (int i = 0; i < n; ++i) {
        Body *b = &w->bodies[i];
        if (!body_is_active(w, i)) continue;
        if (b->body_type != RO_BODY_DYNAMIC) { /* platforms only sleep when both velocities are exactly zero (:1464-1468) */
            int still = vdot(b->linvel, b->linvel) == 0.0f && vdot(b->angvel, b->angvel) == 0.0f;
            if (still) b->time_since_can_sleep += dt; else b->time_since_can_sleep = 0.0f;
            any_can_sleep |= 1;
            continue;
        }
        float linear_threshold = b->normalized_linear_threshold * length_unit;
        pose prev = b->sleep_prev_pose; b->sleep_prev_pose = b->position;
        float sq_angvel = vdot(b->angvel, b->angvel);
        int angular_ok;
        if (b->max_extent > 0.0f) angular_ok = b->angular_threshold >= 0.0f && sq_angvel < 1.5707964f * 1.5707964f;
        else angular_ok = sq_angvel < b->angular_threshold * fabsf(b->angular_threshold);
        float drift = relative_pose_drift(prev, b->position, b->max_extent);
        int can_sleep = angular_ok && drift * 0.5f < linear_threshold * dt;
        if (can_sleep) b->time_since_can_sleep += dt; else b->time_since_can_sleep = 0.0f;
        any_can_sleep |= b->normalized_linear_threshold >= 0.0f;
    }
    /* split bid (solve.rs:200-237): max (time_since_can_sleep, island id) over the eligible awake bodies whose island has pending
     * removals and is out of its cooldown; the winner is next step's (single) pending split (:293-295) */
    int bid_island = -1, bid_tie = 0; float bid_score = 0.0f;
    int observed = 0;
    for (int i = 0; i < n; ++i) {
        const Body *b = &w->bodies[i];
        if (!body_is_active(w, i) || b->island_id < 0) continue;
        observed = 1;
        if (!(b->time_since_can_sleep >= b->time_until_sleep)) continue;
        const PIsland *isl = &w->isl[b->island_id];
        if (!(isl->dirty && w->scan_stamp >= isl->denied)) continue; /* split_allowed (persistent.rs:181-186) */
        if (bid_island < 0 || b->time_since_can_sleep > bid_score) { bid_score = b->time_since_can_sleep; bid_island = b->island_id; bid_tie = 0; }
        else if (b->time_since_can_sleep == bid_score && b->island_id != bid_island) { bid_tie = 1; if (b->island_id > bid_island) bid_island = b->island_id; }
    }
    if (bid_island >= 0) { w->pending_split = bid_island; w->pi_stats[RO_IS_BIDS]++; if (bid_tie) w->pi_stats[RO_IS_BID_TIES]++; }
    if (!observed) return;
    /* begin_sleep_scan / observe_body_for_sleep / finish_sleep_scan (persistent.rs:463-516): an island sleeps once EVERY awake
     * body of it is eligible — unless it lost constraints and holds more than one body: it must split first */
    w->scan_stamp++;
    if (!any_can_sleep) return;
    int *awake = (int *)malloc(sizeof(int) * (size_t)(w->isl_next + 1));
    for (int i = 0; i < w->isl_next; ++i) awake[i] = 0;
    for (int i = 0; i < n; ++i) {
        const Body *b = &w->bodies[i];
        if (body_is_active(w, i) && b->island_id >= 0 && !(b->time_since_can_sleep >= b->time_until_sleep)) awake[b->island_id] = 1;
    }
    for (int i = 0; i < w->isl_next; ++i) {
        PIsland *isl = &w->isl[i];
        if (!isl->used || isl->sleeping || awake[i]) { awake[i] = 1; continue; }
        if (isl->dirty && isl->nbodies > 1) { awake[i] = 1; w->pi_stats[RO_IS_SLEEP_BLOCKED]++; }
    }
    /* mark_island_sleeping + commit_sleeping_chunks -> RigidBody::sleep (rigid_body.rs:804-807) + clear_asleep_pair_solver_hint_counts_of */
    for (int i = 0; i < n; ++i) {
        Body *b = &w->bodies[i];
        if (!body_is_active(w, i) || b->island_id < 0 || awake[b->island_id]) continue;
        b->sleeping = 1; b->time_since_can_sleep = b->time_until_sleep;
        b->linvel = V3(0, 0, 0); b->angvel = V3(0, 0, 0);
        b->slept_at = w->step_seq;
        w->isl[b->island_id].sleeping = 1;
        if (w->pending_split == b->island_id) w->pending_split = -1; /* clear_pending_split_of */
    }
    free(awake);
}

/* NarrowPhase::emit_contact_force_events — solver_graph.rs:462-498; ContactForceEvent::from_contact_pair — geometry/mod.rs:223-258 */
static void emit_contact_force_events(ro_world *w) {
    float dt = w->params.dt, inv_dt = dt == 0.0f ? 0.0f : 1.0f / dt;
    for (int i = 0; i < w->npairs; ++i) {
        Pair *p = &w->pairs[i];
        const Collider *a = &w->colliders[p->c1], *b = &w->colliders[p->c2];
        float ta = (a->active_events & 2u) ? a->force_threshold : FLT_MAX, tb = (b->active_events & 2u) ? b->force_threshold : FLT_MAX;
        float threshold = ta < tb ? ta : tb;
        if (!(threshold < FLT_MAX) || !pair_selected(w, p)) continue; /* force_event_pairs: solver-active pairs with force events enabled */
        float total = 0.0f; /* over every solver manifold of the pair (ContactPair::solver_manifolds: the plain manifold or its clusters) */
        for (int q = 0; q < pair_num_sm(p); ++q) { const Manifold *mq = sm_m(p, q); for (int k = 0; k < mq->npoints; ++k) total += mq->points[k].data.impulse; }
        float total_magnitude = (0.0f + total) * inv_dt;
        if (total_magnitude > threshold) {
            if (w->nforce_events == w->cap_force_events) {
                w->cap_force_events = w->cap_force_events ? 2 * w->cap_force_events : 256;
                w->force_meta = (int32_t *)realloc(w->force_meta, sizeof(int32_t) * 4 * (size_t)w->cap_force_events);
                w->force_vals = (float *)realloc(w->force_vals, sizeof(float) * 8 * (size_t)w->cap_force_events);
            }
            int32_t *m = w->force_meta + 4 * w->nforce_events; float *v = w->force_vals + 8 * w->nforce_events; w->nforce_events++;
            m[0] = p->c1; m[1] = p->c2; m[2] = w->step_seq; m[3] = !p->force_emitted;
            float max_mag = 0.0f; v3 max_dir = V3(0, 0, 0); v3 total_force = V3(0, 0, 0);
            for (int q = 0; q < pair_num_sm(p); ++q) {
                const Manifold *mq = sm_m(p, q); const v3 nq = *sm_normal(p, q);
                float tmi = 0.0f;
                for (int k = 0; k < mq->npoints; ++k) {
                    float imp = mq->points[k].data.impulse;
                    tmi += imp;
                    if (imp > max_mag) { max_mag = imp; max_dir = nq; }
                }
                total_force = vadd(total_force, vmul(nq, tmi));
            }
            total_force = vmul(total_force, inv_dt);
            v[0] = total_force.x; v[1] = total_force.y; v[2] = total_force.z; v[3] = total_magnitude;
            v[4] = max_dir.x; v[5] = max_dir.y; v[6] = max_dir.z; v[7] = max_mag * inv_dt;
            p->force_emitted = 1;
        } else p->force_emitted = 0;
    }
}

/* IslandManager::persistent_island_of (manager.rs:214-220) */
void ro_read_island_labels(ro_world *w, int32_t *out) {
    for (int i = 0; i < w->nbodies; ++i) out[i] = w->bodies[i].body_type == RO_BODY_FIXED ? -1 : w->bodies[i].island_id;
}
void ro_read_island_state(const ro_world *w, int32_t island, int32_t out5[5]) {
    PIsland z = {0, 0, 0, 0, 0};
    const PIsland *i = island >= 0 && island < w->isl_next ? &w->isl[island] : &z;
    out5[0] = i->used; out5[1] = i->nbodies; out5[2] = i->dirty; out5[3] = i->denied; out5[4] = i->sleeping;
}
void ro_read_island_globals(const ro_world *w, int32_t out2[2]) { out2[0] = w->scan_stamp; out2[1] = w->pending_split; }
/* RigidBodyBuilder::ccd_enabled / RigidBody::enable_ccd: the body becomes a bullet (sweeps.rs:29-31) */
void ro_set_ccd_enabled(ro_world *w, int32_t body, int32_t on) { if (body >= 0 && body < w->nbodies) w->bodies[body].ccd_enabled = on != 0; }
void ro_read_ccd_counts(const ro_world *w, int32_t out2[2]) { out2[0] = w->ccd_active_count; out2[1] = w->ccd_clamp_count; }
void ro_read_island_stats(const ro_world *w, int32_t *out) { memcpy(out, w->pi_stats, sizeof(w->pi_stats)); }
void ro_read_slept_at(const ro_world *w, int32_t *out) { for (int i = 0; i < w->nbodies; ++i) out[i] = w->bodies[i].slept_at; }

/* PhysicsPipeline::step_inner — pipeline/physics_pipeline/substep.rs:267-581 */
/* RO_PROFILE=1: wall time per phase of a step, accumulated (printed by ro_profile_dump; tools only) */
#include <time.h>
static double ro_prof_t[16]; static int ro_prof_on = -1;
static double ro_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
#define RO_PROF(k) do { if (ro_prof_on > 0) { double n_ = ro_now(); ro_prof_t[k] += n_ - t_prof; t_prof = n_; } } while (0)
void ro_profile_dump(void) {
    static const char *nm[] = {"wakes+joint links", "broad phase", "narrow phase", "merge links+kinematic", "sleep", "forces", "solve", "force events+ccd", "advance+aabbs"};
    for (int k = 0; k < 9; ++k) fprintf(stderr, "  %-24s %9.3f ms\n", nm[k], ro_prof_t[k] * 1e3);
    for (int k = 0; k < 16; ++k) ro_prof_t[k] = 0.0;
}
static void step_once(ro_world *w) {
    if (ro_prof_on < 0) ro_prof_on = getenv("RO_PROFILE") ? 1 : 0;
    double t_prof = ro_prof_on > 0 ? ro_now() : 0.0;
    w->step_seq++;
    apply_wakes(w);     /* user wake-ups precede the joint edits (substep.rs:288-300) */
    pi_link_joints(w);
    RO_PROF(0);
    /* detect_collisions — solve.rs:45-157 (user-requested wake-ups and pair deletions take effect before the
     * narrow phase reads the awake mask) */
    broad_phase_update(w);
    RO_PROF(1);
    apply_wakes(w);
    narrow_phase_compute_contacts(w);
    RO_PROF(2);
    apply_wakes(w);
    pi_merge_links(w); /* link_contact of this step's begin-touch transitions */
    /* interpolate_kinematic_velocities — substep.rs:242-264, RigidBodyPosition::interpolate_velocity
     * (rigid_body_components.rs:147-194) */
    for (int i = 0; i < w->nbodies; ++i) {
        Body *b = &w->bodies[i];
        if (b->body_type != RO_BODY_KINEMATIC_POSITION || b->sleeping) continue;
        float inv_dt = w->params.dt == 0.0f ? 0.0f : 1.0f / w->params.dt;
        v3 com = pose_tp(b->position, b->local_com);
        pose shift = pose_ident(); shift.t = com;
        pose dpos = pose_mul(pose_mul(pose_mul(pose_inv(shift), b->next_position), pose_inv(b->position)), shift);
        b->linvel = vmul(dpos.t, inv_dt);
        b->angvel = vmul(quat_to_scaled_axis(dpos.r), inv_dt);
    }
    RO_PROF(3);
    /* fused body pass — solve.rs:234-291: sleep timers, then the island sleep decision (update_islands) */
    update_sleep(w);
    RO_PROF(4);
    /* compute_effective_force_and_torque rigid_body_components.rs:1030-1033 */
    for (int i = 0; i < w->nbodies; ++i) {
        Body *b = &w->bodies[i];
        if (!body_is_active(w, i)) continue;
        v3 mass = V3(ro_inv(b->effective_inv_mass.x), ro_inv(b->effective_inv_mass.y), ro_inv(b->effective_inv_mass.z));
        b->force = vadd(b->user_force, vmul(vcmul(w->gravity, mass), b->gravity_scale));
        b->torque = b->user_torque;
    }
    RO_PROF(5);
    solve_velocity_constraints(w);
    RO_PROF(6);
    emit_contact_force_events(w);
    /* run_ccd_motion_clamping (substep.rs:54-82, :496-519): only when some body moved fast */
    if (w->params.max_ccd_substeps != 0) ccd_solve_continuous(w);
    RO_PROF(7);
    /* advance_to_final_positions — substep.rs:84-224; refresh_moved_collider_aabbs :229-240 */
    for (int i = 0; i < w->nbodies; ++i) {
        Body *b = &w->bodies[i];
        if (!body_is_active(w, i)) continue;
        b->position = b->next_position;
        update_world_mass_properties(b);
    }
    for (int i = 0; i < w->ncolliders; ++i) {
        Collider *c = &w->colliders[i];
        if (!body_is_active(w, c->parent)) continue;
        c->pos = pose_mul(w->bodies[c->parent].position, c->pos_wrt_parent);
        bp_set_aabb(w, c);
    }
    RO_PROF(8);
}

void ro_step(ro_world *w, int32_t nsteps) { for (int i = 0; i < nsteps; ++i) step_once(w); }
void ro_get_stats(const ro_world *w, ro_stats *out) { *out = w->stats; }

float ro_total_contact_impulse(const ro_world *w) {
    float total = 0.0f;
    for (int i = 0; i < w->npairs; ++i) {
        const Pair *p = &w->pairs[i];
        float s = 0.0f;
        for (int q = 0; q < pair_num_sm((Pair *)p); ++q) { const Manifold *mq = sm_m((Pair *)p, q); for (int k = 0; k < mq->npoints; ++k) s += mq->points[k].data.impulse; }
        total += s;
    }
    return total;
}
int32_t ro_dump_manifolds(const ro_world *w, int32_t cap, int32_t *meta, float *normal3, float *impulses4) {
    int n = 0;
    for (int i = 0; i < w->npairs; ++i) {
        Pair *p = (Pair *)&w->pairs[i];
        if (p->nsc == 0) continue;
        for (int q = 0; q < pair_num_sm(p); ++q) { /* every solver manifold of the pair: the first in the pair's colour, the others in the overflow colour */
            const int nsc = *sm_nsc(p, q);
            if (nsc == 0) continue;
            if (n < cap) {
                const Manifold *m = sm_m(p, q); const SolverContact *sc = sm_sc(p, q); const v3 nq = *sm_normal(p, q);
                if (meta) { meta[4 * n] = p->c1; meta[4 * n + 1] = p->c2; meta[4 * n + 2] = q == 0 ? p->color : RO_COLOR_OVERFLOW; meta[4 * n + 3] = nsc; }
                if (normal3) { normal3[3 * n] = nq.x; normal3[3 * n + 1] = nq.y; normal3[3 * n + 2] = nq.z; }
                if (impulses4) for (int k = 0; k < 4; ++k) impulses4[4 * n + k] = k < nsc ? m->points[sc[k].cid].data.impulse : 0.0f;
            }
            n++;
        }
    }
    return n;
}
/* ImpulseJointSet::insert — impulse_joint_set.rs.  Scope: locked linear axes only (spherical joints),
 * contacts between the two bodies enabled. */
int32_t ro_add_joint(ro_world *w, const ro_joint_desc *d) {
    const int b1 = (int)(uint32_t)(d->body1 & 0xffffffffu), b2 = (int)(uint32_t)(d->body2 & 0xffffffffu); /* (a handle's index part: the oracle keeps no generations) */
    if (b1 < 0 || b2 < 0 || b1 >= w->nbodies || b2 >= w->nbodies) return -1;
    if ((d->locked_axes & ~0x3fu) != 0 || (d->limit_axes & ~0x3fu) != 0 || (d->motor_axes & ~0x3fu) != 0 || (d->coupled_axes & ~0x3fu) != 0) return -1;
    { uint32_t ca = (d->coupled_axes >> 3) & 7u; if (ca != 0 && ca != 3 && ca != 5 && ca != 6) return -1; } /* limit_angular_coupled: exactly two coupled angular axes (joint_constraint_helper.rs:737-739) */
    if (w->njoints == w->cap_joints) {
        w->cap_joints = w->cap_joints ? w->cap_joints * 2 : 1024;
        w->joints = (Joint *)realloc(w->joints, sizeof(Joint) * w->cap_joints);
        w->active_joints = (int *)realloc(w->active_joints, sizeof(int) * w->cap_joints);
        w->joint_order = (int *)realloc(w->joint_order, sizeof(int) * w->cap_joints);
    }
    Joint *j = &w->joints[w->njoints];
    memset(j, 0, sizeof(*j));
    j->body1 = b1; j->body2 = b2;
    j->local_frame1.t = V3(d->local_anchor1[0], d->local_anchor1[1], d->local_anchor1[2]);
    j->local_frame2.t = V3(d->local_anchor2[0], d->local_anchor2[1], d->local_anchor2[2]);
    j->local_frame1.r = qnormalize(Q(d->local_basis1[0], d->local_basis1[1], d->local_basis1[2], d->local_basis1[3]));
    j->local_frame2.r = qnormalize(Q(d->local_basis2[0], d->local_basis2[1], d->local_basis2[2], d->local_basis2[3]));
    j->locked_axes = d->locked_axes; j->contacts_enabled = d->contacts_enabled;
    j->coupled_axes = d->coupled_axes & 0x3fu;
    j->limit_axes = d->limit_axes & 0x3fu;
    for (int i = 0; i < 6; ++i) { j->limits[i][0] = d->limits[i][0]; j->limits[i][1] = d->limits[i][1]; }
    for (int a = 0; a < 3; ++a) { /* AngularLimitParams::new(min, max) */
        float mn = j->limits[3 + a][0], mx = j->limits[3 + a][1];
        float half_range = (mx - mn) * 0.5f;
        if (half_range >= 3.14159265358979323846f || half_range != half_range) { j->ang_limit_center[a][0] = 1.0f; j->ang_limit_center[a][1] = 0.0f; j->ang_limit_half_range[a] = 10.0f; }
        else { float center = (mn + mx) * 0.5f; j->ang_limit_center[a][0] = cosf(center * 0.5f); j->ang_limit_center[a][1] = sinf(center * 0.5f); j->ang_limit_half_range[a] = half_range; }
    }
    j->motor_axes = d->motor_axes & 0x3fu;
    for (int i = 0; i < 6; ++i) j->motors[i] = d->motors[i];
    j->solver_color = 255; /* default_solver_color: uncoloured */
    w->nc_dirty = 1;
    wake_request(w, b1, 1); wake_request(w, b2, 1); /* insert(.., wake_up = true), substep.rs:289-300 */
    return w->njoints++;
}
/* GenericJoint::set_motor (generic_joint.rs) on ImpulseJointSet::get_mut(handle, wake_up = true) (impulse_joint_set.rs):
 * the axis' motor is enabled, its accumulated impulse kept, both bodies woken */
int32_t ro_set_joint_motor(ro_world *w, int32_t joint, int32_t axis, const ro_joint_motor *m) {
    if (joint < 0 || joint >= w->njoints || w->joints[joint].removed || axis < 0 || axis >= 6) return -1;
    Joint *j = &w->joints[joint];
    j->motor_axes |= 1u << axis;
    j->motors[axis] = *m;
    wake_request(w, j->body1, 1); wake_request(w, j->body2, 1);
    return 0;
}
/* Collider::set_sensor (collider.rs) and NarrowPhase::intersection_pair (narrow_phase/queries.rs:167): -1 = no such pair */
void ro_set_collider_sensor(ro_world *w, int32_t collider, int32_t on) { if (collider >= 0 && collider < w->ncolliders) w->colliders[collider].sensor = on != 0; }
int32_t ro_intersection_pair(const ro_world *w, int32_t c1, int32_t c2) {
    int lo = c1 < c2 ? c1 : c2, hi = c1 < c2 ? c2 : c1;
    for (int i = 0; i < w->npairs; ++i) if (w->pairs[i].alive && w->pairs[i].c1 == lo && w->pairs[i].c2 == hi) return (w->colliders[lo].sensor || w->colliders[hi].sensor) ? w->pairs[i].intersecting : -1;
    return -1;
}
/* RigidBody::set_additional_solver_iterations (rigid_body.rs): extra substeps for the body's connected component */
void ro_set_additional_solver_iterations(ro_world *w, int32_t body, int32_t n) {
    if (body >= 0 && body < w->nbodies) w->bodies[body].additional_solver_iterations = n < 0 ? 0 : n;
}
/* IslandManager::solve_groups as seen from the bodies: the extra substep count of each body's group in the last step */
void ro_read_solve_group_extras(const ro_world *w, int32_t *out) { for (int i = 0; i < w->nbodies; ++i) out[i] = w->bodies[i].last_group_extra; }
void ro_read_joint_motor_impulses(const ro_world *w, float *impulses6) {
    for (int i = 0; i < w->njoints; ++i) for (int a = 0; a < 6; ++a) impulses6[6 * i + a] = w->joints[i].motor_impulses[a];
}
/* ImpulseJointSet::remove (impulse_joint_set.rs:574-...): the joint stops being selected. */
int32_t ro_remove_joint(ro_world *w, int32_t joint) {
    if (joint < 0 || joint >= w->njoints || w->joints[joint].removed) return -1;
    w->joints[joint].removed = 1;
    w->nc_dirty = 1;
    if (w->joints[joint].linked) pi_journal(w, w->joints[joint].body1, w->joints[joint].body2, 0, (uint64_t)(uint32_t)joint); /* ImpulseJointIslandEvent::Unlink -> unlink_joint: a no-op for a joint that was never linked (persistent.rs:396-399) */
    wake_request(w, w->joints[joint].body1, 1); wake_request(w, w->joints[joint].body2, 1); /* remove(.., wake_up = true) */
    memset(w->joints[joint].impulses, 0, sizeof(w->joints[joint].impulses));
    return 0;
}
/* ColliderSet::remove (collider_set.rs) + NarrowPhase::handle_user_changes removing its pairs
 * (pair_management.rs:24-203): the collider keeps its arena slot but can no longer form pairs, and the
 * next broad-phase pass deletes its pairs (DeletePair frees their colours) before any contact is computed. */
int32_t ro_remove_collider(ro_world *w, int32_t collider) {
    if (collider < 0 || collider >= w->ncolliders) return -1;
    Collider *c = &w->colliders[collider];
    if (c->memberships == 0 && c->filter == 0) return -1;
    c->memberships = 0; c->filter = 0;
    w->bp_dirty = 1;
    if (c->parent >= 0) recompute_mass_properties(w, &w->bodies[c->parent]);
    w->coll_arena_gen++; /* Arena::remove (arena.rs:353-380): the generation counts removals, the slot heads the free list */
    free_push(&w->coll_free, &w->ncoll_free, &w->cap_coll_free, collider);
    w->dead_pairs = 1;
    return 0;
}
/* RigidBodySet::remove with remove_attached_colliders = true (rigid_body_set.rs:121-170): attached
 * colliders and joints go too; the arena slot is kept as an inert (fixed, collider-less) body. */
int32_t ro_remove_body(ro_world *w, int32_t body) {
    if (body < 0 || body >= w->nbodies) return -1;
    Body *b = &w->bodies[body];
    for (int k = 0; k < w->nbody_free; ++k) if (w->body_free[k] == body) return -1; /* a free slot: the handle is stale */
    /* the attached colliders go in attachment order (rigid_body_set.rs:140-150 walks rb.colliders()): that is the order of the free list */
    for (int ord = 0, left = 1; left; ++ord) {
        left = 0;
        for (int i = 0; i < w->ncolliders; ++i) {
            const Collider *c = &w->colliders[i];
            if (c->parent != body || (c->memberships == 0 && c->filter == 0)) continue;
            if (c->ord == ord) ro_remove_collider(w, i); else if (c->ord > ord) left = 1;
        }
    }
    for (int i = 0; i < w->njoints; ++i) if (!w->joints[i].removed && (w->joints[i].body1 == body || w->joints[i].body2 == body)) ro_remove_joint(w, i);
    pi_remove_body(w, body); /* IslandManager::rigid_body_removed_or_disabled (manager.rs:62-78) */
    b->body_type = RO_BODY_FIXED;
    b->linvel = V3(0, 0, 0); b->angvel = V3(0, 0, 0);
    b->solver_id = RO_NO_BODY;
    update_world_mass_properties(b);
    w->body_arena_gen++;
    free_push(&w->body_free, &w->nbody_free, &w->cap_body_free, body);
    return 0;
}
int32_t ro_num_joints(const ro_world *w) { return w->njoints; }
/* per joint: (colour, impulse x, y, z) */
void ro_read_joints(const ro_world *w, int32_t *color, float *impulses3) {
    for (int i = 0; i < w->njoints; ++i) {
        if (color) color[i] = w->joints[i].solver_color;
        if (impulses3) for (int k = 0; k < 3; ++k) impulses3[3 * i + k] = w->joints[i].impulses[k];
    }
}

/* ---- hooks for tests/test_convex_oracle.py: the support-mapped queries of ro_convex.h on two shapes given like collider descriptors
 * (shape, half_extents) and the pose of shape 2 in the frame of shape 1 (translation xyz, rotation xyzw) ---- */
static SmShape kat_shape(int32_t shape, const float he[3]) {
    SmShape s; s.shape = shape; s.he = V3(he[0], he[1], he[2]); s.radius = he[0]; s.axis = 1; s.poly = NULL; s.border = 0.0f;
    if (shape == RO_SHAPE_CAPSULE) { s.radius = he[1]; s.axis = (int)he[2]; }
    if (shape == RO_SHAPE_CYLINDER || shape == RO_SHAPE_CONE) { s.radius = he[1]; s.he = V3(he[1], he[0], he[1]); }
    return s;
}
static pose kat_pose(const float p[7]) { pose r; r.t = V3(p[0], p[1], p[2]); r.r = qnormalize(Q(p[3], p[4], p[5], p[6])); return r; }
/* out = hit, p1 xyz, p2 xyz (frame 1), normal xyz (1 -> 2) */
void ro_kat_convex_contact(int32_t sh1, const float he1[3], int32_t sh2, const float he2[3], const float pos12[7], float prediction, float out[10]) {
    SmShape a = kat_shape(sh1, he1), b = kat_shape(sh2, he2);
    v3 p1 = V3(0, 0, 0), p2 = V3(0, 0, 0), n = V3(0, 0, 0);
    int hit = sm_contact(&a, &b, kat_pose(pos12), prediction, V3(0, 0, 0), &p1, &p2, &n);
    out[0] = (float)hit; out[1] = p1.x; out[2] = p1.y; out[3] = p1.z; out[4] = p2.x; out[5] = p2.y; out[6] = p2.z; out[7] = n.x; out[8] = n.y; out[9] = n.z;
}
/* the whole generator on an empty manifold: returns the number of points; pts = up to 8 x (local_p1 xyz, local_p2 xyz, dist, fid1, fid2), n1 = local_n1 */
int32_t ro_kat_convex_manifold(int32_t sh1, const float he1[3], int32_t sh2, const float he2[3], const float pos12[7], float prediction, float *pts, float n1[3]) {
    SmShape a = kat_shape(sh1, he1), b = kat_shape(sh2, he2);
    Manifold m; memset(&m, 0, sizeof(m));
    manifold_pfm_pfm(kat_pose(pos12), &a, &b, prediction, &m);
    for (int i = 0; i < m.npoints; ++i) {
        float *o = pts + 9 * i; const TrackedContact *c = &m.points[i];
        o[0] = c->local_p1.x; o[1] = c->local_p1.y; o[2] = c->local_p1.z; o[3] = c->local_p2.x; o[4] = c->local_p2.y; o[5] = c->local_p2.z;
        o[6] = c->dist; o[7] = (float)c->fid1; o[8] = (float)c->fid2;
    }
    n1[0] = m.local_n1.x; n1[1] = m.local_n1.y; n1[2] = m.local_n1.z;
    return m.npoints;
}
/* the projection of a point on a cylinder / cone: out = proj xyz, inside */
void ro_kat_convex_project(int32_t sh, const float he[3], const float pt[3], float out[4]) {
    SmShape a = kat_shape(sh, he); int inside;
    v3 q = sm_project_point(&a, V3(pt[0], pt[1], pt[2]), &inside);
    out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = (float)inside;
}

/* ---- convex polyhedra (ro_polyhedron.h) ---- */
int32_t ro_add_convex_polyhedron(ro_world *w, int32_t n_points, const float *points_xyz, int32_t n_triangles, const uint32_t *indices) {
    RoPolyhedron *P = (RoPolyhedron *)malloc(sizeof(RoPolyhedron));
    if (ro_poly_build(P, n_points, points_xyz, n_triangles, indices) != 0) { free(P); return -1; }
    w->polys = (RoPolyhedron **)realloc(w->polys, sizeof(RoPolyhedron *) * (w->npolys + 1));
    w->polys[w->npolys] = P;
    return w->npolys++;
}
void ro_read_convex_polyhedron(const ro_world *w, int32_t id, int32_t counts[4], float *points_xyz, float *face_normals, int32_t *face_first, int32_t *face_count,
                               int32_t *loop_vertex, int32_t *loop_edge, float props[20]) {
    const RoPolyhedron *P = w->polys[id];
    counts[0] = P->nv; counts[1] = P->nf; counts[2] = P->nloop; counts[3] = P->ne;
    if (points_xyz) for (int i = 0; i < P->nv; ++i) { points_xyz[3 * i] = P->pts[i].x; points_xyz[3 * i + 1] = P->pts[i].y; points_xyz[3 * i + 2] = P->pts[i].z; }
    if (face_normals) for (int i = 0; i < P->nf; ++i) { face_normals[3 * i] = P->fnormal[i].x; face_normals[3 * i + 1] = P->fnormal[i].y; face_normals[3 * i + 2] = P->fnormal[i].z; }
    if (face_first) for (int i = 0; i < P->nf; ++i) { face_first[i] = P->ffirst[i]; face_count[i] = P->fcount[i]; }
    if (loop_vertex) for (int i = 0; i < P->nloop; ++i) { loop_vertex[i] = P->loop_v[i]; loop_edge[i] = P->loop_e[i]; }
    if (props) {
        float *o = props;
        o[0] = P->centre.x; o[1] = P->centre.y; o[2] = P->centre.z; o[3] = P->half.x; o[4] = P->half.y; o[5] = P->half.z; o[6] = P->origin_radius;
        o[7] = P->sphere_centre.x; o[8] = P->sphere_centre.y; o[9] = P->sphere_centre.z; o[10] = P->sphere_radius;
        o[11] = P->volume; o[12] = P->com.x; o[13] = P->com.y; o[14] = P->com.z;
        o[15] = P->inertia[0][0]; o[16] = P->inertia[1][1]; o[17] = P->inertia[2][2]; o[18] = P->inertia[0][1]; o[19] = P->inertia[0][2];
    }
}

static void comp_free_fwd(struct RoComposite *C) { comp_free(C); }
/* solver manifolds of pair (c1, c2): returns the number of clusters (0 = the pair takes the plain path), -1 = no such pair;
 * nsc_out[k] = solver contacts of solver manifold k (k = 0 is also filled on the plain path) */
int32_t ro_pair_clusters(const ro_world *w, int32_t c1, int32_t c2, int32_t cap, int32_t *nsc_out) {
    const int64_t key = ((int64_t)(c1 < c2 ? c1 : c2) << 32) | (uint32_t)(c1 < c2 ? c2 : c1);
    const int pi = map_find(w, key);
    if (pi < 0) return -1;
    Pair *p = (Pair *)&w->pairs[pi];
    const int n = p->ncl > 0 ? p->ncl : 1;
    for (int k = 0; k < n && k < cap; ++k) nsc_out[k] = *sm_nsc(p, k);
    return p->ncl;
}

/* debug aid (tools/composite_diag.py): the same record as the device's rp_debug_pair_points */
int32_t ro_debug_pair_points(const ro_world *w, int32_t c1, int32_t c2, int32_t cap, float *out) {
    const int64_t key = ((int64_t)c1 << 32) | (uint32_t)c2;
    const int pi = map_find(w, key);
    if (pi < 0) return 0;
    Pair *p = (Pair *)&w->pairs[pi];
    int n = 0;
#define RO_PUT(v) do { if (n < cap) out[n] = (float)(v); ++n; } while (0)
    RO_PUT(p->ncl); RO_PUT(p->plain_sub[0]); RO_PUT(p->plain_sub[1]);
    for (int k = 0; k < pair_num_sm(p); ++k) {
        const Manifold *m = sm_m(p, k);
        RO_PUT(k); RO_PUT(m->npoints); RO_PUT(*sm_nsc(p, k));
        for (int i = 0; i < m->npoints; ++i) { RO_PUT(m->points[i].local_p1.x); RO_PUT(m->points[i].local_p1.y); RO_PUT(m->points[i].local_p1.z); RO_PUT(m->points[i].dist); RO_PUT(m->points[i].data.impulse); RO_PUT(m->points[i].data.warmstart_impulse); }
    }
#undef RO_PUT
    return n;
}
