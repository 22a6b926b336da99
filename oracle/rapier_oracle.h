/*
 * oracle/rapier_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Public (ctypes-friendly) interface of the CPU oracle: a single-threaded, scalar C
 * restatement of rapier3d/f32 `PhysicsPipeline::step()` (reference v0.35.2,
 * /root/reference/src/pipeline/physics_pipeline/substep.rs:267-581) restricted to the
 * hot-path scope of SURVEY.md §8: fat-AABB broad phase pair set, cuboid/ball contact
 * manifolds, persistent greedy pair colouring, colour-ordered TGS-soft contact solver with
 * twist friction, spherical/fixed impulse joints, linearised integrator.
 *
 * PARITY PINNING: the Rust reference cannot be built in this environment (no cargo) and
 * parry3d's manifold generator is not in /root/reference.  The oracle is therefore pinned
 * by the reference's own outcome-level known-answer tests (tests/test_oracle_kat.py):
 * total_contact_impulse.rs:13-75, broad_phase_bvh/mod.rs:281-329, test_staged.rs:86-148,
 * coefficient_combine_rule.rs:60-96.  Manifold-level (parry-internal) parity is UNPINNED.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#ifndef RAPIER_ORACLE_H
#define RAPIER_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* IntegrationParameters — /root/reference/src/dynamics/integration_parameters.rs:181-304,
 * defaults :379-408. */
typedef struct ro_params {
    float dt;
    float contact_natural_frequency, contact_damping_ratio;               /* 30, 10 */
    float static_contact_natural_frequency, static_contact_damping_ratio; /* 60, 10 */
    float joint_natural_frequency, joint_damping_ratio;                   /* 1e6, 1 */
    float warmstart_coefficient;                                          /* 1 */
    float normalized_allowed_linear_error;                                /* 0.005 */
    float normalized_max_corrective_velocity;                             /* 3 */
    float normalized_prediction_distance;                                 /* 0.02 */
    float normalized_max_linear_velocity;                                 /* 400 */
    float normalized_contact_recycle_distance;                            /* 0.05 */
    float length_unit;                                                    /* 1 */
    int32_t num_solver_iterations;                                        /* 4 (substeps) */
    int32_t num_internal_pgs_iterations;                                  /* 1 */
    int32_t num_internal_stabilization_iterations;                        /* 1 */
    int32_t contact_recycling;                                            /* 1 */
    int32_t friction_in_bias_pass;                                        /* 0 */
    int32_t warmstart_joints;                                             /* 0 */
    int32_t max_ccd_substeps;                                             /* 1 (flag only) */
    int32_t friction_model;                                               /* FrictionModel: 0 Simplified (twist), 1 Coulomb */
    float min_ccd_dt;                                                     /* dt / 100; read only by the step splitting of max_ccd_substeps > 1 (substep.rs:410-470), which this oracle does not run */
    int32_t contact_clustering;                                           /* 1; composite pairs are always clustered here */
} ro_params;

/* RigidBodyType — rigid_body_components.rs */
enum { RO_BODY_DYNAMIC = 0, RO_BODY_FIXED = 1, RO_BODY_KINEMATIC_POSITION = 2, RO_BODY_KINEMATIC_VELOCITY = 3 };
enum { RO_FRICTION_SIMPLIFIED = 0, RO_FRICTION_COULOMB = 1 }; /* integration_parameters.rs:13-32 */
enum { RO_SHAPE_BALL = 0, RO_SHAPE_CUBOID = 1, RO_SHAPE_CAPSULE = 2 /* half_extents = (half_height, radius, axis 0|1|2): ColliderBuilder::capsule_x/y/z */,
       RO_SHAPE_HALFSPACE = 3 /* half_extents = the unit outward normal in the collider's frame: ColliderBuilder::halfspace */,
       RO_SHAPE_CYLINDER = 4 /* half_extents = (half_height, radius, -): ColliderBuilder::cylinder (collider.rs:770), axis Y */,
       RO_SHAPE_CONE = 5 /* half_extents = (half_height, radius, -): ColliderBuilder::cone (collider.rs:789), apex at +Y */,
       RO_SHAPE_CONVEX_POLYHEDRON = 6 /* half_extents[0] = the id ro_add_convex_polyhedron returned: ColliderBuilder::convex_mesh / convex_hull (collider.rs:1039, :1070) */,
       /* ColliderBuilder::round_cuboid / round_cylinder / round_cone / round_convex_hull (collider.rs:778-1080): the shape above dilated by a
        * sphere of radius ro_collider_desc.border_radius (parry RoundShape<S>); half_extents as for the inner shape */
       RO_SHAPE_ROUND_CUBOID = 7, RO_SHAPE_ROUND_CYLINDER = 8, RO_SHAPE_ROUND_CONE = 9, RO_SHAPE_ROUND_CONVEX_POLYHEDRON = 10,
       /* composite shapes as ONE collider (half_extents[0] = the id ro_add_compound / ro_add_trimesh / ro_add_heightfield returned):
        * ColliderBuilder::compound (collider.rs:711), ::trimesh (:944), ::heightfield (:1089).  Several manifolds per pair ->
        * contact clustering (contact_clustering.rs:33).  A triangle mesh / height field needs a fixed or kinematic parent (or none). */
       RO_SHAPE_COMPOUND = 11, RO_SHAPE_TRIMESH = 12,
       RO_SHAPE_TRIANGLE = 13 /* internal: one triangle of a mesh as a support-mapped shape (never a collider's own shape) */ };
/* CoefficientCombineRule — coefficient_combine_rule.rs:37-57 */
enum { RO_RULE_AVERAGE = 0, RO_RULE_MIN = 1, RO_RULE_MULTIPLY = 2, RO_RULE_MAX = 3,
       RO_RULE_CLAMPED_SUM = 4, RO_RULE_GEOMETRIC_MEAN = 5 };

typedef struct ro_body_desc {
    int32_t body_type;
    float translation[3];
    float rotation[4]; /* x,y,z,w */
    float linvel[3], angvel[3];
    float linear_damping, angular_damping;
    float gravity_scale;
    float additional_mass;  /* RigidBodyBuilder::additional_mass */
    int32_t dominance;      /* i8 group */
    int32_t gyroscopic;     /* default 1 — rigid_body.rs:1579 */
    int32_t allow_fast_rotation;
    int32_t can_sleep;      /* RigidBodyBuilder::can_sleep — RigidBodyActivation::active() vs cannot_sleep() */
    uint32_t locked_axes;   /* LockedAxes: bit0..2 TRANSLATION_LOCKED_X/Y/Z, bit3..5 ROTATION_LOCKED_X/Y/Z */
} ro_body_desc;

typedef struct ro_collider_desc {
    int32_t shape;
    float half_extents[3]; /* cuboid; ball: radius in [0] */
    float translation[3];  /* pos_wrt_parent (or world pose when parent < 0) */
    float rotation[4];
    float density, friction, restitution;
    int32_t friction_rule, restitution_rule;
    uint32_t collision_memberships, collision_filter; /* InteractionGroups */
    uint32_t active_events;                /* ActiveEvents: bit0 COLLISION_EVENTS, bit1 CONTACT_FORCE_EVENTS */
    float contact_force_event_threshold;   /* ColliderBuilder::contact_force_event_threshold */
    int32_t sensor;                        /* (set through ro_set_collider_sensor; the field keeps the product's layout) */
    float border_radius;                   /* round shapes: RoundShape::border_radius */
} ro_collider_desc;

/* JointMotor (dynamics/joint/generic_joint.rs:200-232); model: 0 = MotorModel::AccelerationBased, 1 = ForceBased */
typedef struct ro_joint_motor { float target_vel, target_pos, stiffness, damping, max_force; int32_t model; } ro_joint_motor;

/* GenericJoint: locked axes, limits and motors of the free axes, coupled axes (RopeJoint / SpringJoint couple the linear axes). */
typedef struct ro_joint_desc {
    uint64_t body1, body2; /* the layout of rp_joint_desc (RigidBodyHandles there); the oracle has no generations: the low 32 bits are the body index */
    float local_anchor1[3], local_anchor2[3];
    float local_basis1[4], local_basis2[4];
    uint32_t locked_axes; /* bit0..2 lin x,y,z ; bit3..5 ang x,y,z */
    int32_t contacts_enabled;
    uint32_t limit_axes;  /* JointAxesMask of the limited (free) axes — GenericJoint::limit_axes */
    float limits[6][2];   /* JointLimits::{min, max} per axis (metres for the linear axes, radians for the angular ones) */
    uint32_t motor_axes;  /* JointAxesMask of the motorised (free) axes — GenericJoint::motor_axes */
    ro_joint_motor motors[6];
    uint32_t coupled_axes; /* GenericJoint::coupled_axes (generic_joint.rs:285): the linear axes in it share ONE limit / motor row along their
                            * combined error, exactly two angular axes in it one limit row; the first coupled axis carries limits and motor */
    uint32_t reserved;
} ro_joint_desc;

#define RO_ISLAND_STATS_MAX 16
typedef struct ro_world ro_world;

void ro_default_params(ro_params *out);
ro_world *ro_world_new(const ro_params *params, const float gravity[3]);
/* the reference takes its IntegrationParameters per step (PhysicsPipeline::step(.., integration_parameters, ..)): they may change between any two steps */
void ro_set_params(ro_world *w, const ro_params *params);
void ro_world_free(ro_world *w);
int32_t ro_add_body(ro_world *w, const ro_body_desc *d);
int32_t ro_add_collider(ro_world *w, const ro_collider_desc *d, int32_t parent_body);
/* twin of rp_world_begin_subworld (include/rapier_hip.h): what is added from now on belongs to a new sub-world; returns its index */
int32_t ro_begin_subworld(ro_world *w);
/* SharedShape::convex_mesh(points, indices): registers a convex polyhedron (closed, outward-wound triangle mesh); returns its id or -1 */
int32_t ro_add_convex_polyhedron(ro_world *w, int32_t n_points, const float *points_xyz, int32_t n_triangles, const uint32_t *indices);
/* Composite shapes (registered once, shared by colliders; -1 = invalid input).  Compound: `parts` are collider descriptors of which only
 * shape (a primitive or round primitive), half_extents, translation, rotation and border_radius are read (SharedShape::compound).
 * TriMesh: vertices + triangle index triples (TriMesh::new, no flags).  HeightField: an nrows x ncols grid of heights (row-major
 * heights[r * ncols + c], r along z, c along x) scaled by `scale` — served by the triangle-mesh machinery with parry's cell
 * triangulation (two triangles per cell, heightfield3.rs). */
int32_t ro_add_compound(ro_world *w, int32_t n_parts, const struct ro_collider_desc *parts);
int32_t ro_add_trimesh(ro_world *w, int32_t n_vertices, const float *vertices_xyz, int32_t n_triangles, const uint32_t *indices);
int32_t ro_add_heightfield(ro_world *w, int32_t nrows, int32_t ncols, const float *heights, const float scale[3]);
/* solver manifolds (clusters) of pair (c1, c2): count, and per cluster the solver-contact count (cap entries) — test / diagnostics */
int32_t ro_pair_clusters(const ro_world *w, int32_t c1, int32_t c2, int32_t cap, int32_t *nsc_out);
/* the canonical form it was given (ro_polyhedron.h): counts = {vertices, faces, loop entries, edges}; then the arrays (NULL = skip) */
void ro_read_convex_polyhedron(const ro_world *w, int32_t id, int32_t counts[4], float *points_xyz, float *face_normals, int32_t *face_first, int32_t *face_count, int32_t *loop_vertex, int32_t *loop_edge, float props[20]);
int32_t ro_add_joint(ro_world *w, const ro_joint_desc *d);
int32_t ro_remove_body(ro_world *w, int32_t body);
/* Index::generation of the occupant of an arena slot (data/arena.rs:58-90): the arena's removal count when it was inserted */
uint32_t ro_body_generation(const ro_world *w, int32_t body);
uint32_t ro_collider_generation(const ro_world *w, int32_t collider);
int32_t ro_remove_collider(ro_world *w, int32_t collider);
int32_t ro_remove_joint(ro_world *w, int32_t joint);
/* GenericJoint::set_motor through ImpulseJointSet::get_mut(handle, wake_up = true): enables the axis' motor */
int32_t ro_set_joint_motor(ro_world *w, int32_t joint, int32_t axis, const ro_joint_motor *m);
/* JointMotor::impulse of the six axes of every joint */
void ro_read_joint_motor_impulses(const ro_world *w, float *impulses6);
/* ColliderBuilder::sensor / NarrowPhase::intersection_pair — oracle only so far: sensor pairs generate no
 * contacts, only Started / Stopped collision events flagged CollisionEventFlags::SENSOR (= 1) */
void ro_set_collider_sensor(ro_world *w, int32_t collider, int32_t on);
int32_t ro_intersection_pair(const ro_world *w, int32_t c1, int32_t c2); /* 1 / 0 = intersecting or not, -1 = no such sensor pair */
/* RigidBody::additional_solver_iterations: the body's whole connected component runs that many extra substeps (substep solve-groups,
 * island_manager/substep_groups.rs).  Oracle only so far: the device ABI does not expose it yet. */
void ro_set_additional_solver_iterations(ro_world *w, int32_t body, int32_t n);
void ro_read_solve_group_extras(const ro_world *w, int32_t *out); /* per body: extra substeps of its solve group in the last step, -1 outside the active set */
/* persistent island id of every body (IslandManager::persistent_island_of, manager.rs:214-220): -1 for fixed / removed bodies.
 * Only equality between two bodies' ids is meaningful in the reference; here the ids are also canonical (see the island section
 * of rapier_oracle.c), so the device's ids can be compared one to one. */
void ro_read_island_labels(ro_world *w, int32_t *out);
/* Counters of the persistent-island machinery since world creation (diagnostics for the parity argument in DESIGN.md section 5):
 * which of the decisions that the reference takes in contact-graph edge order — an order owned by parry's BVH traversal — were
 * actually exercised by a scene. */
enum {
    RO_IS_MERGED = 0,            /* islands absorbed by merge_islands */
    RO_IS_MULTIWAY_GROUPS,       /* merge groups of > 2 islands in one step (the surviving identity may depend on the merge order) */
    RO_IS_REMOVALS,              /* journal entries resolved */
    RO_IS_CONNECTED,             /* local search verdict: still connected */
    RO_IS_DETACHED,              /* local search verdict: detached, component moved out */
    RO_IS_HOT,                   /* both endpoints above the sleep speed: deferred to the global split */
    RO_IS_OVER_BUDGET,           /* SEARCH_BUDGET exceeded */
    RO_IS_SLEEPING_DEFERRED,     /* removal inside a sleeping island */
    RO_IS_GLOBAL_SPLITS,         /* split_island_now runs */
    RO_IS_GLOBAL_SPLIT_PIECES,   /* islands created by them */
    RO_IS_BIDS,                  /* steps with a split bid */
    RO_IS_BID_TIES,              /* ... whose winning score was shared by two or more islands (island-id tie-break) */
    RO_IS_SLEEP_BLOCKED,         /* (island, step): every body eligible, sleep refused by the constraint_remove_count gate */
    RO_IS_ORDER_DEPENDENT,       /* steps in which two detaching removals hit one island (journal order can matter) */
    RO_IS_DETACH_SIZE_TIES,      /* detach between components of equal size (side 0 = body1's side wins) */
    RO_IS_SPLIT_KEEP_TIES,       /* global split whose largest component was not unique */
    RO_ISLAND_STATS
};
void ro_read_island_stats(const ro_world *w, int32_t *out /* RO_ISLAND_STATS */);
/* PersistentIsland of an id: (in use, bodies.len(), constraint_remove_count > 0, split_denied_until, sleeping); PersistentIslands:
 * (sleep_scan_stamp, split_island or -1) */
void ro_read_island_state(const ro_world *w, int32_t island, int32_t out5[5]);
void ro_read_island_globals(const ro_world *w, int32_t out2[2]);
/* the step each body last fell asleep at (0 = never) and the step of each pair's last full narrow-phase update: per-step trace
 * material for bisecting against bench/rapier_ref --dump */
void ro_read_slept_at(const ro_world *w, int32_t *out);
/* RigidBodyBuilder::ccd_enabled (the body sweeps kinematic / dynamic targets too: a "bullet", dynamics/ccd/sweeps.rs:29-41) and the
 * counters of the continuous pass: (body, step) cases of the fast-body criterion, clamped next_positions */
void ro_set_ccd_enabled(ro_world *w, int32_t body, int32_t on);
void ro_read_ccd_counts(const ro_world *w, int32_t out2[2]);
int32_t ro_num_joints(const ro_world *w);
void ro_read_joints(const ro_world *w, int32_t *color, float *impulses3);
void ro_step(ro_world *w, int32_t nsteps);
/* OpenMP threads used by the data-parallel loops (default 1; results do not depend on it). */
void ro_set_threads(int32_t n);
int32_t ro_get_threads(void);
int32_t ro_num_bodies(const ro_world *w);
/* pos7 = (tx,ty,tz, qx,qy,qz,qw) per body, vel6 = (lin, ang) per body, arena order. */
void ro_read_bodies(const ro_world *w, float *pos7, float *vel6);
/* RigidBody::set_linvel/set_angvel(.., wake_up = true) */
void ro_set_body_vel(ro_world *w, int32_t body, const float linvel[3], const float angvel[3]);
/* CollisionEvent (geometry/mod.rs:105-140) and ContactForceEvent (:180-258) raised since the last drain, in emission
 * order.  Collision: 5 ints (collider1, collider2, started, CollisionEventFlags, step).  Force: 4 ints (collider1,
 * collider2, step, started) + 8 floats (total_force xyz, total_force_magnitude, max_force_direction xyz,
 * max_force_magnitude).  Return the number of pending events (only `cap` are written); the lists are cleared. */
int32_t ro_collision_events_drain(ro_world *w, int32_t cap, int32_t *out5);
int32_t ro_force_events_drain(ro_world *w, int32_t cap, int32_t *meta4, float *vals8);
/* RigidBody::{reset_forces, reset_torques} (when `reset`), then add_force / add_torque (.., wake_up = true), rigid_body.rs:1145-1252:
 * the user force persists across steps until reset.  NULL vectors are skipped; dynamic bodies only. */
void ro_add_force(ro_world *w, int32_t body, const float force[3], const float torque[3], int32_t reset);
/* RigidBody::{apply_impulse, apply_torque_impulse} (.., wake_up = true), rigid_body.rs:1304-1343 */
void ro_apply_impulse(ro_world *w, int32_t body, const float impulse[3], const float torque_impulse[3]);
/* RigidBody::set_next_kinematic_position (rigid_body.rs:1085-1093): kinematic bodies only; wakes when the pose differs */
void ro_set_next_kinematic_position(ro_world *w, int32_t body, const float pos7[7]);
/* RigidBody::set_position(.., wake_up = true) */
void ro_set_body_pose(ro_world *w, int32_t body, const float pos7[7]);
/* RigidBodyMassProps::local_mprops of a body: inv_mass, local_com, inv_principal_inertia, principal frame (x,y,z,w) */
void ro_body_mass_props(const ro_world *w, int32_t body, float out11[11]);
/* IslandManager::wake_up (island_manager/sleep.rs:31): wakes the body's whole island. */
void ro_wake_up(ro_world *w, int32_t body, int32_t strong);
/* RigidBody::is_sleeping per body (arena order). */
void ro_read_sleeping(const ro_world *w, int32_t *sleeping);

/* Statistics of the last step (for DESIGN/bench: M and colour histogram). */
typedef struct ro_stats {
    int32_t num_pairs;          /* broad-phase pairs in the pair table */
    int32_t num_active_manifolds; /* M: solver manifolds */
    int32_t num_solver_contacts;
    int32_t num_colors_used;
    int32_t num_parallel_colors; /* colours with >= 32 four-lane chunks */
    int32_t num_full_updates;   /* pairs that took the full narrow-phase path */
    int32_t num_recycled;
    int32_t bp_rebuilt;
} ro_stats;
void ro_get_stats(const ro_world *w, ro_stats *out);
/* Sum over contact pairs of |sum_k impulse_k * normal| — ContactPair::total_impulse_magnitude. */
float ro_total_contact_impulse(const ro_world *w);
/* Dump active manifolds: per manifold (c1,c2,color,count) and per point impulse.  Returns M. */
int32_t ro_dump_manifolds(const ro_world *w, int32_t cap, int32_t *c1c2_color_count,
                          float *normal3, float *impulses4);
float ro_combine_coefficient(float a, float b, int32_t rule_a, int32_t rule_b);

#ifdef __cplusplus
}
#endif
#endif
