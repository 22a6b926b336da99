/*
 * rapier_hip.h — C ABI of librapier_hip.so, the MI355X-native drop-in for the rapier3d (f32)
 * `PhysicsPipeline::step()` hot path.
 *
 * The reference has no FFI around this path (everything is Rust); the seam this
 * ABI replaces is B1/B2 of SURVEY §8(b):
 *   PhysicsPipeline::step      /root/reference/src/pipeline/physics_pipeline/mod.rs:196-246
 *   PhysicsWorld (owner of the ten sets)   /root/reference/src/pipeline/physics_world.rs:61-157
 *   RigidBodySet::insert       /root/reference/src/dynamics/rigid_body_set.rs:70
 *   ColliderSet::insert_with_parent  /root/reference/src/geometry/collider_set.rs:49
 *   ImpulseJointSet::insert    /root/reference/src/dynamics/joint/impulse_joint/impulse_joint_set.rs
 *   Index{index,generation}    /root/reference/src/data/arena.rs:58-90  (handles)
 *   Counters                   /root/reference/src/counters/mod.rs:18
 * A Rust shim binding these gives RigidBodySet/ColliderSet/PhysicsPipeline
 * newtypes with the reference's method names.
 *
 * Conventions: C linkage, no exceptions cross the boundary, every call returns int32 status
 * (0 = ok, <0 = error; text via rp_last_error).  The opaque rp_world owns all host and device
 * memory; the caller owns every array it passes.  One thread per world at a time; different
 * worlds may be stepped concurrently on different devices.  All state lives in HBM between
 * calls; rp_step enqueues on the world's HIP stream and returns without host sync unless
 * stated.  There is NO CPU fallback: if no HIP device is usable rp_world_create fails.
 */
#ifndef RAPIER_HIP_H
#define RAPIER_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RP_OK 0
#define RP_ERR_INVALID (-1)
#define RP_ERR_DEVICE (-2)
#define RP_ERR_CAPACITY (-3)
#define RP_ERR_NONFINITE (-4)

/* IntegrationParameters — integration_parameters.rs:181-304 (defaults :379-408): every field of the 3-D build, the two
 * SpringCoefficients (contact_softness, static_contact_softness: natural_frequency + damping_ratio each) spelt out, bools and usizes
 * as int32.  joint_natural_frequency / joint_damping_ratio are the JointSoftness constants of joint_constraint_helper.rs (1e6 Hz, 1). */
typedef struct rp_integration_params {
    float dt;
    float contact_natural_frequency, contact_damping_ratio;
    float static_contact_natural_frequency, static_contact_damping_ratio;
    float joint_natural_frequency, joint_damping_ratio;
    float warmstart_coefficient;
    float normalized_allowed_linear_error;
    float normalized_max_corrective_velocity;
    float normalized_prediction_distance;
    float normalized_max_linear_velocity;
    float normalized_contact_recycle_distance;
    float length_unit;
    int32_t num_solver_iterations;
    int32_t num_internal_pgs_iterations;
    int32_t num_internal_stabilization_iterations;
    int32_t contact_recycling;
    int32_t friction_in_bias_pass;
    int32_t warmstart_joints;
    int32_t max_ccd_substeps;
    int32_t friction_model; /* FrictionModel (integration_parameters.rs:13-32): RP_FRICTION_SIMPLIFIED (default) | RP_FRICTION_COULOMB */
    float min_ccd_dt;       /* integration_parameters.rs:200 (default dt / 100): the shortest CCD substep, read only by the step splitting of
                             * max_ccd_substeps > 1 (substep.rs:410-470).  This library runs every step as ONE CCD substep (max_ccd_substeps
                             * is 0 = off or >= 1 = on), so the value is validated (finite, >= 0), stored and handed back, and has no effect */
    int32_t contact_clustering; /* integration_parameters.rs:278 (default true = 1): pairs that hold several manifolds (composite shapes) are
                                 * solved as clusters (pair_update.rs:350).  0 is accepted for worlds without composite shapes, where it
                                 * changes nothing (a primitive pair holds one manifold); a world that holds a compound / mesh / height
                                 * field refuses 0 with RP_ERR_INVALID (its unclustered form is not built) */
} rp_integration_params;

enum { RP_FRICTION_SIMPLIFIED = 0, RP_FRICTION_COULOMB = 1 };

/* RigidBodyType — rigid_body_components.rs */
enum { RP_BODY_DYNAMIC = 0, RP_BODY_FIXED = 1, RP_BODY_KINEMATIC_POSITION = 2, RP_BODY_KINEMATIC_VELOCITY = 3 };
enum { RP_SHAPE_BALL = 0, RP_SHAPE_CUBOID = 1,
       RP_SHAPE_CAPSULE = 2 /* ColliderBuilder::capsule_x/y/z (collider.rs): half_extents = (half_height, radius, axis 0|1|2) */,
       RP_SHAPE_HALFSPACE = 3 /* ColliderBuilder::halfspace(outward_normal): half_extents = the unit normal in the collider's frame; the
                                 solid side is dot(normal, p) <= 0.  Parent: none, a fixed or a kinematic body (an unbounded shape
                                 on a dynamic body is refused); weighs nothing (MassProperties::zero) */,
       RP_SHAPE_CYLINDER = 4 /* ColliderBuilder::cylinder(half_height, radius) (collider.rs:770): half_extents = (half_height, radius, -),
                                axis Y */,
       RP_SHAPE_CONE = 5 /* ColliderBuilder::cone(half_height, radius) (collider.rs:789): half_extents = (half_height, radius, -), base at
                            -half_height, apex at +half_height; its centre of mass sits a quarter of the height above the base */,
       RP_SHAPE_CONVEX_POLYHEDRON = 6 /* ColliderBuilder::convex_mesh / convex_hull (collider.rs:1039, :1070): half_extents[0] = the id
                                         rp_convex_polyhedron_create returned (a whole number stored as a float) */,
       /* ColliderBuilder::round_cuboid / round_cylinder / round_cone / round_convex_hull / round_convex_mesh (collider.rs:700-1090): the
        * shape above dilated by a sphere of radius rp_collider_desc.border_radius (parry RoundShape<S>); half_extents as for the inner
        * shape; mass properties are the inner shape's (RoundShape::mass_properties) */
       RP_SHAPE_ROUND_CUBOID = 7, RP_SHAPE_ROUND_CYLINDER = 8, RP_SHAPE_ROUND_CONE = 9, RP_SHAPE_ROUND_CONVEX_POLYHEDRON = 10,
       /* Composite shapes as ONE collider — ColliderBuilder::compound (collider.rs:711), ::trimesh (:944), ::heightfield (:1089):
        * half_extents[0] = the id rp_compound_create / rp_trimesh_create / rp_heightfield_create returned.  A pair with such a collider
        * holds one manifold per sub-shape pair in reach; several manifolds are merged into solver clusters by normal
        * (contact_clustering.rs:33), whose warm-start data is carried by position (:129).  A triangle mesh / height field needs a fixed
        * or kinematic parent (or none) and weighs nothing; a compound's mass properties are the sum of its parts'. */
       RP_SHAPE_COMPOUND = 11, RP_SHAPE_TRIMESH = 12,
       RP_SHAPE_TRIANGLE = 13 /* internal to the shape dispatchers: one triangle of a mesh; never a collider's shape */ };
enum { RP_RULE_AVERAGE = 0, RP_RULE_MIN, RP_RULE_MULTIPLY, RP_RULE_MAX, RP_RULE_CLAMPED_SUM, RP_RULE_GEOMETRIC_MEAN };

/* RigidBodyBuilder — /root/reference/src/dynamics/rigid_body.rs:1560-1900 */
typedef struct rp_body_desc {
    int32_t body_type;
    float translation[3];
    float rotation[4]; /* unit quaternion x,y,z,w */
    float linvel[3], angvel[3];
    float linear_damping, angular_damping;
    float gravity_scale;
    float additional_mass;
    int32_t dominance;
    int32_t gyroscopic;
    int32_t allow_fast_rotation;
    int32_t can_sleep; /* RigidBodyBuilder::can_sleep (rigid_body.rs:1845): 1 = RigidBodyActivation::active(), 0 = cannot_sleep() */
    uint32_t locked_axes; /* LockedAxes (rigid_body_components.rs:271-288): bit0..2 TRANSLATION_LOCKED_X/Y/Z, bit3..5 ROTATION_LOCKED_X/Y/Z */
    int32_t additional_solver_iterations; /* RigidBodyBuilder::additional_solver_iterations (rigid_body.rs): extra TGS substeps for the body's
                                           * whole connected component (island_manager/substep_groups.rs:44-229); at most 15 distinct
                                           * positive values per world.  Worlds that use it are solved by one workgroup. */
    int32_t ccd_enabled; /* RigidBodyBuilder::ccd_enabled (rigid_body.rs): 1 = a "bullet" whose continuous-collision sweep also meets kinematic
                          * and dynamic bodies (dynamics/ccd/sweeps.rs:22-41).  Every fast dynamic body sweeps the FIXED colliders
                          * whatever this flag says, unless IntegrationParameters::max_ccd_substeps = 0 (ccd_solver.rs:17-25). */
} rp_body_desc;

/* ColliderBuilder — /root/reference/src/geometry/collider.rs:600-1130 */
typedef struct rp_collider_desc {
    int32_t shape;
    float half_extents[3]; /* cuboid half extents; ball radius in [0]; capsule (half_height, radius, axis) */
    float translation[3];  /* pos_wrt_parent (world pose when parent handle is RP_INVALID_HANDLE) */
    float rotation[4];
    float density, friction, restitution;
    int32_t friction_rule, restitution_rule;
    uint32_t collision_memberships, collision_filter;
    uint32_t active_events;              /* ActiveEvents (pipeline/event_handler.rs:11-24): RP_EVENTS_COLLISION | RP_EVENTS_CONTACT_FORCE */
    float contact_force_event_threshold; /* ColliderBuilder::contact_force_event_threshold (collider.rs:1040) */
    int32_t sensor;                      /* ColliderBuilder::sensor(true) (collider.rs): the collider's pairs live in the intersection graph
                                          * (narrow_phase/intersections.rs:17-175): no contacts, no forces, no wake-ups, only Started / Stopped
                                          * collision events flagged RP_COLLISION_EVENT_SENSOR and rp_intersection_pairs_read */
    float border_radius;                 /* RP_SHAPE_ROUND_*: RoundShape::border_radius (> 0); ignored for the other shapes */
} rp_collider_desc;

enum { RP_EVENTS_COLLISION = 1, RP_EVENTS_CONTACT_FORCE = 2 };
enum { RP_COLLISION_EVENT_SENSOR = 1, RP_COLLISION_EVENT_REMOVED = 2 }; /* CollisionEventFlags::{SENSOR, REMOVED} (geometry/mod.rs:95-102) */

/* CollisionEvent::{Started, Stopped}(collider1, collider2, flags) — geometry/mod.rs:105-140 */
typedef struct rp_collision_event {
    int32_t collider1, collider2;
    int32_t started;   /* 1 = Started, 0 = Stopped */
    int32_t flags;     /* CollisionEventFlags */
    int32_t step;      /* 1-based step (since the device world was built) that raised the event */
} rp_collision_event;
/* ContactForceEvent — geometry/mod.rs:180-258 */
typedef struct rp_contact_force_event {
    int32_t collider1, collider2;
    int32_t step;
    int32_t started;   /* the pair was not above its threshold in the previous step */
    float total_force[3];
    float total_force_magnitude;
    float max_force_direction[3];
    float max_force_magnitude;
} rp_contact_force_event;

/* JointMotor — /root/reference/src/dynamics/joint/generic_joint.rs:200-232 (the accumulated impulse is read back with
 * rp_impulse_joints_read_motor_impulses); JointMotor::default(): all zero, max_force = FLT_MAX, AccelerationBased */
#define RP_MOTOR_ACCELERATION_BASED 0 /* MotorModel::AccelerationBased — motor_model.rs:24-31 */
#define RP_MOTOR_FORCE_BASED 1        /* MotorModel::ForceBased */
typedef struct rp_joint_motor {
    float target_vel, target_pos, stiffness, damping, max_force;
    int32_t model;
} rp_joint_motor;

/* GenericJoint — /root/reference/src/dynamics/joint/generic_joint.rs:255-355; the local frames are (local_anchor, local_basis) like
 * GenericJoint::local_frame1/2 */
typedef struct rp_joint_desc {
    uint64_t body1, body2; /* RigidBodyHandles (generation << 32 | index), as ImpulseJointSet::insert(body1, body2, ..) takes them
                            * (impulse_joint_set.rs:329-375); a stale or removed handle is refused with RP_ERR_INVALID */
    float local_anchor1[3], local_anchor2[3];
    float local_basis1[4], local_basis2[4];
    uint32_t locked_axes; /* JointAxesMask: bit0..2 LIN_X,Y,Z ; bit3..5 ANG_X,Y,Z */
    int32_t contacts_enabled;
    uint32_t limit_axes;  /* GenericJoint::limit_axes: JointAxesMask of the limited (free) axes */
    float limits[6][2];   /* JointLimits::{min, max} per axis (GenericJoint::limits, generic_joint.rs:230-245): metres / radians */
    uint32_t motor_axes;  /* GenericJoint::motor_axes: JointAxesMask of the motorised (free) axes */
    rp_joint_motor motors[6]; /* GenericJoint::motors */
    uint32_t coupled_axes;    /* GenericJoint::coupled_axes (generic_joint.rs:285): the linear axes in the mask share ONE limit row (max distance:
                               * RopeJoint, rope_joint.rs:31-38) and ONE motor row (SpringJoint, spring_joint.rs:31-40) along their combined
                               * error — limits and motor are those of the first coupled axis; exactly two angular axes in the mask share one
                               * limit row (joint_constraint_helper.rs:725-790); a motor on coupled angular axes does nothing, as in the reference */
    uint32_t reserved;        /* 0 (keeps the size a multiple of the handles' 8-byte alignment without implicit padding) */
} rp_joint_desc;

/* Counters mirror (ms, from hipEvents) — /root/reference/src/counters/{mod,stages_counters,
 * collision_detection_counters,solver_counters}.rs; plus device-side scene statistics. */
typedef struct rp_counters {
    float step_time_ms;            /* Counters::step_time (last rp_step call, per step) */
    float collision_detection_ms;  /* StagesCounters::collision_detection_time */
    float broad_phase_ms;          /* CollisionDetectionCounters::broad_phase_time (collider poses + pair-set maintenance); timed full steps only */
    float narrow_phase_ms;         /* CollisionDetectionCounters::narrow_phase_time (recycle test, contact determination, colouring); full steps only */
    float island_construction_ms;  /* StagesCounters::island_construction_time (sleep decision, solver-graph buckets, contact islands); full steps only */
    float solver_ms;               /* StagesCounters::solver_time */
    float velocity_assembly_ms;    /* SolverCounters::velocity_assembly_time (0: fused into the solve kernels) */
    float velocity_resolution_ms;  /* SolverCounters::velocity_resolution_time: k_island_solve, the TGS loop of all LDS-resident islands */
    float velocity_update_ms;      /* SolverCounters::velocity_update_time: the global solve path (large islands, free bodies) + writeback */
    int32_t num_pairs;             /* CollisionDetectionCounters::ncontact_pairs */
    int32_t num_manifolds;         /* SolverCounters::nconstraints (solver manifolds M) */
    int32_t num_solver_contacts;   /* SolverCounters::ncontacts */
    int32_t num_colors;            /* colours in use */
    int32_t num_parallel_stages;   /* colours with >= 32 four-lane chunks (init.rs:169) */
    int32_t num_dynamic_bodies;
    int32_t bp_rebuilds;           /* broad-phase pair-set rebuilds so far */
    int32_t full_updates;          /* narrow-phase full updates in the last step */
    int32_t overflow_flags;        /* nonzero = a device buffer overflowed (see rp_last_error) */
    int32_t quarantined;           /* bodies with non-finite state detected (Quarantine) */
    int32_t fast_steps;            /* step graphs enqueued on the steady-state fast path */
    int32_t full_steps;            /* step graphs enqueued on the full path */
    int32_t replayed_steps;        /* fast steps that gave up on the device and were replayed on the full path */
    int32_t num_sleeping_bodies;   /* dynamic bodies asleep (IslandManager: bodies outside the active set) */
    int32_t ccd_active_count;      /* (body, step) occurrences so far of RigidBodyCcd::is_moving_fast_with_next_position (worker.rs:845-865):
                                    * the bodies the continuous-collision pass looked at (full steps) */
    int32_t ccd_clamp_count;       /* (body, step) cases in which CCDSolver::solve_continuous clamped next_position to a time of impact */
    int32_t num_tiles;             /* LDS tiles the global solver path's big component is cut into (0 = colour stages run as launches) */
    int32_t tile_sweeps;           /* 1 = the last enqueued step ran its biased / relaxed sweeps as one launch over the tiles */
    int32_t bp_large_list;         /* colliders the broad phase keeps on its brute-force list: those spanning more than 3 grid cells (ground slabs,
                                    * half-spaces) and those that met a full hash bucket in the last full rebuild */
    int32_t lean_steps;            /* step graphs enqueued without the rebuild launches (colouring, layout, toucher ranks, tiling): worlds on the
                                    * per-stage / tile path whose contact graph stands still; validated on the device, a step whose narrow phase
                                    * found new work is resumed by the next full graph (counted in replayed_steps) */
    int32_t fused_steps;           /* of fast_steps: those enqueued as the ONE-kernel fused step (k_island_solve validates the step itself) */
    int32_t num_islands;           /* contact islands the last layout rebuild handed to the LDS-resident island kernel (a bundle of tiny islands counts once) */
    int32_t num_global_bodies;     /* awake dynamic bodies it left to the global solver path (components too large for an island, bodies with joints, free bodies) */
    int32_t fused_disabled;        /* how often a fused step waited ~1 s for a workgroup that never became resident (another process or stream
                                    * holds CUs): that step was aborted and replayed, and this world
                                    * holds back from the one-kernel fused step (it keeps the two-kernel fast graph) and tries it again 4,096 steps
                                    * later, four times as many after every further loss (RP_ONE_LAUNCH_RETRY=<steps>, 0 = never) — a count that
                                    * keeps rising = the GPU stays shared */
    int32_t fused_launches;        /* launches that carried the fused_steps: a world whose islands fit one island per workgroup takes up to 32 fused
                                    * steps per launch (k_island_solve_steps: a step boundary inside the launch is a workgroup barrier and the
                                    * next step reads what this one wrote from the CU's own caches); fused_steps / fused_launches = steps per launch */
    int32_t joint_net_steps;       /* of lean_steps: those whose whole TGS loop was ONE launch that keeps every tile's joints in registers
                                    * (k_joint_net_step: worlds of spherical impulse joints without a single contact manifold — b3d_joint_grid) */
    int32_t joint_net_disabled;    /* how often a tile of that launch waited ~2 s for a neighbouring tile whose workgroup never became resident (another
                                    * process or stream holds CUs): the step died without writing anything and was resumed by the full graph, and the
                                    * world takes the sweep launches until the same retry rule brings the launch back (tile_step_steps' launch counts here too) */
    int32_t tile_step_steps;       /* of lean_steps: those whose whole TGS loop was ONE launch over the LDS tiles of a contact world (k_tile_step: the
                                    * prepare / increment / biased / relaxed launches of every substep as phases of one kernel, a tile waits for its
                                    * neighbouring tiles' flags instead of a kernel boundary — b3d_large_pyramid) */
} rp_counters;

#define RP_INVALID_HANDLE 0xffffffffffffffffull

typedef struct rp_world rp_world;

/* PhysicsWorld::new + the device to live on.  Fails (RP_ERR_DEVICE) when no HIP device. */
int32_t rp_world_create(const rp_integration_params *params, const float gravity[3], int32_t device, rp_world **out);
int32_t rp_world_destroy(rp_world *w);
const char *rp_last_error(const rp_world *w);
void rp_default_params(rp_integration_params *out);            /* IntegrationParameters::default() */
int32_t rp_params_get(const rp_world *w, rp_integration_params *out);
int32_t rp_params_set(rp_world *w, const rp_integration_params *params);

/* RigidBodySet::insert ×n; handles = generation<<32 | index (arena.rs:58-90); removed slots are reused first, LIFO (arena.rs:260-290). */
int32_t rp_bodies_insert(rp_world *w, int32_t n, const rp_body_desc *descs, uint64_t *handles_out);
/* ColliderSet::insert_with_parent ×n (parent RP_INVALID_HANDLE = ColliderSet::insert).  Device path scope: cuboid / ball
 * shapes.  A body may carry any number of colliders at any pos_wrt_parent (compound bodies): its mass, centre of mass and
 * principal inertia are the sum of the colliders' MassProperties (rigid_body_components.rs:421-489). */
int32_t rp_colliders_insert(rp_world *w, int32_t n, const rp_collider_desc *descs, const uint64_t *parents, uint64_t *handles_out);
/* SharedShape::convex_mesh(points, indices) — or SharedShape::convex_hull(points) when `indices` is NULL (ColliderBuilder::convex_mesh /
 * convex_hull, /root/reference/src/geometry/collider.rs:1039, :1070): registers a convex polyhedron with the world and hands out the id
 * colliders refer to (shape = RP_SHAPE_CONVEX_POLYHEDRON, half_extents[0] = id; any number of colliders may share one polyhedron).
 * `indices` = n_triangles x 3 vertex indices of a closed triangle mesh wound counter-clockwise seen from outside; triangles with equal
 * normals become one polygonal face.  RP_ERR_INVALID where the reference's builders return None (no volume, not closed) and for more
 * than 256 hull vertices.  The polyhedron's centre of mass and inertia tensor follow MassProperties::from_convex_polyhedron. */
int32_t rp_convex_polyhedron_create(rp_world *w, int32_t n_points, const float *points_xyz, int32_t n_triangles, const uint32_t *indices, int32_t *id_out);
/* SharedShape::compound(parts) (ColliderBuilder::compound, collider.rs:711): registers a compound shape; of every part descriptor only
 * shape (a primitive or round primitive: no half-space, no composite), half_extents, translation, rotation and border_radius are
 * read.  Colliders refer to it with shape = RP_SHAPE_COMPOUND, half_extents[0] = id. */
int32_t rp_compound_create(rp_world *w, int32_t n_parts, const rp_collider_desc *parts, int32_t *id_out);
/* SharedShape::trimesh(vertices, indices) (ColliderBuilder::trimesh, collider.rs:944; no TriMeshFlags): n_triangles x 3 vertex indices.
 * Colliders: shape = RP_SHAPE_TRIMESH, half_extents[0] = id, on a fixed or kinematic body (or none). */
int32_t rp_trimesh_create(rp_world *w, int32_t n_vertices, const float *vertices_xyz, int32_t n_triangles, const uint32_t *indices, int32_t *id_out);
/* SharedShape::heightfield(heights, scale) (ColliderBuilder::heightfield, collider.rs:1089): heights[r * ncols + c] over the unit square
 * (r along z, c along x) scaled by `scale`; every cell is cut into two triangles along its (r, c) -> (r + 1, c + 1) diagonal and the
 * field is served by the triangle-mesh path: colliders use shape = RP_SHAPE_TRIMESH with the id returned here. */
int32_t rp_heightfield_create(rp_world *w, int32_t nrows, int32_t ncols, const float *heights, const float scale[3], int32_t *id_out);
/* ConvexPolyhedron::points / faces / ... of a registered polyhedron as the library holds it (canonical form: DESIGN.md §4.4).
 * counts = {vertices, faces, vertex-loop entries, edges}; then, each optional (NULL = skip): the vertices recentred on the centre of
 * their bounding box, the unit face normals, per face the first entry and the length of its counter-clockwise vertex loop, the
 * loops' vertices and edges, props = {box centre xyz, box half extents xyz, max |vertex|, bounding-sphere centre xyz and radius,
 * volume, centre of mass xyz, inertia tensor about it at unit density: xx, yy, zz, xy, xz}. */
int32_t rp_convex_polyhedron_read(const rp_world *w, int32_t id, int32_t counts[4], float *points_xyz, float *face_normals, int32_t *face_first, int32_t *face_count,
                                  int32_t *loop_vertex, int32_t *loop_edge, float props[20]);
/* ImpulseJointSet::insert ×n (impulse_joint_set.rs).  handles_out = ImpulseJointHandles: generation << 32 | index of the slot in the
 * set's arena (joint_ids: Arena<..>, data/arena.rs:58-90, 260-290): the slot of a removed joint is handed out again LIFO with the
 * arena's removal count as its generation, a handle of an earlier occupant is refused with RP_ERR_INVALID by every call.
 * Device path scope: any JointAxesMask of locked
 * linear / angular axes (spherical 0x07, revolute 0x37, prismatic without limits 0x3e, fixed 0x3f; the free axis is the
 * local frame's X axis as in RevoluteJointBuilder / PrismaticJointBuilder); contacts_enabled = 0 filters the contact pairs between
 * the two bodies (pair_update.rs:191-201); limit_axes / limits bound the free axes (limit_linear, limit_angular:
 * joint_constraint_helper.rs:166-208, 468-564); motor_axes / motors drive them (motor_linear, motor_angular: :285-331, 566-625;
 * motor rows are solved before the lock and limit rows, joint_velocity_constraint.rs:186-246); coupled_axes: see rp_joint_desc. */
int32_t rp_impulse_joints_insert(rp_world *w, int32_t n, const rp_joint_desc *descs, uint64_t *handles_out);
/* GenericJoint::set_motor / set_motor_velocity / set_motor_position / set_motor_max_force / set_motor_model
 * (generic_joint.rs:538-603) through ImpulseJointSet::get_mut(handle, wake_up_connected_bodies = true)
 * (impulse_joint_set.rs:235-250): axes[i] (0..5 = LinX..AngZ) of joint handles[i] gets motors[i], its motor is enabled,
 * and both bodies of the joint are woken. */
int32_t rp_impulse_joints_set_motor(rp_world *w, int32_t n, const uint64_t *handles, const int32_t *axes, const rp_joint_motor *motors);
/* ImpulseJointSet::get(handle) (impulse_joint_set.rs:222-233): the live descriptors of n joints — as inserted, with every
 * rp_impulse_joints_set_motor edit since; body1 / body2 are RigidBodyHandles.  A stale or removed handle: RP_ERR_INVALID. */
int32_t rp_impulse_joints_get(const rp_world *w, int32_t n, const uint64_t *handles, rp_joint_desc *descs_out);
/* JointMotor::impulse of the six axes of n joints (NULL handles = all, insertion order), as written back by the last step. */
int32_t rp_impulse_joints_read_motor_impulses(rp_world *w, int32_t n, const uint64_t *handles, float *impulse6_out);
/* ImpulseJoint::impulses (per locked linear dof, as written back by the last step) and the persistent
 * solver colour of n joints (NULL handles = all, insertion order). */
int32_t rp_impulse_joints_read(rp_world *w, int32_t n, const uint64_t *handles, int32_t *color_out, float *impulse3_out);

/* RigidBodySet::remove (with its attached colliders and joints, rigid_body_set.rs:121-170),
 * ColliderSet::remove (collider_set.rs), ImpulseJointSet::remove (impulse_joint_set.rs).  Arena slots are
 * kept as tombstones: indices stay stable, removed handles become invalid, the pairs of a removed
 * collider are deleted by the next broad-phase pass while every other pair keeps its warm-start data. */
int32_t rp_bodies_remove(rp_world *w, int32_t n, const uint64_t *handles);
int32_t rp_colliders_remove(rp_world *w, int32_t n, const uint64_t *handles);
int32_t rp_impulse_joints_remove(rp_world *w, int32_t n, const uint64_t *handles);

/* PhysicsWorld::step() × nsteps with hooks = &(), events = &() (physics_world.rs:120-157). */
int32_t rp_step(rp_world *w, uint32_t nsteps);
/* Block until every enqueued step has finished (hipStreamSynchronize). */
int32_t rp_sync(rp_world *w);

/* RigidBody::position()/linvel()/angvel() for n handles (NULL handles = all bodies in arena order):
 * pos7 = tx,ty,tz,qx,qy,qz,qw ; vel6 = linvel, angvel.  Synchronises. */
int32_t rp_bodies_read(rp_world *w, int32_t n, const uint64_t *handles, float *pos7_out, float *vel6_out);
/* RigidBody::set_linvel/set_angvel/set_position(.., wake_up = true) (user changes).  NULL arrays are left
 * untouched.  The written bodies are woken (strong); a moved body also wakes every body it has a contact
 * pair with (pair_management.rs:236-258). */
int32_t rp_bodies_write(rp_world *w, int32_t n, const uint64_t *handles, const float *pos7, const float *vel6);
/* RigidBody::{reset_forces, reset_torques} (when `reset` != 0) followed by add_force / add_torque (.., wake_up = true)
 * (rigid_body.rs:1145-1252) for n dynamic bodies: force3 / torque3 are n x 3 (NULL = skip).  The user force persists across
 * steps until reset, exactly like RigidBodyForces::user_force. */
int32_t rp_bodies_add_force(rp_world *w, int32_t n, const uint64_t *handles, const float *force3, const float *torque3, int32_t reset);
/* RigidBody::{apply_impulse, apply_torque_impulse} (.., wake_up = true) (rigid_body.rs:1304-1343): linvel += impulse *
 * effective_inv_mass, angvel += effective_world_inv_inertia * torque_impulse. */
int32_t rp_bodies_apply_impulse(rp_world *w, int32_t n, const uint64_t *handles, const float *impulse3, const float *torque_impulse3);
/* Sleeping — RigidBodyActivation (rigid_body_components.rs:1300-1480), whole-island sleep
 * (island_manager/manager.rs:335-388, sleep.rs).  Bodies built with can_sleep = 1 fall asleep with their whole
 * contact island once EVERY member stayed below the motion thresholds for time_until_sleep (0.5 s), and wake
 * island-wide on a begin-touch, a deleted touching pair, a removed collider or a user change.
 * rp_bodies_wake_up = IslandManager::wake_up(handle, strong) (sleep.rs:31), effective at the next step;
 * rp_bodies_is_sleeping = RigidBody::is_sleeping (NULL handles are not accepted).  Impulse joints link the islands of their
 * bodies; a joint whose bodies sleep leaves the solver selection (impulse_joint_set.rs:504-572). */
/* RigidBody::set_additional_solver_iterations (rigid_body.rs): counts[i] >= 0 extra substeps for the component of handles[i]. */
int32_t rp_bodies_set_additional_solver_iterations(rp_world *w, int32_t n, const uint64_t *handles, const int32_t *counts);
int32_t rp_bodies_wake_up(rp_world *w, int32_t n, const uint64_t *handles, int32_t strong);
/* RigidBody::set_next_kinematic_position (rigid_body.rs:1085-1093) for n kinematic bodies: the pose to reach by the
 * end of the next step.  Position-based kinematic bodies get their velocity from it (interpolate_kinematic_velocities,
 * substep.rs:242-264) and land on it exactly; the body is woken when the pose differs from its current one.
 * Kinematic bodies are solver bodies with zero inverse mass: they push dynamic bodies and are never pushed.
 * Worlds holding kinematic bodies take the full step path. */
int32_t rp_bodies_set_next_kinematic_position(rp_world *w, int32_t n, const uint64_t *handles, const float *pos7);
int32_t rp_bodies_is_sleeping(rp_world *w, int32_t n, const uint64_t *handles, int32_t *sleeping_out);
/* IslandManager::persistent_island_of (island_manager/manager.rs:214-220): the persistent island of each body — one connected
 * component of the touching-contact / joint graph, merged eagerly, split lazily (persistent.rs, local_split.rs, global_split.rs:
 * an island that lost constraints may not sleep before its deferred, cooldown-throttled split).  -1 for fixed and removed bodies,
 * and for every body of a world that holds no sleepable body (such a world keeps no islands; the first sleepable body bootstraps
 * them, persistent.rs:600-625).  As in the reference only EQUALITY of two ids is meaningful. */
int32_t rp_bodies_persistent_island(rp_world *w, int32_t n, const uint64_t *handles, int32_t *island_out);
/* RigidBodySet::iter / ColliderSet::iter as handles (Arena::iter, data/arena.rs:665-700): the handle of every arena row in index
 * order — generation << 32 | index of the row's occupant (a free row: of the occupant removed last; only rp_bodies_read still accepts
 * it).  Removed slots are handed out again LIFO with the arena's removal count as their generation (arena.rs:260-290, 353-380): a
 * handle of an earlier occupant is stale and rejected with RP_ERR_INVALID everywhere.  Returns the row count, writes min(rows, cap). */
int32_t rp_bodies_handles(const rp_world *w, int32_t cap, uint64_t *handles_out);
int32_t rp_colliders_handles(const rp_world *w, int32_t cap, uint64_t *handles_out);
/* ImpulseJointSet::iter as handles (impulse_joint_set.rs:282-291): one entry per joint ever inserted, in insertion order — the order of
 * the NULL-handle reads of rp_impulse_joints_read: the handle of a live joint, RP_INVALID_HANDLE for a removed one.  Returns the
 * number of entries, writes min(entries, cap). */
int32_t rp_impulse_joints_handles(const rp_world *w, int32_t cap, uint64_t *handles_out);
/* Proximity groups: connected components of the non-fixed bodies over everything that can couple them within a step — every live
 * broad-phase pair (fat AABBs overlap: ColliderPair events of broad_phase_bvh/mod.rs:171-263) and every impulse joint.  Two bodies in
 * different groups cannot interact before one of them moves out of its fat AABB: the unit of island sharding over GPUs (SURVEY §8e;
 * the reference's islands, island_manager/persistent.rs, are the touching subset).  -1 for fixed / removed bodies; only EQUALITY of
 * two ids is meaningful.  Needs at least one step (the pair set is built by the first broad-phase pass). */
int32_t rp_bodies_proximity_group(rp_world *w, int32_t n, const uint64_t *handles, int32_t *group_out);
/* Shard guard of a world that holds ONE shard of a larger scene: n axis-aligned boxes (min xyz / max xyz, e.g. one per proximity
 * group of the other shards, already inflated by whatever clearance the caller wants) that hold the bodies of OTHER shards.  When
 * the fat AABB of a non-fixed body of this world is rewritten so that it overlaps one of them, the shards are no longer independent
 * (the reference would have created the ColliderPair; a sharded run cannot): the next rp_sync / read returns RP_ERR_INVALID.
 * LIMIT: the boxes are static — where the other shards' bodies WERE when the caller took them.  Two bodies of different shards
 * that both leave their boxes and meet in between are not seen by either guard.  A caller that shards a world whose groups roam
 * (not the pyramids of the benchmark scenes) must refresh the boxes periodically: gather the groups' current boxes over all
 * ranks and call this function again (one all-gather of 6 floats per group).
 * n = 0 removes the guard. */
int32_t rp_world_set_shard_guard(rp_world *w, int32_t n, const float *box_min3, const float *box_max3);
/* The bodies whose rewritten fat AABB overlapped a guard box since the last call (handles; returns their number, writes min(n, cap);
 * call with cap >= n to consume them): the guard bit is cleared, rp_sync and the reads succeed again, and the world can go on — the
 * caller moves the bodies' proximity group to the shard that owns the box they reached ("when an AddPair links bodies
 * on different shards, migrate the smaller island": rapier_amd/sharding.py migrate_groups) and sets fresh guards on both sides. */
int32_t rp_world_shard_guard_take_hits(rp_world *w, int32_t cap, uint64_t *bodies_out);
/* How long a hit may wait for the caller, in seconds (steps between two rp_world_shard_guard_take_hits x dt; default 0): the device
 * tests every rewritten fat AABB inflated by |linvel of its body| x seconds, so a fast body is caught that much earlier while slow
 * neighbours of a foreign box are left alone.  The boxes themselves should carry the same allowance for THEIR group's speed. */
int32_t rp_world_set_shard_guard_horizon(rp_world *w, float seconds);
/* max |linvel| over the non-fixed bodies (device reduction, one 4-byte read-back; pending steps run first).  A caller that consumes the
 * guard's hits every k steps inflates the boxes by 2 * speed * dt * k so that no body crosses a box between two looks
 * (no reference counterpart: the reference has one address space). */
int32_t rp_world_max_linear_speed(rp_world *w, float *out);
/* Batches of small worlds (no reference counterpart: the reference steps one World per call; the closest reference construct is ONE
 * World whose PhysicsHooks::filter_contact_pair, physics_hooks.rs:203, rejects pairs across the groups).  Everything inserted after
 * rp_world_begin_subworld belongs to a new sub-world (returns its index; the bodies / colliders inserted before the first call are
 * sub-world 0); colliders of different sub-worlds never pair, so sub-worlds may overlap in space; they share the integration
 * parameters and every launch of rp_step.  The result is that of the one reference world described above — for n copies of one scene,
 * bit for bit the result of stepping each copy alone (colour populations then agree; tests/test_gpu_subworlds.py). */
int32_t rp_world_begin_subworld(rp_world *w);
/* rp_step(worlds[i], steps) for every i, enqueued back to back on the worlds' own streams before anything is waited for. */
int32_t rp_step_many(rp_world *const *worlds, int32_t n, int32_t steps);
int32_t rp_num_bodies(const rp_world *w);

/* ---- The collective of a sharded world (no reference counterpart: the reference is one address space) ----------
 * Islands shard over the GPUs of a node, one process and one rp_world per GPU, with no collective in a step.  What crosses xGMI is ONE
 * RCCL all-gather of packed body state when a caller wants the whole world back.  RCCL (librccl.so.1) is bound at run time when the
 * first communicator is asked for; the library has no link-time dependency on it. */
typedef struct rp_comm rp_comm;
typedef struct rp_comm_id { char internal[128]; } rp_comm_id; /* ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128) */
/* ncclGetUniqueId: called by ONE rank; the 128 bytes travel to the other ranks by any host channel (MPI, a file, torch.distributed). */
int32_t rp_comm_unique_id(rp_comm_id *id_out);
/* ncclCommInitRank on `device` (collective: every rank of the job calls it with the same id and world_size). */
int32_t rp_comm_create(const rp_comm_id *id, int32_t world_size, int32_t rank, int32_t device, rp_comm **out);
int32_t rp_comm_destroy(rp_comm *c);
const char *rp_comm_last_error(const rp_comm *c); /* c == NULL: the calling thread's last rp_comm_unique_id / rp_comm_create failure */
/* The global id of every arena row of this shard (ids[i] for body index i; rows beyond n, and a world that never calls this, use the
 * arena index): what the packed rows carry so that a gather can be scattered into the order of the unsharded world. */
int32_t rp_world_set_global_ids(rp_world *w, int32_t n, const int64_t *ids);
/* Packs the state of the world's live non-fixed bodies ON THE DEVICE (pending steps run first): one 64-byte row per body =
 * { int64 global id, translation xyz, rotation xyzw, linvel xyz, angvel xyz, 0 } (16 words), rows in no particular order.  Hands out
 * the device pointer (owned by the world, valid until the next pack / gather / destroy) and the row count. */
int32_t rp_world_pack_bodies(rp_world *w, const void **dev_rows_out, int32_t *n_rows_out);
/* The readback collective: packs this shard's bodies (as above, padded with id -1 rows to rows_per_rank — a value every rank agrees on,
 * >= the largest shard's body count), ncclAllGather on the world's stream (device to device over xGMI; no host copy in front of it),
 * one D2H of the gathered rows, and a scatter by global id into pos7_out[7 * n_global] / vel6_out[6 * n_global] (either may be NULL;
 * rows nobody owns — the replicated fixed bodies — are left as the caller initialised them).  rows_of_rank_out (optional,
 * world_size ints) = the bodies each rank contributed.  RP_ERR_CAPACITY when this rank owns more than rows_per_rank bodies. */
int32_t rp_shard_all_gather(rp_world *w, rp_comm *c, int32_t rows_per_rank, int64_t n_global, float *pos7_out, float *vel6_out, int32_t *rows_of_rank_out);

/* NarrowPhase::contact_pairs() analogue: for each active solver manifold: (collider1, collider2,
 * colour, num solver contacts), world normal, total normal impulse per solver contact.
 * Returns the number of active manifolds (may exceed cap; only cap are written). */
int32_t rp_contacts_read(rp_world *w, int32_t cap, int32_t *c1_c2_color_count, float *normal3, float *impulse4);

/* Quarantine (pipeline/physics_pipeline/quarantine.rs:68-131): handles of the bodies whose state went
 * non-finite (rolled back to the last valid pose and stopped).  Returns the count (may exceed cap). */
int32_t rp_quarantine_read(rp_world *w, int32_t cap, uint64_t *handles_out);

/* EventHandler::{handle_collision_event, handle_contact_force_event} (pipeline/event_handler.rs:94-160) as queues
 * instead of callbacks: the device appends an event whenever a pair with ActiveEvents::COLLISION_EVENTS starts / stops
 * touching (contacts.rs:316-323; a deleted touching pair or a removed collider raises Stopped, the latter with
 * RP_COLLISION_EVENT_REMOVED) and, after every step, for every solver-active pair with ActiveEvents::CONTACT_FORCE_EVENTS
 * whose total contact force exceeds the smaller of the two colliders' thresholds (solver_graph.rs:462-498).  With
 * out == NULL a call returns the number of queued events and consumes nothing.  Otherwise it writes the oldest
 * min(queued, cap) events (sorted by step, collider1, collider2), removes exactly those from the queue and returns how many
 * it wrote; the rest stays queued for the next call.  (The queues hold max(65,536, pair slots) events — a step raises at most one
 * collision and one force event per pair, so a queue that is read every step cannot overflow; one that did overflow between two reads
 * has dropped the newest events and says so in rp_last_error.) */
int32_t rp_collision_events_read(rp_world *w, int32_t cap, rp_collision_event *out);
/* NarrowPhase::intersection_pairs (narrow_phase/queries.rs:150-190): every pair that involves a sensor collider, as triples
 * (collider1, collider2, intersecting 0|1) in triples3[3 * i ..]; returns the number of such pairs (may exceed cap). */
int32_t rp_intersection_pairs_read(rp_world *w, int32_t cap, int32_t *triples3);
int32_t rp_contact_force_events_read(rp_world *w, int32_t cap, rp_contact_force_event *out);

/* PhysicsPipeline::counters; `enable_timers` != 0 turns on hipEvent stage timing (off = no events). */
int32_t rp_counters_enable(rp_world *w, int32_t enable_timers);
int32_t rp_counters_read(rp_world *w, rp_counters *out);

/* Average device time (ms) of k_island_solve (the TGS velocity-solve loop of every LDS-resident island)
 * per step since the last call, measured with hipEvents on the world's stream while timers are
 * enabled (bench.py roofline leg). */
int32_t rp_solver_loop_time_ms(rp_world *w, float *avg_ms_per_step, int32_t *steps_measured);

#ifdef __cplusplus
}
#endif
#endif
