"""Closed-form scene generators for the hot-path configs.

Each generator restates the formulas of the reference example it is named after and
returns a :class:`Scene` of plain numpy descriptor arrays (the same layouts the C ABI in
``include/rapier_hip.h`` takes).  No RNG anywhere: the scenes are closed-form.

* ``many_pyramids``  — /root/reference/examples3d/b3d_many_pyramids.rs:9-64
* ``large_pyramid``  — /root/reference/examples3d/b3d_large_pyramid.rs:15-36
* ``joint_grid``     — /root/reference/examples3d/b3d_joint_grid.rs:17-53
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

BODY_DYNAMIC, BODY_FIXED, BODY_KINEMATIC_POSITION, BODY_KINEMATIC_VELOCITY = 0, 1, 2, 3  # RigidBodyType
SHAPE_BALL, SHAPE_CUBOID, SHAPE_CAPSULE = 0, 1, 2  # capsule: half_extents = (half_height, radius, axis 0|1|2) = ColliderBuilder::capsule_x/y/z
SHAPE_HALFSPACE = 3  # half_extents = the unit outward normal in the collider frame = ColliderBuilder::halfspace (fixed or kinematic parents only)
# ColliderBuilder::round_cuboid / round_cylinder / round_cone / round_convex_hull: the inner shape's half_extents + collider_desc(border_radius=...)
SHAPE_ROUND_CUBOID, SHAPE_ROUND_CYLINDER, SHAPE_ROUND_CONE, SHAPE_ROUND_CONVEX_POLYHEDRON = 7, 8, 9, 10
SHAPE_CONVEX = SHAPE_CONVEX_POLYHEDRON = 6  # half_extents[0] = the id Scene.add_convex_polyhedron returned = ColliderBuilder::convex_mesh / convex_hull (collider.rs:1039, :1070)
# Composite shapes as ONE collider: half_extents[0] = the id Scene.add_compound / add_trimesh / add_heightfield returned (ColliderBuilder::compound,
# ::trimesh, ::heightfield — collider.rs:711, :944, :1089); a triangle mesh / height field needs a fixed or kinematic parent (or none)
SHAPE_COMPOUND, SHAPE_TRIMESH = 11, 12
SHAPE_TRIANGLE = 13  # internal to the shape dispatchers (one triangle of a mesh); never a collider's shape
SHAPE_CYLINDER, SHAPE_CONE = 4, 5  # half_extents = (half_height, radius, -) = ColliderBuilder::cylinder / cone (axis Y, a cone's apex at +Y)
RULE_AVERAGE, RULE_MIN, RULE_MULTIPLY, RULE_MAX, RULE_CLAMPED_SUM, RULE_GEOMETRIC_MEAN = range(6)

# Field-for-field mirrors of rp_body_desc / rp_collider_desc / rp_joint_desc / rp_integration_params.
BODY_DTYPE = np.dtype([
    ("body_type", "<i4"), ("translation", "<f4", 3), ("rotation", "<f4", 4),
    ("linvel", "<f4", 3), ("angvel", "<f4", 3),
    ("linear_damping", "<f4"), ("angular_damping", "<f4"), ("gravity_scale", "<f4"),
    ("additional_mass", "<f4"), ("dominance", "<i4"), ("gyroscopic", "<i4"),
    ("allow_fast_rotation", "<i4"), ("can_sleep", "<i4"), ("locked_axes", "<u4"), ("additional_solver_iterations", "<i4"), ("ccd_enabled", "<i4"),
], align=False)
COLLIDER_DTYPE = np.dtype([
    ("shape", "<i4"), ("half_extents", "<f4", 3), ("translation", "<f4", 3), ("rotation", "<f4", 4),
    ("density", "<f4"), ("friction", "<f4"), ("restitution", "<f4"),
    ("friction_rule", "<i4"), ("restitution_rule", "<i4"),
    ("collision_memberships", "<u4"), ("collision_filter", "<u4"),
    ("active_events", "<u4"), ("contact_force_event_threshold", "<f4"), ("sensor", "<i4"), ("border_radius", "<f4"),
], align=False)

ACTIVE_EVENTS_COLLISION, ACTIVE_EVENTS_CONTACT_FORCE = 1, 2  # ActiveEvents bits
MOTOR_DTYPE = np.dtype([  # rp_joint_motor = JointMotor (generic_joint.rs:200-232)
    ("target_vel", "<f4"), ("target_pos", "<f4"), ("stiffness", "<f4"), ("damping", "<f4"), ("max_force", "<f4"), ("model", "<i4"),
], align=False)
MOTOR_ACCELERATION_BASED, MOTOR_FORCE_BASED = 0, 1  # MotorModel
F32_MAX = float(np.finfo(np.float32).max)
JOINT_DTYPE = np.dtype([  # rp_joint_desc: body1 / body2 are RigidBodyHandles (generation << 32 | index; a scene holds plain indices = generation 0)
    ("body1", "<u8"), ("body2", "<u8"), ("local_anchor1", "<f4", 3), ("local_anchor2", "<f4", 3),
    ("local_basis1", "<f4", 4), ("local_basis2", "<f4", 4), ("locked_axes", "<u4"),
    ("contacts_enabled", "<i4"), ("limit_axes", "<u4"), ("limits", "<f4", (6, 2)),
    ("motor_axes", "<u4"), ("motors", MOTOR_DTYPE, 6), ("coupled_axes", "<u4"), ("reserved", "<u4"),
], align=False)


def motor_desc(target_vel=0.0, target_pos=0.0, stiffness=0.0, damping=0.0, max_force=F32_MAX, model=MOTOR_ACCELERATION_BASED) -> np.ndarray:
    """JointMotor::default() (generic_joint.rs:216-228) with the fields GenericJoint::set_motor* write."""
    m = np.zeros((), dtype=MOTOR_DTYPE)
    m["target_vel"], m["target_pos"], m["stiffness"], m["damping"] = target_vel, target_pos, stiffness, damping
    m["max_force"], m["model"] = max_force, model
    return m
PARAMS_DTYPE = np.dtype([
    ("dt", "<f4"),
    ("contact_natural_frequency", "<f4"), ("contact_damping_ratio", "<f4"),
    ("static_contact_natural_frequency", "<f4"), ("static_contact_damping_ratio", "<f4"),
    ("joint_natural_frequency", "<f4"), ("joint_damping_ratio", "<f4"),
    ("warmstart_coefficient", "<f4"),
    ("normalized_allowed_linear_error", "<f4"), ("normalized_max_corrective_velocity", "<f4"),
    ("normalized_prediction_distance", "<f4"), ("normalized_max_linear_velocity", "<f4"),
    ("normalized_contact_recycle_distance", "<f4"), ("length_unit", "<f4"),
    ("num_solver_iterations", "<i4"), ("num_internal_pgs_iterations", "<i4"),
    ("num_internal_stabilization_iterations", "<i4"), ("contact_recycling", "<i4"),
    ("friction_in_bias_pass", "<i4"), ("warmstart_joints", "<i4"), ("max_ccd_substeps", "<i4"),
    ("friction_model", "<i4"), ("min_ccd_dt", "<f4"), ("contact_clustering", "<i4"),
], align=False)

FRICTION_SIMPLIFIED, FRICTION_COULOMB = 0, 1  # FrictionModel, integration_parameters.rs:13-32

LOCK_LIN = 0b000111       # JointAxesMask::LIN_AXES (spherical joint)
LOCK_ALL = 0b111111       # LOCKED_FIXED_AXES (fixed joint)
LOCK_REVOLUTE = 0b110111  # LOCKED_REVOLUTE_AXES: everything but the rotation about the frame's X axis
LOCK_PRISMATIC = 0b111110  # LOCKED_PRISMATIC_AXES: everything but the translation along the frame's X axis


def default_params() -> np.ndarray:
    """IntegrationParameters::default() — integration_parameters.rs:379-408."""
    p = np.zeros((), dtype=PARAMS_DTYPE)
    p["dt"] = np.float32(1.0) / np.float32(60.0)
    p["contact_natural_frequency"], p["contact_damping_ratio"] = 30.0, 10.0
    p["static_contact_natural_frequency"], p["static_contact_damping_ratio"] = 60.0, 10.0
    p["joint_natural_frequency"], p["joint_damping_ratio"] = 1.0e6, 1.0
    p["warmstart_coefficient"] = 1.0
    p["normalized_allowed_linear_error"] = 0.005
    p["normalized_max_corrective_velocity"] = 3.0
    p["normalized_prediction_distance"] = 0.02
    p["normalized_max_linear_velocity"] = 400.0
    p["normalized_contact_recycle_distance"] = 0.05
    p["length_unit"] = 1.0
    p["num_solver_iterations"] = 4
    p["num_internal_pgs_iterations"] = 1
    p["num_internal_stabilization_iterations"] = 1
    p["contact_recycling"] = 1
    p["friction_in_bias_pass"] = 0
    p["warmstart_joints"] = 0
    p["max_ccd_substeps"] = 1
    p["friction_model"] = FRICTION_SIMPLIFIED
    p["min_ccd_dt"] = np.float32(1.0) / np.float32(60.0) / np.float32(100.0)
    p["contact_clustering"] = 1
    return p


def body_desc(body_type=BODY_DYNAMIC, translation=(0, 0, 0), rotation=(0, 0, 0, 1), linvel=(0, 0, 0),
              angvel=(0, 0, 0), linear_damping=0.0, angular_damping=0.0, gravity_scale=1.0,
              additional_mass=0.0, dominance=0, gyroscopic=1, allow_fast_rotation=0, can_sleep=0, locked_axes=0,
              additional_solver_iterations=0, ccd_enabled=0) -> np.ndarray:
    """RigidBodyBuilder defaults — /root/reference/src/dynamics/rigid_body.rs:1560-1600 — except
    ``can_sleep``: the builder's default is true, every b3d benchmark scene calls ``.can_sleep(false)``
    (b3d_many_pyramids.rs:52) and so do the generators here unless a scene asks for sleeping."""
    b = np.zeros((), dtype=BODY_DTYPE)
    b["body_type"] = body_type
    b["translation"] = translation
    b["rotation"] = rotation
    b["linvel"], b["angvel"] = linvel, angvel
    b["linear_damping"], b["angular_damping"] = linear_damping, angular_damping
    b["gravity_scale"], b["additional_mass"] = gravity_scale, additional_mass
    b["dominance"], b["gyroscopic"], b["allow_fast_rotation"] = dominance, gyroscopic, allow_fast_rotation
    b["can_sleep"] = can_sleep
    b["locked_axes"] = locked_axes  # LockedAxes bits: 1,2,4 = translation x,y,z; 8,16,32 = rotation x,y,z
    b["additional_solver_iterations"] = additional_solver_iterations  # extra TGS substeps for the body's whole component
    b["ccd_enabled"] = ccd_enabled  # RigidBodyBuilder::ccd_enabled: a "bullet" also sweeps kinematic / dynamic targets (dynamics/ccd/sweeps.rs:29-41)
    return b


def collider_desc(shape=SHAPE_CUBOID, half_extents=(0.5, 0.5, 0.5), translation=(0, 0, 0),
                  rotation=(0, 0, 0, 1), density=1.0, friction=0.5, restitution=0.0,
                  friction_rule=RULE_AVERAGE, restitution_rule=RULE_AVERAGE,
                  memberships=0xFFFFFFFF, filter=0xFFFFFFFF, active_events=0, contact_force_event_threshold=0.0, sensor=0, border_radius=0.0) -> np.ndarray:
    """ColliderBuilder defaults — /root/reference/src/geometry/collider.rs:1125-1127 (friction 0.5,
    restitution 0, density 1, rule Average)."""
    c = np.zeros((), dtype=COLLIDER_DTYPE)
    c["shape"] = shape
    he = np.zeros(3, np.float32)
    he[: len(np.atleast_1d(half_extents))] = np.atleast_1d(half_extents)
    c["half_extents"] = he
    c["translation"], c["rotation"] = translation, rotation
    c["density"], c["friction"], c["restitution"] = density, friction, restitution
    c["friction_rule"], c["restitution_rule"] = friction_rule, restitution_rule
    c["collision_memberships"], c["collision_filter"] = memberships, filter
    c["active_events"], c["contact_force_event_threshold"] = active_events, contact_force_event_threshold
    c["sensor"] = sensor  # ColliderBuilder::sensor(true): intersection events only
    c["border_radius"] = border_radius  # round shapes (SHAPE_ROUND_*): RoundShape::border_radius
    return c


@dataclass
class Scene:
    name: str
    gravity: tuple = (0.0, -10.0, 0.0)
    params: np.ndarray = field(default_factory=default_params)
    bodies: list = field(default_factory=list)
    colliders: list = field(default_factory=list)
    collider_parents: list = field(default_factory=list)
    joints: list = field(default_factory=list)
    polyhedra: list = field(default_factory=list)   # (points (n, 3) f32, triangles (m, 3) u32 or None = "take the convex hull")
    subworlds: list = field(default_factory=list)   # batch(): [(first body, first collider, first joint)] of every sub-world (empty: one world)
    composites: list = field(default_factory=list)  # ("compound", parts: COLLIDER_DTYPE array) | ("trimesh", vertices (n, 3) f32, triangles (m, 3) u32) | ("heightfield", heights (r, c) f32, scale (3,))

    def add_compound(self, parts) -> int:
        """SharedShape::compound(parts): `parts` = collider descriptors (collider_desc(...)) of which shape, half_extents, translation,
        rotation and border_radius are read.  Colliders use the returned id: add_collider(b, shape=SHAPE_COMPOUND, half_extents=(id, 0, 0))"""
        self.composites.append(("compound", np.array(list(parts), dtype=COLLIDER_DTYPE)))
        return len(self.composites) - 1

    def add_trimesh(self, vertices, triangles) -> int:
        """SharedShape::trimesh(vertices, indices): add_collider(b, shape=SHAPE_TRIMESH, half_extents=(id, 0, 0))"""
        self.composites.append(("trimesh", np.ascontiguousarray(vertices, np.float32).reshape(-1, 3), np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)))
        return len(self.composites) - 1

    def add_heightfield(self, heights, scale) -> int:
        """SharedShape::heightfield(heights, scale): heights[r, c] (r along z, c along x) over the unit square scaled by `scale`; used
        like a triangle mesh: add_collider(b, shape=SHAPE_TRIMESH, half_extents=(id, 0, 0))"""
        self.composites.append(("heightfield", np.ascontiguousarray(heights, np.float32), np.asarray(scale, np.float32).reshape(3)))
        return len(self.composites) - 1

    def add_convex_polyhedron(self, points, triangles=None) -> int:
        """SharedShape::convex_mesh(points, indices) — or convex_hull(points) when `triangles` is None; colliders use the returned id:
        add_collider(b, shape=SHAPE_CONVEX, half_extents=(id, 0, 0))"""
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        tris = None if triangles is None else np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
        self.polyhedra.append((pts, tris))
        return len(self.polyhedra) - 1

    def add_body(self, **kw) -> int:
        self.bodies.append(body_desc(**kw))
        return len(self.bodies) - 1

    def enable_sleep(self, on: bool = True) -> "Scene":
        """RigidBodyBuilder::can_sleep(on) for every dynamic body of the scene."""
        for b in self.bodies:
            if int(b["body_type"]) == BODY_DYNAMIC:
                b["can_sleep"] = 1 if on else 0
        return self

    def enable_events(self, active_events: int = 3, threshold: float = 0.0) -> "Scene":
        """ColliderBuilder::active_events / contact_force_event_threshold for every collider of the scene."""
        for c in self.colliders:
            c["active_events"], c["contact_force_event_threshold"] = active_events, threshold
        return self

    def add_collider(self, parent: int, **kw) -> int:
        self.colliders.append(collider_desc(**kw))
        self.collider_parents.append(parent)
        return len(self.colliders) - 1

    def add_joint(self, body1, body2, anchor1, anchor2, locked_axes=LOCK_LIN, contacts_enabled=1,
                  basis1=(0, 0, 0, 1), basis2=(0, 0, 0, 1), limits=None, motors=None, coupled_axes=0) -> int:
        """``limits`` = {axis: (min, max)} with axis 0..2 = translation along the frame's X/Y/Z (metres), 3..5 = rotation about
        them (radians): GenericJoint::set_limits.  ``motors`` = {axis: motor_desc(...) or its keyword dict}:
        GenericJoint::set_motor / set_motor_velocity / set_motor_position / set_motor_max_force / set_motor_model."""
        j = np.zeros((), dtype=JOINT_DTYPE)
        j["body1"], j["body2"] = body1, body2
        j["local_anchor1"], j["local_anchor2"] = anchor1, anchor2
        j["local_basis1"] = basis1
        j["local_basis2"] = basis2
        j["locked_axes"], j["contacts_enabled"] = locked_axes, contacts_enabled
        j["coupled_axes"] = coupled_axes   # GenericJoint::coupled_axes (RopeJoint / SpringJoint: the three linear axes)
        for axis, (lo, hi) in (limits or {}).items():
            j["limit_axes"] |= np.uint32(1 << axis)
            j["limits"][axis] = (lo, hi)
        for a in range(6):
            j["motors"][a] = motor_desc()
        for axis, m in (motors or {}).items():
            j["motor_axes"] |= np.uint32(1 << axis)
            j["motors"][axis] = motor_desc(**m) if isinstance(m, dict) else m
        self.joints.append(j)
        return len(self.joints) - 1

    def add_rope_joint(self, body1, body2, anchor1, anchor2, max_dist: float, contacts_enabled=1) -> int:
        """RopeJoint::new(max_dist) (rope_joint.rs:31-38, set_max_distance :124-127): the three linear axes coupled, limited to [0, max_dist]"""
        return self.add_joint(body1, body2, anchor1, anchor2, locked_axes=0, contacts_enabled=contacts_enabled, limits={0: (0.0, max_dist)}, coupled_axes=LOCK_LIN)

    def add_spring_joint(self, body1, body2, anchor1, anchor2, rest_length: float, stiffness: float, damping: float, contacts_enabled=1) -> int:
        """SpringJoint::new(rest_length, stiffness, damping) (spring_joint.rs:31-40): the three linear axes coupled, a force-based position
        motor on LinX towards the rest length"""
        return self.add_joint(body1, body2, anchor1, anchor2, locked_axes=0, contacts_enabled=contacts_enabled, coupled_axes=LOCK_LIN,
                              motors={0: dict(target_pos=rest_length, stiffness=stiffness, damping=damping, model=MOTOR_FORCE_BASED)})

    # Packed arrays ---------------------------------------------------------------------
    def body_array(self) -> np.ndarray:
        return np.array(self.bodies, dtype=BODY_DTYPE) if self.bodies else np.zeros(0, BODY_DTYPE)

    def collider_array(self) -> np.ndarray:
        return np.array(self.colliders, dtype=COLLIDER_DTYPE) if self.colliders else np.zeros(0, COLLIDER_DTYPE)

    def parent_array(self) -> np.ndarray:
        return np.array(self.collider_parents, dtype=np.int32)

    def joint_array(self) -> np.ndarray:
        return np.array(self.joints, dtype=JOINT_DTYPE) if self.joints else np.zeros(0, JOINT_DTYPE)

    @property
    def num_dynamic(self) -> int:
        return int(sum(int(b["body_type"]) == BODY_DYNAMIC for b in self.bodies))


def batch(scenes, name: str | None = None) -> Scene:
    """Several small scenes as the SUB-WORLDS of one scene (rp_world_begin_subworld): bodies, colliders and joints concatenated (indices
    shifted), `subworlds` = where each one begins.  Colliders of different sub-worlds never pair, so the scenes may overlap in space;
    they must agree on gravity and integration parameters (one world steps them)."""
    scenes = list(scenes)
    first = scenes[0]
    out = Scene(name=name or f"batch_{len(scenes)}x_{first.name}", gravity=tuple(first.gravity), params=first.params.copy())
    for sc in scenes:
        if tuple(sc.gravity) != tuple(first.gravity) or sc.params.tobytes() != first.params.tobytes():
            raise ValueError("batch: the sub-worlds of a batch share gravity and integration parameters")
        if sc.subworlds:
            raise ValueError("batch: a batch of batches is not supported")
        nb, npoly, ncomp = len(out.bodies), len(out.polyhedra), len(out.composites)
        out.subworlds.append((nb, len(out.colliders), len(out.joints)))
        out.bodies += [b.copy() for b in sc.bodies]
        # registered shapes (convex polyhedra, compounds / meshes / height fields) are world-wide tables: the ids a collider — or a
        # compound's part — names move behind the ones already registered
        out.polyhedra += list(sc.polyhedra)
        for comp in sc.composites:
            if comp[0] == "compound":
                parts = comp[1].copy()
                for part in parts:
                    if int(part["shape"]) in (SHAPE_CONVEX_POLYHEDRON, SHAPE_ROUND_CONVEX_POLYHEDRON):
                        part["half_extents"][0] += npoly
                out.composites.append(("compound", parts))
            else:
                out.composites.append(comp)
        for c in sc.colliders:
            cc = c.copy()
            if int(cc["shape"]) in (SHAPE_CONVEX_POLYHEDRON, SHAPE_ROUND_CONVEX_POLYHEDRON):
                cc["half_extents"][0] += npoly
            elif int(cc["shape"]) in (SHAPE_COMPOUND, SHAPE_TRIMESH):
                cc["half_extents"][0] += ncomp
            out.colliders.append(cc)
        out.collider_parents += [(p + nb if p >= 0 else p) for p in sc.collider_parents]
        for j in sc.joints:
            jj = j.copy(); jj["body1"], jj["body2"] = int(j["body1"]) + nb, int(j["body2"]) + nb
            out.joints.append(jj)
    return out

def _f(x):
    return np.float32(x)


def _small_pyramid(scene: Scene, base_count: int, extent, center_x, base_z):
    """create_small_pyramid — b3d_many_pyramids.rs:9-29 (f32 arithmetic, same op order)."""
    extent = _f(extent)
    for i in range(base_count):
        y = (_f(2.0) * _f(i) + _f(1.0)) * extent
        for j in range(i, base_count):
            x = (_f(i) + _f(1.0)) * extent + _f(2.0) * _f(j - i) * extent + _f(center_x) - _f(0.5)
            b = scene.add_body(translation=(x, y, _f(base_z)))
            scene.add_collider(b, half_extents=(extent, extent, extent), density=100.0)


def large_world(grid: int = 1000, cell: float = 10.0) -> Scene:
    """b3d_large_world.rs:20-41 — box3d's `large_world` benchmark: a grid x grid floor of PARENTLESS fixed cuboids (half extents
    (cell / 2, 0.25, cell / 2); one million at the release settings) under gravity (0, -10, 0); the dynamic spheres are dropped while
    the world runs (`large_world_drop`).  The descriptors are built as one array: a million Scene.add_collider calls take longer
    than the benchmark."""
    s = Scene(name=f"b3d_large_world_{grid}", gravity=(0.0, -10.0, 0.0))
    half_span = _f(0.5) * _f(cell) * _f(grid)
    k = np.arange(grid, dtype=np.float32)
    xs = (-half_span + (k + _f(0.5)) * _f(cell)).astype(np.float32)      # x = -half_span + (i + 0.5) * cell, f32 like the reference
    cols = np.zeros(grid * grid, dtype=COLLIDER_DTYPE)
    cols[:] = collider_desc(half_extents=(0.5 * cell, 0.25, 0.5 * cell))
    t = np.zeros((grid * grid, 3), np.float32)
    t[:, 0] = np.repeat(xs, grid)                                       # i outer, j inner: the reference's insertion order
    t[:, 2] = np.tile(xs, grid)
    cols["translation"] = t
    s.colliders = list(cols)
    s.collider_parents = [-1] * (grid * grid)
    return s


def large_world_drop(idx: int, grid: int = 1000, cell: float = 10.0, spheres: int = 100):
    """translation of the idx-th dropped sphere (b3d_large_world.rs:46-66: a coarse side x side grid over the inner 80 % of the floor,
    y = 1.5; one sphere of radius 0.5 every 5 steps)"""
    side = 1
    while side * side < spheres:
        side += 1
    half_span = _f(0.5) * _f(cell) * _f(grid)
    gi, gj = idx % side, idx // side
    inset = _f(0.1) * _f(2.0) * half_span
    usable = _f(2.0) * half_span - _f(2.0) * inset
    x = -half_span + inset + (_f(gi) + _f(0.5)) * (usable / _f(side))
    z = -half_span + inset + (_f(gj) + _f(0.5)) * (usable / _f(side))
    return (float(x), 1.5, float(z))


def many_pyramids(rows: int = 14, cols: int = 14, base_count: int = 10, col_range=None, pyramids=None) -> Scene:
    """b3d_many_pyramids.rs:36-64.  ``col_range=(lo, hi)`` keeps only pyramid columns lo..hi-1, ``pyramids`` (a boolean mask over
    the row-major pyramid index r * cols + c) only the selected islands: the multi-GPU island shards (the ground is replicated)."""
    s = Scene(name=f"b3d_many_pyramids_{rows}x{cols}")
    extent = _f(0.5)
    ground_extent = extent * _f(cols) * (_f(base_count) + _f(1.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -1.0, 0.0))
    s.add_collider(g, half_extents=(ground_extent, 1.0, ground_extent))
    base_width = _f(2.0) * extent * _f(base_count)
    base_z = -ground_extent + _f(2.0) * extent
    delta_z = _f(2.0) * (ground_extent - _f(2.0) * extent) / (_f(rows) - _f(1.0)) if rows > 1 else _f(0.0)
    lo, hi = col_range if col_range is not None else (0, cols)
    for _r in range(rows):
        for j in range(cols):
            center_x = -ground_extent + _f(j) * (base_width + _f(2.0) * extent) + _f(2.0) * extent
            if lo <= j < hi and (pyramids is None or pyramids[_r * cols + j]):
                _small_pyramid(s, base_count, extent, center_x, base_z)
        base_z = base_z + delta_z
    return s


def pyramid10() -> Scene:
    """C1 plumbing scene: one 10-base pyramid."""
    s = many_pyramids(rows=1, cols=1)
    s.name = "pyramid10"
    return s


def large_pyramid(base_count: int = 200) -> Scene:
    """b3d_large_pyramid.rs:15-36."""
    s = Scene(name=f"b3d_large_pyramid_{base_count}")
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -1.0, 0.0))
    s.add_collider(g, half_extents=(400.0, 1.0, 400.0))
    extent = _f(0.5)
    for i in range(base_count):
        y = (_f(2.0) * _f(i) + _f(1.0)) * extent
        for j in range(i, base_count):
            x = (_f(i) + _f(1.0)) * extent + _f(2.0) * _f(j - i) * extent - _f(base_count) * extent
            b = s.add_body(translation=(x, y, 0.0))
            s.add_collider(b, half_extents=(extent, extent, extent), density=100.0)
    return s


def joint_grid(n: int = 100) -> Scene:
    """b3d_joint_grid.rs:17-53: n x n balls (r 0.4), spherical joints, row i == 0 fixed."""
    s = Scene(name=f"b3d_joint_grid_{n}")
    ids = [-1] * (n * n)
    index = 0
    for k in range(n):
        for i in range(n):
            b = s.add_body(body_type=BODY_FIXED if i == 0 else BODY_DYNAMIC, translation=(_f(k), -_f(i), 0.0))
            s.add_collider(b, shape=SHAPE_BALL, half_extents=(0.4, 0.0, 0.0), density=1.0)
            if i > 0:
                s.add_joint(ids[index - 1], b, (0.0, -0.5, 0.0), (0.0, 0.5, 0.0))
            if k > 0:
                s.add_joint(ids[index - n], b, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0))
            ids[index] = b
            index += 1
    return s


def joint_net(n: int = 32) -> Scene:
    """The pinned net of spherical joints of crates/rapier3d/tests/joint_stability.rs:105-175
    (`joint_net_remains_stable`): n x n balls (r 0.4, spacing 1), row 0 pinned every 4th ball and at
    the last one, default gravity (0, -9.81, 0)."""
    s = Scene(name=f"joint_net_{n}", gravity=(0.0, -9.81, 0.0))
    h = [-1] * (n * n)
    for i in range(n):
        for j in range(n):
            fixed = i == 0 and (j % 4 == 0 or j == n - 1)
            b = s.add_body(body_type=BODY_FIXED if fixed else BODY_DYNAMIC, translation=(_f(j), -_f(i), 0.0))
            s.add_collider(b, shape=SHAPE_BALL, half_extents=(0.4, 0.0, 0.0), density=1.0)
            h[i * n + j] = b
    for i in range(n):
        for j in range(n):
            if i > 0:
                s.add_joint(h[(i - 1) * n + j], h[i * n + j], (0.0, -0.5, 0.0), (0.0, 0.5, 0.0))
            if j > 0:
                s.add_joint(h[i * n + j - 1], h[i * n + j], (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0))
    return s


def joint_chain(n: int = 8, with_boxes: bool = False) -> Scene:
    """A pendulum chain of n bodies hanging from a fixed one by spherical joints, released
    horizontally (not a reference scene): exercises joints with large motion and, with boxes next to a
    wall slab, joints + contacts on the same bodies."""
    s = Scene(name=f"joint_chain_{n}{'_boxes' if with_boxes else ''}", gravity=(0.0, -9.81, 0.0))
    prev = s.add_body(body_type=BODY_FIXED, translation=(0.0, 5.0, 0.0))
    s.add_collider(prev, shape=SHAPE_BALL, half_extents=(0.2, 0.0, 0.0))
    if with_boxes:
        g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
        s.add_collider(g, half_extents=(20.0, 0.5, 20.0))
    for i in range(n):
        b = s.add_body(translation=(_f(i + 1), 5.0, 0.0))
        if with_boxes:
            s.add_collider(b, half_extents=(0.3, 0.3, 0.3), density=2.0)
        else:
            s.add_collider(b, shape=SHAPE_BALL, half_extents=(0.3, 0.0, 0.0), density=2.0)
        s.add_joint(prev, b, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0))
        prev = b
    return s


def box_stack(height: int = 3, gap: float = 0.0) -> Scene:
    """Small cube stack on a slab (the 3-cube stack of test_staged.rs:86-148)."""
    s = Scene(name=f"box_stack_{height}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(10.0, 0.5, 10.0))
    for i in range(height):
        b = s.add_body(translation=(0.0, 0.5 + i * (1.0 + gap), 0.0))
        s.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    return s


def tumble(n: int = 64, seed: int = 7, balls: bool = True) -> Scene:
    """Seeded dynamic test scene (not a reference scene): rotated cuboids (and balls) with initial
    velocities dropped on a slab inside a shallow box of walls — exercises full narrow-phase updates,
    edge/edge SAT axes, 5-8 point manifold reduction, ball cases, pair creation/deletion and
    begin/end-touch recolouring."""
    rng = np.random.default_rng(seed)
    s = Scene(name=f"tumble_{n}_{seed}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(12.0, 0.5, 12.0))
    for sx, sz, hx, hz in ((6.0, 0.0, 0.25, 6.0), (-6.0, 0.0, 0.25, 6.0), (0.0, 6.0, 6.0, 0.25), (0.0, -6.0, 6.0, 0.25)):
        wb = s.add_body(body_type=BODY_FIXED, translation=(sx, 1.0, sz))
        s.add_collider(wb, half_extents=(hx, 1.0, hz))
    side = int(np.ceil(n ** (1.0 / 3.0)))
    k = 0
    for iy in range(side * 2):
        for ix in range(side):
            for iz in range(side):
                if k >= n:
                    break
                q = rng.normal(size=4).astype(np.float32)
                q /= np.linalg.norm(q)
                pos = (np.float32(1.3 * (ix - side / 2) + 0.1 * rng.random()), np.float32(1.0 + 1.4 * iy),
                       np.float32(1.3 * (iz - side / 2) + 0.1 * rng.random()))
                lv = (rng.normal(size=3) * 1.5).astype(np.float32)
                av = (rng.normal(size=3) * 2.0).astype(np.float32)
                b = s.add_body(translation=pos, rotation=tuple(q), linvel=tuple(lv), angvel=tuple(av),
                               linear_damping=0.05 if k % 5 == 0 else 0.0, angular_damping=0.1 if k % 7 == 0 else 0.0)
                if balls and k % 3 == 2:
                    s.add_collider(b, shape=SHAPE_BALL, half_extents=(np.float32(0.3 + 0.2 * rng.random()), 0, 0),
                                   restitution=0.6 if k % 2 == 0 else 0.0, friction=0.4)
                else:
                    he = (0.25 + 0.35 * rng.random(size=3)).astype(np.float32)
                    s.add_collider(b, half_extents=tuple(he), friction=np.float32(0.2 + 0.6 * rng.random()),
                                   restitution=0.3 if k % 4 == 0 else 0.0, density=np.float32(0.5 + 2.0 * rng.random()))
                k += 1
    return s


def sleep_impact(height: float = 12.0, stack: int = 3) -> Scene:
    """Sleeping test scene (not a reference scene): a cube stack that falls asleep long before a small cube dropped
    from ``height`` reaches it; the impact wakes the whole island (contacts.rs:333-351), then everything
    falls asleep again.  A second, untouched stack nearby must stay asleep throughout."""
    s = Scene(name=f"sleep_impact_{int(height)}_{stack}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(10.0, 0.5, 10.0))
    for x in (0.0, 4.0):
        for i in range(stack):
            b = s.add_body(translation=(x, 0.5 + i * 1.0, 0.0), can_sleep=1)
            s.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    drop = s.add_body(translation=(0.15, height, 0.1), rotation=(0.0, 0.19866933, 0.0, 0.98006658), can_sleep=1)
    s.add_collider(drop, half_extents=(0.3, 0.3, 0.3), density=2.0)
    return s


def kinematic_platform(position_based: bool = False, boxes: int = 3) -> Scene:
    """Kinematic test scene (not a reference scene): a slab-sized kinematic platform carrying a small cube stack
    over a fixed floor, plus a free dynamic cube on the floor next to it.  Velocity-based: the platform is given
    a velocity; position-based: the test feeds ``set_next_kinematic_position`` every step."""
    s = Scene(name=f"kinematic_platform_{'pos' if position_based else 'vel'}_{boxes}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(20.0, 0.5, 20.0))
    p = s.add_body(body_type=BODY_KINEMATIC_POSITION if position_based else BODY_KINEMATIC_VELOCITY, translation=(0.0, 1.0, 0.0),
                   linvel=(0.0, 0.0, 0.0) if position_based else (0.6, 0.15, 0.0), angvel=(0.0, 0.0, 0.0) if position_based else (0.0, 0.2, 0.0))
    s.add_collider(p, half_extents=(3.0, 0.25, 3.0))
    for i in range(boxes):
        b = s.add_body(translation=(0.2 * i, 1.25 + 0.5 + i * 1.0, 0.1 * i))
        s.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    f = s.add_body(translation=(5.0, 0.5, 0.0))
    s.add_collider(f, half_extents=(0.5, 0.5, 0.5))
    return s


# local joint frame whose X axis is the body's Z axis (RevoluteJointBuilder::new(Vector::Z)): rotation by -90 deg about Y
AXIS_Z_BASIS = (0.0, -0.70710678, 0.0, 0.70710678)


def jointed_pairs(n: int = 1) -> Scene:
    """The jointed pair of test_staged.rs:86-148 (a revolute joint about Z between two elevated cubes, next to the
    3-cube stack), ``n`` times; plus a door on a revolute hinge about Y on a fixed post and a cube welded to another
    by a fixed joint (all six axes locked).  Exercises locked angular axes (JointConstraintHelper::lock_angular)."""
    s = Scene(name=f"jointed_pairs_{n}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(30.0, 0.5, 30.0))
    for i in range(3):
        b = s.add_body(translation=(0.0, 0.5 + i * 1.0, 0.0))
        s.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    for k in range(n):
        a = s.add_body(translation=(5.0 + 4.0 * k, 3.0, 0.0))
        s.add_collider(a, half_extents=(0.5, 0.5, 0.5))
        b = s.add_body(translation=(6.5 + 4.0 * k, 3.0, 0.0), linvel=(0.0, 0.0, 0.3 * (k % 3)))
        s.add_collider(b, half_extents=(0.5, 0.5, 0.5))
        s.add_joint(a, b, (0.75, 0.0, 0.0), (-0.75, 0.0, 0.0), locked_axes=LOCK_REVOLUTE, basis1=AXIS_Z_BASIS, basis2=AXIS_Z_BASIS)
    post = s.add_body(body_type=BODY_FIXED, translation=(-5.0, 1.5, 0.0))
    s.add_collider(post, half_extents=(0.1, 1.5, 0.1))
    door = s.add_body(translation=(-4.0, 1.5, 0.0), angvel=(0.0, 1.5, 0.0), linvel=(0.0, 0.0, -1.5))
    s.add_collider(door, half_extents=(0.8, 1.0, 0.05), density=2.0)
    axis_y = (0.0, 0.0, 0.70710678, 0.70710678)   # frame X axis = body Y axis
    s.add_joint(post, door, (0.0, 0.0, 0.0), (-1.0, 0.0, 0.0), locked_axes=LOCK_REVOLUTE, basis1=axis_y, basis2=axis_y)
    w1 = s.add_body(translation=(-10.0, 4.0, 0.0), angvel=(0.5, 0.2, 0.0))
    s.add_collider(w1, half_extents=(0.5, 0.5, 0.5))
    w2 = s.add_body(translation=(-8.8, 4.0, 0.0))
    s.add_collider(w2, half_extents=(0.5, 0.5, 0.5), density=3.0)
    s.add_joint(w1, w2, (0.6, 0.0, 0.0), (-0.6, 0.0, 0.0), locked_axes=LOCK_ALL)
    return s


def compound_bodies(n: int = 12) -> Scene:
    """Compound-body test scene (not a reference scene): hammers (a handle cuboid + an offset, rotated head cuboid +
    a ball at the other end) and L-shapes dropped with spin onto a slab.  Exercises summed MassProperties (offset centre
    of mass, non-trivial principal frame), several colliders per body and pairs that share a body."""
    s = Scene(name=f"compound_bodies_{n}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(15.0, 0.5, 15.0))
    for i in range(n):
        x, z = _f(2.5 * (i % 4) - 3.75), _f(2.5 * (i // 4) - 2.5)
        b = s.add_body(translation=(x, _f(1.5 + 0.8 * (i % 3)), z), angvel=(_f(0.5 * (i % 3)), _f(0.3 * i), _f(-0.4 * (i % 2))),
                       linvel=(_f(0.2 * (i % 2)), 0.0, 0.0))
        if i % 2 == 0:   # hammer
            s.add_collider(b, half_extents=(0.6, 0.1, 0.1), density=1.0)
            s.add_collider(b, half_extents=(0.15, 0.3, 0.2), translation=(0.7, 0.0, 0.0), rotation=(0.0, 0.0, 0.38268343, 0.92387953), density=4.0)
            s.add_collider(b, shape=SHAPE_BALL, half_extents=(0.15, 0.0, 0.0), translation=(-0.7, 0.0, 0.0), density=0.5)
        else:            # L-shape
            s.add_collider(b, half_extents=(0.5, 0.15, 0.15), translation=(0.5, 0.0, 0.0), density=2.0)
            s.add_collider(b, half_extents=(0.15, 0.5, 0.15), translation=(0.0, 0.5, 0.0), density=2.0)
    return s


def locked_axes_scene() -> Scene:
    """LockedAxes test scene (not a reference scene): spinning cubes dropped on a slab and on each other with different
    translation / rotation locks (RigidBodyBuilder::locked_axes)."""
    s = Scene(name="locked_axes", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(10.0, 0.5, 10.0))
    locks = [0x00, 0x38, 0x04 | 0x08 | 0x10, 0x3F, 0x07, 0x10, 0x01, 0x22]
    for i, la in enumerate(locks):
        # a locked degree of freedom keeps whatever velocity it starts with (only its inverse mass is zeroed): start those at rest
        lv = (0.0 if la & 1 else 0.5, 0.0, 0.0 if la & 4 else _f(0.3 * (i % 3)))
        av = (0.0 if la & 8 else 1.0, 0.0 if la & 16 else 0.5, 0.0 if la & 32 else -0.7)
        b = s.add_body(translation=(_f(1.2 * (i % 4) - 1.8), _f(1.0 + 1.3 * (i // 4)), _f(0.2 * (i % 2))), rotation=(0.1, 0.05, 0.0, 0.9937304),
                       linvel=lv, angvel=av, locked_axes=la)
        s.add_collider(b, half_extents=(0.4, 0.5, 0.3), density=1.5)
    top = s.add_body(translation=(-1.6, 4.0, 0.1), angvel=(0.0, 0.0, 2.0))   # falls onto the locked ones
    s.add_collider(top, half_extents=(1.5, 0.2, 0.5))
    return s


def overlapping_chain(n: int = 6, contacts_enabled: int = 0) -> Scene:
    """A chain of cuboid links whose neighbours OVERLAP at the joints (like the limbs of a ragdoll), joined by spherical
    joints with ``contacts_enabled = false`` (GenericJoint::contacts_enabled, pair_update.rs:191-201), dropped on a slab."""
    s = Scene(name=f"overlapping_chain_{n}_{contacts_enabled}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(20.0, 0.5, 20.0))
    prev = None
    for i in range(n):
        b = s.add_body(translation=(_f(i) * 1.0, 2.0, 0.0), angvel=(0.0, 0.0, _f(0.2 * (i % 2))))
        s.add_collider(b, half_extents=(0.6, 0.15, 0.15), density=2.0)
        if prev is not None:
            s.add_joint(prev, b, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0), contacts_enabled=contacts_enabled)
        prev = b
    return s


def kinematic_crane(n: int = 5) -> Scene:
    """A chain of balls hanging by spherical joints from a velocity-based kinematic trolley (a joint with a kinematic side:
    a solver body with zero inverse mass), above a slab with a few cubes the chain sweeps through."""
    s = Scene(name=f"kinematic_crane_{n}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(20.0, 0.5, 20.0))
    trolley = s.add_body(body_type=BODY_KINEMATIC_VELOCITY, translation=(-3.0, 4.0, 0.0), linvel=(1.0, 0.0, 0.0))
    s.add_collider(trolley, half_extents=(0.3, 0.1, 0.3))
    prev = trolley
    for i in range(n):
        b = s.add_body(translation=(-3.0, _f(4.0 - 0.7 * (i + 1)), 0.0), can_sleep=1)
        s.add_collider(b, shape=SHAPE_BALL, half_extents=(0.25, 0.0, 0.0), density=3.0)
        s.add_joint(prev, b, (0.0, -0.35, 0.0), (0.0, 0.35, 0.0))
        prev = b
    for k in range(3):
        c = s.add_body(translation=(_f(0.5 + 1.2 * k), 0.5, 0.0), can_sleep=1)
        s.add_collider(c, half_extents=(0.3, 0.5, 0.3))
    return s


def limited_joints() -> Scene:
    """Joint-limit test scene (not a reference scene): a weight on a vertical prismatic slider with limits (it drops to
    the lower stop), a door on a revolute hinge with limits +-0.6 rad pushed into its stop, and a free-swinging limited
    pendulum; next to a small stack."""
    s = Scene(name="limited_joints", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(20.0, 0.5, 20.0))
    axis_y = (0.0, 0.0, 0.70710678, 0.70710678)   # frame X axis = body Y axis
    rail = s.add_body(body_type=BODY_FIXED, translation=(0.0, 5.0, 0.0))
    s.add_collider(rail, half_extents=(0.1, 0.1, 0.1))
    w = s.add_body(translation=(0.0, 5.0, 0.0))
    s.add_collider(w, half_extents=(0.3, 0.3, 0.3), density=2.0)
    s.add_joint(rail, w, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), locked_axes=LOCK_PRISMATIC, basis1=axis_y, basis2=axis_y, limits={0: (-1.5, 0.5)})
    post = s.add_body(body_type=BODY_FIXED, translation=(-5.0, 1.5, 0.0))
    s.add_collider(post, half_extents=(0.1, 1.5, 0.1))
    door = s.add_body(translation=(-4.0, 1.5, 0.0), angvel=(0.0, 2.5, 0.0), linvel=(0.0, 0.0, -2.5))
    s.add_collider(door, half_extents=(0.8, 1.0, 0.05), density=2.0)
    s.add_joint(post, door, (0.0, 0.0, 0.0), (-1.0, 0.0, 0.0), locked_axes=LOCK_REVOLUTE, basis1=axis_y, basis2=axis_y, limits={3: (-0.6, 0.6)})
    pivot = s.add_body(body_type=BODY_FIXED, translation=(5.0, 4.0, 0.0))
    s.add_collider(pivot, shape=SHAPE_BALL, half_extents=(0.1, 0.0, 0.0))
    bob = s.add_body(translation=(6.5, 4.0, 0.0))
    s.add_collider(bob, shape=SHAPE_BALL, half_extents=(0.3, 0.0, 0.0), density=3.0)
    s.add_joint(pivot, bob, (0.0, 0.0, 0.0), (-1.5, 0.0, 0.0), locked_axes=LOCK_REVOLUTE, basis1=AXIS_Z_BASIS, basis2=AXIS_Z_BASIS, limits={3: (-0.8, 0.3)})
    for i in range(2):
        b = s.add_body(translation=(2.0, 0.5 + i, 0.0))
        s.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    return s


def motorised_joints() -> Scene:
    """Joint-motor test scene (not a reference scene): a wheel spun by a velocity motor on a fixed axle, a lift on a limited
    prismatic rail driven by a force-based position motor with a force cap, an arm on a spherical joint held by three angular
    position motors, and a two-wheeled cart whose wheels (revolute joints between dynamic bodies) are driven along the ground."""
    s = Scene(name="motorised_joints", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(30.0, 0.5, 30.0))
    axis_y = (0.0, 0.0, 0.70710678, 0.70710678)   # frame X axis = body Y axis
    axle = s.add_body(body_type=BODY_FIXED, translation=(-6.0, 3.0, 0.0))
    wheel = s.add_body(translation=(-6.0, 3.0, 0.0))
    s.add_collider(wheel, half_extents=(0.8, 0.8, 0.1), density=2.0)
    s.add_joint(axle, wheel, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), locked_axes=LOCK_REVOLUTE, basis1=AXIS_Z_BASIS, basis2=AXIS_Z_BASIS,
                motors={3: dict(target_vel=3.0, damping=5.0)})
    rail = s.add_body(body_type=BODY_FIXED, translation=(0.0, 4.0, 0.0))
    lift = s.add_body(translation=(0.0, 4.0, 0.0))
    s.add_collider(lift, half_extents=(0.4, 0.2, 0.4), density=3.0)
    s.add_joint(rail, lift, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), locked_axes=LOCK_PRISMATIC, basis1=axis_y, basis2=axis_y, limits={0: (-2.0, 0.4)},
                motors={0: dict(target_pos=-0.5, stiffness=400.0, damping=40.0, max_force=60.0, model=MOTOR_FORCE_BASED)})
    pivot = s.add_body(body_type=BODY_FIXED, translation=(6.0, 4.0, 0.0))
    arm = s.add_body(translation=(7.0, 4.0, 0.0), angvel=(0.5, 1.0, -0.5))
    s.add_collider(arm, half_extents=(0.8, 0.1, 0.15), density=1.5)
    s.add_joint(pivot, arm, (0.0, 0.0, 0.0), (-1.0, 0.0, 0.0), locked_axes=LOCK_LIN,
                motors={3: dict(target_pos=0.3, stiffness=80.0, damping=8.0), 4: dict(target_pos=-0.2, stiffness=80.0, damping=8.0),
                        5: dict(target_pos=0.5, stiffness=80.0, damping=8.0)})
    cart = s.add_body(translation=(0.0, 0.7, 6.0))
    s.add_collider(cart, half_extents=(1.0, 0.15, 0.4), density=1.0)
    for sx in (-0.8, 0.8):
        wh = s.add_body(translation=(sx, 0.4, 6.0))
        s.add_collider(wh, shape=SHAPE_BALL, half_extents=(0.4, 0.0, 0.0), density=1.0, friction=1.0)
        s.add_joint(cart, wh, (sx, -0.3, 0.0), (0.0, 0.0, 0.0), locked_axes=LOCK_REVOLUTE, basis1=AXIS_Z_BASIS, basis2=AXIS_Z_BASIS, contacts_enabled=0,
                    motors={3: dict(target_vel=-4.0, damping=30.0, max_force=5.0)})
    for i in range(2):
        b = s.add_body(translation=(3.0, 0.5 + i, 0.0))
        s.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    return s


def capsules(n: int = 6) -> Scene:
    """Capsule test scene (not a reference scene): capsules lying along X and Z dropped in a criss-cross pile (capsule-capsule),
    tilted capsules falling on the slab and on a box (cuboid-capsule, both collider orders), balls dropped on capsules
    (capsule-ball), a compound dumbbell of a capsule and two balls."""
    s = Scene(name=f"capsules_{n}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(30.0, 0.5, 30.0))
    for i in range(n):                       # criss-cross pile
        b = s.add_body(translation=(0.05 * i, 0.4 + 0.85 * i, -0.03 * i))
        s.add_collider(b, shape=SHAPE_CAPSULE, half_extents=(0.9, 0.35, 0.0 if i % 2 == 0 else 2.0))
    box = s.add_body(translation=(6.0, 0.5, 0.0))
    s.add_collider(box, half_extents=(1.0, 0.5, 1.0))
    for i in range(3):                       # tilted capsules: on the box (box first in the pair), on the slab, on each other
        b = s.add_body(translation=(6.0 + 0.3 * i, 2.0 + 1.4 * i, 0.1 * i), rotation=(0.0, 0.0, 0.25881905, 0.96592583), angvel=(0.0, 0.5, 0.0))
        s.add_collider(b, shape=SHAPE_CAPSULE, half_extents=(0.6, 0.3, 1.0))
    cap0 = s.add_body(translation=(-6.0, 0.4, 0.0))    # a capsule inserted BEFORE a box: capsule is collider 1 of that pair
    s.add_collider(cap0, shape=SHAPE_CAPSULE, half_extents=(1.2, 0.4, 0.0))
    b = s.add_body(translation=(-6.0, 1.5, 0.0), rotation=(0.0, 0.38268343, 0.0, 0.92387953))
    s.add_collider(b, half_extents=(0.5, 0.4, 0.5))
    for i in range(2):                       # balls on capsules, both collider orders
        b = s.add_body(translation=(-6.2 + 0.5 * i, 3.0 + i, 0.1))
        s.add_collider(b, shape=SHAPE_BALL, half_extents=(0.3, 0.0, 0.0))
    ball0 = s.add_body(translation=(0.0, 0.5, 6.0))
    s.add_collider(ball0, shape=SHAPE_BALL, half_extents=(0.5, 0.0, 0.0))
    b = s.add_body(translation=(0.1, 2.0, 6.0))
    s.add_collider(b, shape=SHAPE_CAPSULE, half_extents=(0.7, 0.25, 2.0))
    d = s.add_body(translation=(3.0, 3.0, 6.0), angvel=(1.0, 0.0, 2.0))     # dumbbell
    s.add_collider(d, shape=SHAPE_CAPSULE, half_extents=(0.8, 0.15, 0.0))
    s.add_collider(d, shape=SHAPE_BALL, half_extents=(0.35, 0.0, 0.0), translation=(0.95, 0.0, 0.0))
    s.add_collider(d, shape=SHAPE_BALL, half_extents=(0.35, 0.0, 0.0), translation=(-0.95, 0.0, 0.0))
    return s


def reference_pile(nx: int = 12, ny: int = 3, nz: int = 12, chain: bool = True, sleep: bool = True) -> Scene:
    """The stress scene of the reference's simd_backend_determinism.rs:61-139 (nx, ny, nz = 12, 3, 12 + a 4-ball spherical-joint
    chain) and, with (14, 2, 14) and no chain, the pile of parallel_path_parity.rs:113-134: a jittered grid of unit cubes that
    settles asymmetrically; bodies keep the builder's can_sleep default (true)."""
    # `Vector::Y * -9.81` = (-0.0, -9.81, -0.0): the zeros keep their sign (they vanish in `user_force + gravity * mass`)
    s = Scene(name=f"reference_pile_{nx}x{ny}x{nz}", gravity=(-0.0, -9.81, -0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(30.0 if not chain else 20.0, 0.5, 30.0 if not chain else 20.0))
    f32 = np.float32
    off = f32(nx) * f32(0.5)
    for i in range(nx):
        for j in range(ny):
            for k in range(nz):
                jitter = np.fmod(f32(i) * f32(0.013) + f32(k) * f32(0.017), f32(0.05))
                b = s.add_body(translation=(float(f32(i) * f32(1.05) - off + jitter), float(f32(j) * f32(1.05) + f32(0.55)),
                                            float(f32(k) * f32(1.05) - off - jitter)), can_sleep=1 if sleep else 0)
                s.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    if chain:
        prev = s.add_body(body_type=BODY_FIXED, translation=(0.0, 8.0, 0.0))
        for i in range(4):
            # `0.6 * (i + 1) as Real` is an f32 product (simd_backend_determinism.rs:108-112): 0x3fe66667 for i = 2, not the
            # narrowed f64 product 0x3fe66666
            b = s.add_body(translation=(float(f32(0.6) * f32(i + 1)), 8.0, 0.0), can_sleep=1 if sleep else 0)
            s.add_collider(b, shape=SHAPE_BALL, half_extents=(0.25, 0.0, 0.0))
            # `Vector::X * -0.3` = (-0.3, -0.0, -0.0): the zeros keep their sign
            s.add_joint(prev, b, (0.3, 0.0, 0.0), (-0.3, -0.0, -0.0), locked_axes=LOCK_LIN)
            prev = b
    return s


def reference_cluster(seed: int, height: float = 6.0):
    """spawn_cluster of parallel_path_parity.rs:80-100: 12 kicked unit cubes as (body_desc, collider_desc) pairs"""
    f32 = np.float32
    out = []
    for i in range(12):
        a = f32(seed * 7 + i * 13) * f32(0.011)
        b = f32(seed * 11 + i * 5) * f32(0.017)
        body = body_desc(translation=(float(np.fmod(f32(i), f32(4.0)) * f32(1.1) - f32(2.2) + a), float(f32(height) + f32(i // 4) * f32(1.1)),
                                      float(np.fmod(b, f32(3.0)) - f32(1.5))),
                         linvel=(float(np.fmod(a, f32(1.5)) - f32(0.75)), 0.0, float(np.fmod(b, f32(1.5)) - f32(0.75))), can_sleep=1)
        out.append((body, collider_desc(half_extents=(0.5, 0.5, 0.5))))
    return out


def solve_groups_scene() -> Scene:
    """Substep solve-groups (RigidBody::additional_solver_iterations; not a reference scene): a default-cadence stack, a jointed chain
    whose heavy end ball carries 6 extra substeps (the joint lifts the whole chain), a 150:1 stack with 12 extra substeps on the heavy
    cube riding a velocity-based kinematic platform (lifted to that group) — three groups with 4, 10 and 16 substeps."""
    s = Scene(name="solve_groups", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(100.0, 0.5, 100.0))
    for i in range(4):
        b = s.add_body(translation=(0.0, 0.5 + i * 1.0, 0.0))
        s.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    prev = s.add_body(body_type=BODY_FIXED, translation=(30.0, 10.0, 0.0))
    for i in range(5):
        b = s.add_body(translation=(30.0 + (i + 1.0), 10.0, 0.0), additional_solver_iterations=6 if i == 4 else 0)
        s.add_collider(b, shape=SHAPE_BALL, half_extents=(0.3, 0.0, 0.0), density=50.0 if i == 4 else 1.0)
        s.add_joint(prev, b, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0), locked_axes=LOCK_LIN)
        prev = b
    plat = s.add_body(body_type=BODY_KINEMATIC_VELOCITY, translation=(-20.0, 2.0, 0.0), linvel=(0.3, 0.0, 0.0))
    s.add_collider(plat, half_extents=(2.0, 0.25, 2.0))
    lo = s.add_body(translation=(-20.0, 2.75, 0.0))
    s.add_collider(lo, half_extents=(0.5, 0.5, 0.5), density=1.0)
    hi = s.add_body(translation=(-20.0, 3.75, 0.0), additional_solver_iterations=12)
    s.add_collider(hi, half_extents=(0.5, 0.5, 0.5), density=150.0, restitution=0.3)
    return s


def sensor_scene() -> Scene:
    """Sensors (ColliderBuilder::sensor; not a reference scene): a cuboid trigger volume over the floor that a capsule, a box and a
    ball fall through, a ball sensor riding a falling body, and a sensor part on a compound body."""
    s = Scene(name="sensors", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(100.0, 0.5, 100.0))
    trig = s.add_body(body_type=BODY_FIXED, translation=(0.0, 3.0, 0.0))
    s.add_collider(trig, half_extents=(4.0, 0.5, 4.0), active_events=ACTIVE_EVENTS_COLLISION, sensor=1)
    cap = s.add_body(translation=(-2.0, 6.0, 0.0), rotation=(0.0, 0.0, 0.3428978, 0.9393727))
    s.add_collider(cap, shape=SHAPE_CAPSULE, half_extents=(0.5, 0.25, 1.0))
    box = s.add_body(translation=(0.0, 6.5, 0.0), rotation=(0.1435722, 0.1060205, 0.0342708, 0.9833474))
    s.add_collider(box, half_extents=(0.3, 0.3, 0.3))
    ball = s.add_body(translation=(2.0, 7.0, 0.0))
    s.add_collider(ball, shape=SHAPE_BALL, half_extents=(0.3, 0.0, 0.0))
    rider = s.add_body(translation=(0.5, 9.0, 0.5), gravity_scale=0.5)
    s.add_collider(rider, half_extents=(0.25, 0.25, 0.25))
    s.add_collider(rider, shape=SHAPE_BALL, half_extents=(0.9, 0.0, 0.0), density=0.0, sensor=1, active_events=ACTIVE_EVENTS_COLLISION)
    return s



def halfspace_scene(n_side: int = 4) -> Scene:
    """Half-spaces (ColliderBuilder::halfspace; not a reference scene): a ground plane without a parent body, a slanted plane on
    a fixed body inserted AFTER the dynamic bodies (so it is the second collider of its pairs), a slowly rising plane on a
    velocity-based kinematic body, and a grid of tumbling cuboids, balls and capsules (all three axes) dropped between them."""
    s = Scene(name="halfspaces", gravity=(0.0, -9.81, 0.0))
    s.add_collider(-1, shape=SHAPE_HALFSPACE, half_extents=(0.0, 1.0, 0.0), friction=0.7, active_events=ACTIVE_EVENTS_COLLISION)
    lift = s.add_body(body_type=BODY_KINEMATIC_VELOCITY, translation=(0.0, -0.6, 0.0), linvel=(0.0, 0.15, 0.0))
    s.add_collider(lift, shape=SHAPE_HALFSPACE, half_extents=(0.0, 1.0, 0.0), translation=(0.0, 0.0, 0.0), friction=0.4,
                   memberships=0x4, filter=0x4)                   # only the colliders with bit 2 ride it
    rng = np.random.default_rng(7)
    k = 0
    for i in range(n_side):
        for j in range(n_side):
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            b = s.add_body(translation=(1.4 * i - 2.0, 1.0 + 0.5 * ((i + j) % 3), 1.4 * j - 2.0), rotation=tuple(float(x) for x in q),
                           angvel=tuple(float(x) for x in rng.uniform(-2, 2, size=3)))
            kind = k % 4
            grp = dict(memberships=0x3, filter=0x3) if k % 5 else dict(memberships=0x7, filter=0x7)
            if kind == 0:
                s.add_collider(b, half_extents=(0.3, 0.2, 0.4), **grp)
            elif kind == 1:
                s.add_collider(b, shape=SHAPE_BALL, half_extents=(0.3, 0.0, 0.0), restitution=0.4, **grp)
            else:
                s.add_collider(b, shape=SHAPE_CAPSULE, half_extents=(0.35, 0.2, float(k % 3)), **grp)
            k += 1
    ramp = s.add_body(body_type=BODY_FIXED, translation=(3.0, 0.0, 0.0))
    s.add_collider(ramp, shape=SHAPE_HALFSPACE, half_extents=(-0.6, 0.8, 0.0), friction=0.2)
    return s


def convex_clutter(n: int = 40, seed: int = 3, ground: str = "cuboid", compound: bool = True) -> Scene:
    """Seeded test scene (not a reference scene) for the support-mapped shapes: cylinders and cones (ColliderBuilder::cylinder / cone,
    collider.rs:770, :789) tumbling among cuboids, balls and capsules on a slab (`ground` = "cuboid"), a wide fixed disc ("cylinder") or
    a half-space ("halfspace") inside four walls — every pair the parry dispatcher sends through contact_manifold_pfm_pfm, the
    ball and the half-space arms, plus one compound body carrying a cylinder and a cone."""
    rng = np.random.default_rng(seed)
    s = Scene(name=f"convex_clutter_{n}_{seed}_{ground}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    if ground == "cylinder":
        s.add_collider(g, shape=SHAPE_CYLINDER, half_extents=(0.5, 9.0, 0.0))
    elif ground == "halfspace":
        s.add_collider(g, shape=SHAPE_HALFSPACE, half_extents=(0.0, 1.0, 0.0), translation=(0.0, 0.5, 0.0))
    else:
        s.add_collider(g, half_extents=(9.0, 0.5, 9.0))
    for sx, sz, hx, hz in ((4.5, 0.0, 0.25, 4.5), (-4.5, 0.0, 0.25, 4.5), (0.0, 4.5, 4.5, 0.25), (0.0, -4.5, 4.5, 0.25)):
        wb = s.add_body(body_type=BODY_FIXED, translation=(sx, 1.0, sz))
        s.add_collider(wb, half_extents=(hx, 1.0, hz))
    pillar = s.add_body(body_type=BODY_FIXED, translation=(0.0, 0.6, 0.0))
    s.add_collider(pillar, shape=SHAPE_CONE, half_extents=(0.6, 0.8, 0.0))
    side = int(np.ceil(n ** (1.0 / 3.0)))
    k = 0
    for iy in range(side * 2):
        for ix in range(side):
            for iz in range(side):
                if k >= n:
                    break
                q = rng.normal(size=4).astype(np.float32)
                q /= np.linalg.norm(q)
                pos = (np.float32(1.4 * (ix - side / 2) + 0.1 * rng.random()), np.float32(1.8 + 1.5 * iy),
                       np.float32(1.4 * (iz - side / 2) + 0.1 * rng.random()))
                lv = (rng.normal(size=3) * 1.0).astype(np.float32)
                av = (rng.normal(size=3) * 2.0).astype(np.float32)
                b = s.add_body(translation=pos, rotation=tuple(q), linvel=tuple(lv), angvel=tuple(av),
                               angular_damping=0.2 if k % 4 == 0 else 0.0)
                kind = k % 6
                fr = np.float32(0.3 + 0.5 * rng.random())
                if kind in (0, 3):
                    s.add_collider(b, shape=SHAPE_CYLINDER, half_extents=(np.float32(0.2 + 0.3 * rng.random()), np.float32(0.2 + 0.3 * rng.random()), 0.0),
                                   friction=fr, density=np.float32(0.5 + 1.5 * rng.random()))
                elif kind == 1:
                    s.add_collider(b, shape=SHAPE_CONE, half_extents=(np.float32(0.3 + 0.3 * rng.random()), np.float32(0.25 + 0.25 * rng.random()), 0.0),
                                   friction=fr, restitution=0.2 if k % 2 == 0 else 0.0)
                elif kind == 2:
                    s.add_collider(b, half_extents=tuple((0.2 + 0.3 * rng.random(size=3)).astype(np.float32)), friction=fr)
                elif kind == 4:
                    s.add_collider(b, shape=SHAPE_BALL, half_extents=(np.float32(0.25 + 0.2 * rng.random()), 0, 0), friction=fr)
                else:
                    s.add_collider(b, shape=SHAPE_CAPSULE, half_extents=(np.float32(0.3 + 0.2 * rng.random()), np.float32(0.15 + 0.15 * rng.random()), float(k % 3)), friction=fr)
                k += 1
    if compound:  # a mallet: cylinder handle + cone head, off-centre
        b = s.add_body(translation=(2.5, 3.0, -2.5), angvel=(0.5, 1.0, 0.0))
        s.add_collider(b, shape=SHAPE_CYLINDER, half_extents=(0.6, 0.12, 0.0), density=0.8)
        s.add_collider(b, shape=SHAPE_CONE, half_extents=(0.3, 0.35, 0.0), translation=(0.0, 0.85, 0.0), rotation=(0.0, 0.0, 0.70710678, 0.70710678), density=2.0)
    return s


def issue_810_disc() -> Scene:
    """crates/rapier3d/tests/issue_810_cubes_thin_cylinder_tunnel.rs:61-95: twenty 0.1 m cubes dropped from y = 20 onto a thin fixed
    cylinder disc (radius 10, half height 0.05), spread over it on a golden-angle spiral"""
    s = Scene(name="issue_810", gravity=(0.0, -9.81, 0.0))
    disc = s.add_body(body_type=BODY_FIXED, translation=(0.0, -2.0, 0.0))
    s.add_collider(disc, shape=SHAPE_CYLINDER, half_extents=(0.05, 10.0, 0.0))
    for k in range(20):
        r, a = np.float32(k) * np.float32(0.45), np.float32(k) * np.float32(2.399)
        b = s.add_body(translation=(float(r * np.cos(a)), 20.0, float(r * np.sin(a))))
        s.add_collider(b, half_extents=(0.05, 0.05, 0.05))
    return s


def polyhedra_clutter(n: int = 24, seed: int = 2, hulls: bool = True) -> Scene:
    """Seeded test scene (not a reference scene) for convex polyhedra (ColliderBuilder::convex_hull / convex_mesh, collider.rs:1039, :1070):
    random point clouds, prisms, wedges and a box given as a polyhedron, tumbling with cuboids, balls, capsules, cylinders and cones on a
    slab and a half-space ramp inside four walls; several bodies share one registered polyhedron, one compound body carries two, a
    fixed polyhedron stands in the middle.  `hulls`: register the clouds as point sets (convex_hull) — the caller may replace them by
    explicit triangle lists (convex_mesh)."""
    rng = np.random.default_rng(seed)
    s = Scene(name=f"polyhedra_clutter_{n}_{seed}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(9.0, 0.5, 9.0))
    s.add_collider(g, shape=SHAPE_HALFSPACE, half_extents=(-0.5, 0.8660254, 0.0), translation=(3.0, 0.5, 0.0))     # a ramp along +x
    for sx, sz, hx, hz in ((4.5, 0.0, 0.25, 4.5), (-4.5, 0.0, 0.25, 4.5), (0.0, 4.5, 4.5, 0.25), (0.0, -4.5, 4.5, 0.25)):
        wb = s.add_body(body_type=BODY_FIXED, translation=(sx, 1.0, sz))
        s.add_collider(wb, half_extents=(hx, 1.0, hz))
    shapes = []
    shapes.append(s.add_convex_polyhedron(np.array([[x, y, z] for x in (-.35, .35) for y in (-.25, .25) for z in (-.3, .3)], np.float32)))   # a box
    shapes.append(s.add_convex_polyhedron(np.float32([[np.cos(a) * 0.35, h, np.sin(a) * 0.35] for h in (-0.3, 0.3) for a in np.linspace(0, 2 * np.pi, 6)[:-1]])))  # pentagonal prism
    shapes.append(s.add_convex_polyhedron(np.float32([[-0.4, -0.2, -0.3], [0.4, -0.2, -0.3], [0.4, -0.2, 0.3], [-0.4, -0.2, 0.3], [-0.4, 0.3, -0.3], [-0.4, 0.3, 0.3]])))  # wedge
    for k in range(3):
        shapes.append(s.add_convex_polyhedron((rng.standard_normal((18 + 6 * k, 3)) * (0.22 + 0.04 * k)).astype(np.float32) + np.float32([0.1 * k, 0.0, 0.05])))       # clouds, off-centre
    rock = s.add_body(body_type=BODY_FIXED, translation=(0.0, 0.45, 0.0), rotation=(0.1, 0.3, 0.0, 0.9486833))
    s.add_collider(rock, shape=SHAPE_CONVEX, half_extents=(shapes[5], 0, 0))
    side = int(np.ceil(n ** (1.0 / 3.0)))
    k = 0
    for iy in range(side * 2):
        for ix in range(side):
            for iz in range(side):
                if k >= n:
                    break
                q = rng.normal(size=4).astype(np.float32)
                q /= np.linalg.norm(q)
                pos = (np.float32(1.4 * (ix - side / 2) + 0.1 * rng.random()), np.float32(2.0 + 1.5 * iy), np.float32(1.4 * (iz - side / 2) + 0.1 * rng.random()))
                b = s.add_body(translation=pos, rotation=tuple(q), linvel=tuple((rng.normal(size=3) * 1.0).astype(np.float32)), angvel=tuple((rng.normal(size=3) * 2.0).astype(np.float32)),
                               angular_damping=0.2 if k % 4 == 0 else 0.0)
                fr = np.float32(0.3 + 0.5 * rng.random())
                kind = k % 8
                if kind in (0, 2, 4, 6):
                    s.add_collider(b, shape=SHAPE_CONVEX, half_extents=(shapes[(k // 2) % len(shapes)], 0, 0), friction=fr, density=np.float32(0.8 + 1.5 * rng.random()),
                                   restitution=0.2 if k % 6 == 0 else 0.0)
                elif kind == 1:
                    s.add_collider(b, half_extents=tuple((0.2 + 0.3 * rng.random(size=3)).astype(np.float32)), friction=fr)
                elif kind == 3:
                    s.add_collider(b, shape=SHAPE_BALL, half_extents=(np.float32(0.25 + 0.2 * rng.random()), 0, 0), friction=fr)
                elif kind == 5:
                    s.add_collider(b, shape=SHAPE_CAPSULE, half_extents=(np.float32(0.3 + 0.2 * rng.random()), np.float32(0.15 + 0.15 * rng.random()), float(k % 3)), friction=fr)
                else:
                    s.add_collider(b, shape=SHAPE_CYLINDER if k % 16 == 7 else SHAPE_CONE, half_extents=(np.float32(0.25 + 0.2 * rng.random()), np.float32(0.2 + 0.2 * rng.random()), 0.0), friction=fr)
                k += 1
    b = s.add_body(translation=(2.5, 3.5, -2.5), angvel=(0.5, 1.0, 0.0))                                              # a dumbbell of two polyhedra
    s.add_collider(b, shape=SHAPE_CONVEX, half_extents=(shapes[1], 0, 0), translation=(-0.5, 0.0, 0.0), density=1.5)
    s.add_collider(b, shape=SHAPE_CONVEX, half_extents=(shapes[3], 0, 0), translation=(0.5, 0.1, 0.0), rotation=(0.0, 0.3826834, 0.0, 0.9238795))
    return s


def round_clutter(n: int = 30, seed: int = 6) -> Scene:
    """Seeded test scene (not a reference scene) for the round shapes (ColliderBuilder::round_cuboid / round_cylinder / round_cone /
    round_convex_hull, collider.rs:700-1090: parry RoundShape<S> = the inner shape dilated by a border radius): they tumble with plain
    cuboids, balls, capsules and cylinders on a round-cuboid slab and a half-space ramp inside four walls; a fixed round cone stands in
    the middle."""
    rng = np.random.default_rng(seed)
    s = Scene(name=f"round_clutter_{n}_{seed}", gravity=(0.0, -9.81, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, shape=SHAPE_ROUND_CUBOID, half_extents=(9.0, 0.4, 9.0), border_radius=0.1)
    s.add_collider(g, shape=SHAPE_HALFSPACE, half_extents=(-0.5, 0.8660254, 0.0), translation=(3.0, 0.5, 0.0))
    for sx, sz, hx, hz in ((4.5, 0.0, 0.25, 4.5), (-4.5, 0.0, 0.25, 4.5), (0.0, 4.5, 4.5, 0.25), (0.0, -4.5, 4.5, 0.25)):
        wb = s.add_body(body_type=BODY_FIXED, translation=(sx, 1.0, sz))
        s.add_collider(wb, half_extents=(hx, 1.0, hz))
    pid = s.add_convex_polyhedron((rng.standard_normal((20, 3)) * 0.25).astype(np.float32))
    pillar = s.add_body(body_type=BODY_FIXED, translation=(0.0, 0.65, 0.0))
    s.add_collider(pillar, shape=SHAPE_ROUND_CONE, half_extents=(0.5, 0.6, 0.0), border_radius=0.08)
    side = int(np.ceil(n ** (1.0 / 3.0)))
    k = 0
    for iy in range(side * 2):
        for ix in range(side):
            for iz in range(side):
                if k >= n:
                    break
                q = rng.normal(size=4).astype(np.float32)
                q /= np.linalg.norm(q)
                pos = (np.float32(1.4 * (ix - side / 2) + 0.1 * rng.random()), np.float32(2.0 + 1.5 * iy), np.float32(1.4 * (iz - side / 2) + 0.1 * rng.random()))
                b = s.add_body(translation=pos, rotation=tuple(q), linvel=tuple((rng.normal(size=3) * 1.0).astype(np.float32)), angvel=tuple((rng.normal(size=3) * 2.0).astype(np.float32)),
                               angular_damping=0.2 if k % 4 == 0 else 0.0)
                fr, br = np.float32(0.3 + 0.5 * rng.random()), np.float32(0.03 + 0.09 * rng.random())
                kind = k % 8
                if kind == 0:
                    s.add_collider(b, shape=SHAPE_ROUND_CUBOID, half_extents=tuple((0.15 + 0.25 * rng.random(size=3)).astype(np.float32)), border_radius=br, friction=fr)
                elif kind == 1:
                    s.add_collider(b, shape=SHAPE_ROUND_CYLINDER, half_extents=(np.float32(0.2 + 0.2 * rng.random()), np.float32(0.2 + 0.2 * rng.random()), 0.0), border_radius=br, friction=fr)
                elif kind == 2:
                    s.add_collider(b, shape=SHAPE_ROUND_CONE, half_extents=(np.float32(0.25 + 0.2 * rng.random()), np.float32(0.2 + 0.2 * rng.random()), 0.0), border_radius=br, friction=fr)
                elif kind == 3:
                    s.add_collider(b, shape=SHAPE_ROUND_CONVEX_POLYHEDRON, half_extents=(pid, 0, 0), border_radius=br, friction=fr, density=1.5)
                elif kind == 4:
                    s.add_collider(b, half_extents=tuple((0.2 + 0.3 * rng.random(size=3)).astype(np.float32)), friction=fr)
                elif kind == 5:
                    s.add_collider(b, shape=SHAPE_BALL, half_extents=(np.float32(0.25 + 0.2 * rng.random()), 0, 0), friction=fr)
                elif kind == 6:
                    s.add_collider(b, shape=SHAPE_CAPSULE, half_extents=(np.float32(0.3 + 0.2 * rng.random()), np.float32(0.15 + 0.15 * rng.random()), float(k % 3)), friction=fr)
                else:
                    s.add_collider(b, shape=SHAPE_CYLINDER, half_extents=(np.float32(0.25 + 0.2 * rng.random()), np.float32(0.2 + 0.2 * rng.random()), 0.0), friction=fr)
                k += 1
    return s


def shapes_rain(n: int = 8000, seed: int = 1, spacing: float = 1.6) -> Scene:
    """Measurement scene (not a reference scene) for the support-mapped shapes at scale: `n` bodies — cuboids, balls, capsules, cylinders,
    cones, convex polyhedra (eight registered clouds, shared) and the round variants in equal parts — in three layers over a large slab,
    random orientations and spins, falling into a loose carpet: thousands of GJK / EPA manifolds per step while it lands."""
    rng = np.random.default_rng(seed)
    s = Scene(name=f"shapes_rain_{n}_{seed}", gravity=(0.0, -9.81, 0.0))
    side = int(np.ceil(np.sqrt(n / 3.0)))
    half = 0.5 * spacing * side + 2.0
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.5, 0.0))
    s.add_collider(g, half_extents=(half, 0.5, half))
    polys = [s.add_convex_polyhedron((rng.standard_normal((12 + 4 * k, 3)) * 0.22).astype(np.float32)) for k in range(8)]
    k = 0
    for layer in range(3):
        for ix in range(side):
            for iz in range(side):
                if k >= n:
                    break
                q = rng.normal(size=4).astype(np.float32)
                q /= np.linalg.norm(q)
                pos = (np.float32(spacing * (ix - side / 2) + 0.2 * rng.random()), np.float32(1.0 + 1.3 * layer + 0.2 * rng.random()), np.float32(spacing * (iz - side / 2) + 0.2 * rng.random()))
                b = s.add_body(translation=pos, rotation=tuple(q), angvel=tuple((rng.normal(size=3) * 1.5).astype(np.float32)))
                kind, br = k % 10, np.float32(0.04 + 0.04 * rng.random())
                a, c = np.float32(0.2 + 0.15 * rng.random()), np.float32(0.2 + 0.15 * rng.random())
                if kind == 0:
                    s.add_collider(b, half_extents=(a, c, np.float32(0.2 + 0.15 * rng.random())))
                elif kind == 1:
                    s.add_collider(b, shape=SHAPE_BALL, half_extents=(a, 0, 0))
                elif kind == 2:
                    s.add_collider(b, shape=SHAPE_CAPSULE, half_extents=(a, np.float32(0.5 * c), float(k % 3)))
                elif kind == 3:
                    s.add_collider(b, shape=SHAPE_CYLINDER, half_extents=(a, c, 0.0))
                elif kind == 4:
                    s.add_collider(b, shape=SHAPE_CONE, half_extents=(a, c, 0.0))
                elif kind == 5:
                    s.add_collider(b, shape=SHAPE_CONVEX, half_extents=(polys[(k // 10) % 8], 0, 0))
                elif kind == 6:
                    s.add_collider(b, shape=SHAPE_ROUND_CUBOID, half_extents=(a, c, a), border_radius=br)
                elif kind == 7:
                    s.add_collider(b, shape=SHAPE_ROUND_CYLINDER, half_extents=(a, c, 0.0), border_radius=br)
                elif kind == 8:
                    s.add_collider(b, shape=SHAPE_ROUND_CONE, half_extents=(a, c, 0.0), border_radius=br)
                else:
                    s.add_collider(b, shape=SHAPE_ROUND_CONVEX_POLYHEDRON, half_extents=(polys[(k // 10) % 8], 0, 0), border_radius=br)
                k += 1
    return s


# ---- the reference's box3d ports that need convex hulls (examples3d/b3d_junkyard.rs, b3d_washer.rs; not BASELINE configs) ----
def _b3_rock(radius: float = 1.5) -> np.ndarray:
    """b3d_junkyard.rs:102-124 (box3d b3CreateRock): 10 points of a Fibonacci lattice on a sphere, the angle advanced by the rotation
    recurrence of the original, in f32"""
    f = np.float32
    n = 10
    phi = (f(1.0) + np.sqrt(f(5.0))) / f(2.0)
    theta = f(2.0) * f(np.pi) / phi
    ds, dc = np.sin(theta, dtype=np.float32), np.cos(theta, dtype=np.float32)
    c, s = f(1.0), f(0.0)
    pts = []
    for i in range(n):
        z = f(1.0) - (f(2.0) * f(i) + f(1.0)) / f(n)
        rxy = np.sqrt(f(1.0) - z * z)
        pts.append((f(radius) * rxy * c, f(radius) * rxy * s, f(radius) * z))
        c, s = dc * c - ds * s, ds * c + dc * s
    return np.array(pts, np.float32)


def _b3_cylinder(height: float, radius: float, y_offset: float, sides: int) -> np.ndarray:
    """b3d_junkyard.rs:84-100 (box3d b3CreateCylinder): 2 * sides points of a Y cylinder with its base at y_offset"""
    f = np.float32
    da, a, pts = f(2.0) * f(np.pi) / f(sides), f(0.0), []
    for _ in range(sides):
        sa, ca = np.sin(a, dtype=np.float32), np.cos(a, dtype=np.float32)
        pts.append((f(radius) * ca, f(y_offset), f(radius) * sa))
        pts.append((f(radius) * ca, f(y_offset) + f(height), f(radius) * sa))
        a = a + da
    return np.array(pts, np.float32)


def junkyard(layers: int = 24, nx: int = 21, nz: int = 21) -> Scene:
    """examples3d/b3d_junkyard.rs:10-81 (box3d `junkyard`): a walled arena (one fixed body, five cuboids), `layers` x 21 x 21 convex
    "rocks" that share ONE hull (SharedShape::convex_hull(&create_rock(1.5))), and an orbiting kinematic position-based pusher — a
    32-point cylinder hull — driven every step through junkyard_pusher_target.  Full size: 10,584 rocks."""
    s = Scene(name=f"b3d_junkyard_{layers}x{nx}x{nz}", gravity=(0.0, -10.0, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -1.0, 0.0))
    s.add_collider(g, half_extents=(120.0, 1.0, 120.0))
    for hx, hy, hz, off in ((1.0, 8.0, 50.0, (-50.0, 8.0, 0.0)), (1.0, 8.0, 50.0, (50.0, 8.0, 0.0)), (50.0, 8.0, 1.0, (0.0, 8.0, -50.0)), (50.0, 8.0, 1.0, (0.0, 8.0, 50.0))):
        s.add_collider(g, half_extents=(hx, hy, hz), translation=off)
    rock = s.add_convex_polyhedron(_b3_rock(1.5))
    height = np.float32(24.0)
    for y in range(layers):
        for x in range(nx):
            for z in range(nz):
                pos = (np.float32(-40.0) + np.float32(4.0) * np.float32(x), np.float32(4.0) * np.float32(y) + height + np.float32(1.0), np.float32(-40.0) + np.float32(4.0) * np.float32(z))
                b = s.add_body(translation=tuple(float(v) for v in pos))
                s.add_collider(b, shape=SHAPE_CONVEX, half_extents=(rock, 0, 0))
    pusher = s.add_body(body_type=BODY_KINEMATIC_POSITION, translation=(35.0, 0.0, 0.0))
    s.add_collider(pusher, shape=SHAPE_CONVEX, half_extents=(s.add_convex_polyhedron(_b3_cylinder(24.0, 4.0, 0.0, 16)), 0, 0))
    s.pusher = pusher
    return s


def junkyard_pusher_target(step: int) -> np.ndarray:
    """b3d_junkyard.rs:69-78: the pusher's next_kinematic_translation before step `step` (1-based): 35 m from the axis, -6 degrees per second"""
    f = np.float32
    degrees = f(0.0)
    for _ in range(step):
        degrees = degrees + f(-6.0) * (f(1.0) / f(60.0))
    rad = degrees * f(np.pi) / f(180.0)
    return np.array([f(35.0) * np.cos(rad, dtype=np.float32), 0.0, f(35.0) * np.sin(rad, dtype=np.float32), 0.0, 0.0, 0.0, 1.0], np.float32)


def _qrot_f32(q, v):
    """glam Quat * Vec3 in f32"""
    f = np.float32
    b = np.array(q[:3], np.float32); w = f(q[3]); v = np.array(v, np.float32)
    b2 = b @ b
    return (v * (w * w - b2) + b * ((v @ b) * f(2.0)) + np.cross(b, v).astype(np.float32) * (w * f(2.0))).astype(np.float32)


def washer(grid: int = 20) -> Scene:
    """examples3d/b3d_washer.rs:10-100 (box3d `washer`): a kinematic velocity-based ring — 36 outer segments + 4 paddles, each its own
    8-point convex hull, on ONE body that turns at 25 degrees per second — tumbling a grid^3 block of 0.4 m cubes of density 1000.
    Full size: 8,000 cubes."""
    f = np.float32
    s = Scene(name=f"b3d_washer_{grid}", gravity=(0.0, -10.0, 0.0))
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -1.0, 0.0))
    s.add_collider(g, half_extents=(60.0, 1.0, 60.0))
    ring = s.add_body(body_type=BODY_KINEMATIC_VELOCITY, translation=(0.0, 21.0, 0.0), angvel=(0.0, 0.0, float(f(np.pi) / f(180.0) * f(25.0))), linvel=(0.001, -0.002, 0.0))
    r0, r1, r2 = f(14.0), f(16.0), f(18.0)
    neg_d, pos_d = np.array([0, 0, -10], np.float32), np.array([0, 0, 10], np.float32)
    angle = f(np.pi) / f(18.0)

    def axis_z(a):
        return (0.0, 0.0, float(np.sin(a * f(0.5), dtype=np.float32)), float(np.cos(a * f(0.5), dtype=np.float32)))
    q, qo = axis_z(angle), axis_z(f(0.1) * angle)
    qo_inv = (-qo[0], -qo[1], -qo[2], qo[3])
    u1 = np.array([1, 0, 0], np.float32)
    for i in range(36):
        u2 = np.array([1, 0, 0], np.float32) if i == 35 else _qrot_f32(q, u1)
        a1, a2 = _qrot_f32(qo_inv, u1), _qrot_f32(qo, u2)
        pts = [neg_d + r1 * a1, neg_d + r2 * a1, neg_d + r1 * a2, neg_d + r2 * a2, pos_d + r1 * a1, pos_d + r2 * a1, pos_d + r1 * a2, pos_d + r2 * a2]
        s.add_collider(ring, shape=SHAPE_CONVEX, half_extents=(s.add_convex_polyhedron(np.array(pts, np.float32)), 0, 0))
        if i % 9 == 0:
            pts = [neg_d + r0 * u1, neg_d + r1 * u1, neg_d + r0 * u2, neg_d + r1 * u2, pos_d + r0 * u1, pos_d + r1 * u1, pos_d + r0 * u2, pos_d + r1 * u2]
            s.add_collider(ring, shape=SHAPE_CONVEX, half_extents=(s.add_convex_polyhedron(np.array(pts, np.float32)), 0, 0))
        u1 = u2
    a = f(0.2)
    x = f(-2.0) * a * f(grid)
    for _ in range(grid):
        y = f(-2.0) * a * f(grid) + f(21.0)
        for _ in range(grid):
            z = f(-2.0) * a * f(grid)
            for _ in range(grid):
                b = s.add_body(translation=(float(x), float(y), float(z)))
                s.add_collider(b, half_extents=(float(a), float(a), float(a)), density=1000.0)
                z = z + f(4.0) * a
            y = y + f(4.0) * a
        x = x + f(4.0) * a
    return s


def compound3(num: int = 8, numy: int = 15) -> Scene:
    """examples3d/compound3.rs:8-80: U-shaped bodies raining on a slab — the lower half of the layers as THREE colliders on one body,
    the upper half as ONE compound collider of the same three cuboids.  Full size: 8 x 15 x 8 = 960 bodies."""
    f = np.float32
    s = Scene(name=f"compound3_{num}x{numy}")
    g = s.add_body(body_type=BODY_FIXED, translation=(0.0, -0.1, 0.0))
    s.add_collider(g, half_extents=(50.0, 0.1, 50.0))
    rad = f(0.2)
    shift = rad * f(4.0) + rad
    cx, cy = shift * f(num // 2), shift / f(2.0)
    offset = -f(num) * (rad * f(2.0) + rad) * f(0.5)
    arm = float(rad * f(10.0))
    for j in range(numy):
        for i in range(num):
            for k in range(num):
                x = f(i) * shift * f(5.0) - cx + offset
                y = f(j) * (shift * f(5.0)) + cy + f(3.0)
                z = f(k) * shift * f(2.0) - cx + offset
                b = s.add_body(translation=(float(x), float(y), float(z)), can_sleep=1)
                if j < numy // 2:
                    s.add_collider(b, half_extents=(arm, float(rad), float(rad)))
                    s.add_collider(b, half_extents=(float(rad), arm, float(rad)), translation=(arm, arm, 0.0))
                    s.add_collider(b, half_extents=(float(rad), arm, float(rad)), translation=(-arm, arm, 0.0))
                else:
                    cid = s.add_compound([collider_desc(half_extents=(arm, float(rad), float(rad))),
                                          collider_desc(half_extents=(float(rad), arm, float(rad)), translation=(arm, arm, 0.0)),
                                          collider_desc(half_extents=(float(rad), arm, float(rad)), translation=(-arm, arm, 0.0))])
                    s.add_collider(b, shape=SHAPE_COMPOUND, half_extents=(cid, 0, 0))
        offset -= f(0.05) * rad * (f(num) - f(1.0))
    return s


def heightfield3(num: int = 8, layers: int = 20, nsubdivs: int = 20, mesh: bool = False) -> Scene:
    """examples3d/heightfield3.rs:8-100 (mesh=False) and trimesh3.rs:8-100 (mesh=True: the same terrain through
    HeightField::to_trimesh, here the height field's own two triangles per cell handed to SharedShape::trimesh): cuboids, balls,
    round cylinders, cones, capsules and three-cuboid compounds, layer by layer, on a rolling 100 x 100 terrain with raised borders.
    Full size: 8 x 20 x 8 = 1,280 bodies."""
    f = np.float32
    s = Scene(name=f"{'trimesh3' if mesh else 'heightfield3'}_{num}x{layers}")
    i, j = np.meshgrid(np.arange(nsubdivs + 1), np.arange(nsubdivs + 1), indexing="ij")
    x = i.astype(np.float32) * f(100.0) / f(nsubdivs)
    z = j.astype(np.float32) * f(100.0) / f(nsubdivs)
    h = (np.sin(x, dtype=np.float32) + np.cos(z, dtype=np.float32)).astype(np.float32)
    h[0, :] = h[nsubdivs, :] = 10.0
    h[:, 0] = h[:, nsubdivs] = 10.0
    g = s.add_body(body_type=BODY_FIXED)
    if mesh:
        # HeightField::to_trimesh: vertex (r, c) at ((c / n - 0.5) sx, h[r, c] sy, (r / n - 0.5) sz), two triangles per cell
        n = nsubdivs
        verts = np.zeros(((n + 1) * (n + 1), 3), np.float32)
        for r in range(n + 1):
            for c in range(n + 1):
                verts[r * (n + 1) + c] = ((f(c) / f(n) - f(0.5)) * f(100.0), h[r, c], (f(r) / f(n) - f(0.5)) * f(100.0))
        tris = []
        for r in range(n):
            for c in range(n):
                p00, p10, p01, p11 = r * (n + 1) + c, (r + 1) * (n + 1) + c, r * (n + 1) + c + 1, (r + 1) * (n + 1) + c + 1
                tris += [(p00, p10, p11), (p00, p11, p01)]
        s.add_collider(g, shape=SHAPE_TRIMESH, half_extents=(s.add_trimesh(verts, np.array(tris, np.uint32)), 0, 0))
    else:
        s.add_collider(g, shape=SHAPE_TRIMESH, half_extents=(s.add_heightfield(h, (100.0, 1.0, 100.0)), 0, 0))
    rad = f(1.0)
    shift = rad * f(2.0) + rad
    cx, cy = shift * f(num // 2), shift / f(2.0)
    r = float(rad)
    for jj in range(layers):
        for ii in range(num):
            for kk in range(num):
                b = s.add_body(translation=(float(f(ii) * shift - cx), float(f(jj) * shift + cy + f(3.0)), float(f(kk) * shift - cx)), can_sleep=1)
                kind = jj % 6
                if kind == 0:
                    s.add_collider(b, half_extents=(r, r, r))
                elif kind == 1:
                    s.add_collider(b, shape=SHAPE_BALL, half_extents=(r, 0.0, 0.0))
                elif kind == 2:
                    s.add_collider(b, shape=SHAPE_ROUND_CYLINDER, half_extents=(r, r, 0.0), border_radius=r / 10.0)
                elif kind == 3:
                    s.add_collider(b, shape=SHAPE_CONE, half_extents=(r, r, 0.0))
                elif kind == 4:
                    s.add_collider(b, shape=SHAPE_CAPSULE, half_extents=(r, r, 1.0))
                else:
                    cid = s.add_compound([collider_desc(half_extents=(r, r / 2.0, r / 2.0)),
                                          collider_desc(half_extents=(r / 2.0, r, r / 2.0), translation=(r, 0.0, 0.0)),
                                          collider_desc(half_extents=(r / 2.0, r, r / 2.0), translation=(-r, 0.0, 0.0))])
                    s.add_collider(b, shape=SHAPE_COMPOUND, half_extents=(cid, 0, 0))
    return s
