/* ro_ccd.h — the continuous-collision pass of the oracle (test infrastructure only, like the rest of oracle/).
 *
 * DRIVER: restated from /root/reference/src/dynamics/ccd/ccd_solver.rs:158-340 (solve_continuous, apply_clamps) and
 * sweeps.rs:22-41 (tiers), :268-351 (cast_collider_pair: solid pairs stop only at 0 < fraction < max_fraction), :470-640
 * (sweep_fast_body), with the activation criterion of rigid_body_components.rs:1131-1157.
 *
 * TIME OF IMPACT: the reference calls parry3d's query::sweep_toi::{Sweep, ToiProxy, sweep_time_of_impact} (parry3d 0.30.2, not
 * under /root/reference, no lockfile): a Box2D-style root finder over point-cloud proxies.  Its source cannot be read here, so
 * the query is NOT restated: it is replaced by conservative advancement over a LOWER BOUND of the distance —
 *   - Sweep: the centre of mass moves on a straight line, the rotation is the normalised linear blend of the end rotations
 *     (parry: same centre-of-mass pivot; its rotation interpolation is not visible from the tree);
 *   - separation(t): the largest separation over the SAT axes the manifold generators already use (face normals both ways and
 *     edge x edge for cuboids, cuboid axes against a segment for capsules), exact point projections for balls, segment / segment
 *     closest points for capsules, the support point for half-spaces: never above the true distance, equal to it for face, edge-edge
 *     and point features;
 *   - advance: t += (separation - target) / (approach speed along the separating direction + rotation bound), where the rotation
 *     bound is 4 tan(angle / 4) per unit fraction (the fastest instant of the blend) times the largest distance from the centre of
 *     mass of a point whose motion can change the separation (a ball: its centre; a capsule: its segment's ends; a cuboid: its
 *     vertices), until separation < target + tolerance (hit) or t >= max_fraction (miss).  The iteration cap ends in a MISS: it is
 *     only reached by a body that hovers within a few slops of a surface while tumbling or skimming along it, well inside the
 *     prediction distance, where the speculative contacts of the narrow phase hold it (a stop there would pin a sliding body at
 *     every step: the stutter of the reference's issue 932);
 *   - Box2D's constants: with slop = IntegrationParameters::allowed_linear_error, an impact is the moment the CORE shapes (a ball's
 *     centre, a capsule's segment, a polytope itself) come within max(slop, total radius - slop) of each other, tolerance slop / 4 —
 *     round shapes must overlap by a slop before they count, which is what lets a ball roll over the seams of a tiled floor.
 * Outcome-level pins: the reference's own CCD tests (ccd_default_vs_fixed.rs, issue_217, issue_932) restated in
 * tests/test_ccd_oracle.py.  Fraction-level parity with parry is unpinned and stated so in DESIGN.md.
 *
 * Every function here has a line-for-line twin in rapier_amd/csrc/rp_ccd.h (same operations in the same order: the device result
 * is compared bit for bit). */
#ifndef RO_CCD_H
#define RO_CCD_H

#define RO_CCD_MAX_ITERS 48

/* parry Sweep::from_poses(start, end, local_com) / transform_at(fraction) — see the header note */
typedef struct { v3 c0, c1; quat q0, q1; v3 local_com; } CcdSweep;
static inline CcdSweep ccd_sweep_from_poses(pose start, pose end, v3 local_com) {
    CcdSweep s;
    s.c0 = pose_tp(start, local_com); s.c1 = pose_tp(end, local_com);
    s.q0 = start.r; s.q1 = end.r;
    if (qdot(s.q0, s.q1) < 0.0f) s.q1 = Q(-s.q1.x, -s.q1.y, -s.q1.z, -s.q1.w);
    s.local_com = local_com;
    return s;
}
static inline pose ccd_sweep_transform_at(const CcdSweep *s, float t) {
    v3 c = vadd(s->c0, vmul(vsub(s->c1, s->c0), t));
    quat q = qnormalize(Q(s->q0.x + (s->q1.x - s->q0.x) * t, s->q0.y + (s->q1.y - s->q0.y) * t, s->q0.z + (s->q1.z - s->q0.z) * t,
                          s->q0.w + (s->q1.w - s->q0.w) * t));
    pose p; p.r = q; p.t = vsub(c, qrot(q, s->local_com));
    return p;
}

/* a collider's shape as the CCD query sees it: kind, (cuboid half extents | capsule half height, radius, axis | ball radius in x |
 * half-space normal) */
typedef SmShape CcdShape; /* ro_convex.h: the same record serves the support-mapped queries (cylinders, cones) */

static inline v3 ccd_clamp_box(v3 p, v3 he) { return V3(ro_clampf(p.x, -he.x, he.x), ro_clampf(p.y, -he.y, he.y), ro_clampf(p.z, -he.z, he.z)); }
/* separation of a point from a solid box (negative constant when inside) and the unit direction from the box to the point */
static inline float ccd_point_box(v3 p, v3 he, v3 *dir) {
    v3 dv = vsub(p, ccd_clamp_box(p, he));
    float dist = vlen(dv);
    if (!(dist > 0.0f)) { *dir = V3(0, 1, 0); return -1.0f; }
    *dir = vmul(dv, 1.0f / dist);
    return dist;
}
static inline float ccd_point_dir(v3 dv, v3 *dir) { /* |dv| and its direction; a zero vector counts as overlap */
    float dist = vlen(dv);
    if (!(dist > 0.0f)) { *dir = V3(0, 1, 0); return -1.0f; }
    *dir = vmul(dv, 1.0f / dist);
    return dist;
}

/* Lower bound of the distance between target shape 1 and fast shape 2 (pos12 = pose of 2 in the frame of 1) and the unit direction
 * n1, in the frame of 1, from 1 towards 2, along which it was measured.  A value <= 0 means "touching or overlapping". */
static inline float ccd_separation(const CcdShape *s1, const CcdShape *s2, pose pos12, v3 *n1) {
    const pose pos21 = pose_inv(pos12);
    if (s1->shape >= RO_SHAPE_CYLINDER || s2->shape >= RO_SHAPE_CYLINDER || s1->border > 0.0f || s2->border > 0.0f) { /* cylinders, cones, polyhedra, round shapes: the exact distance of the cores by GJK (ro_convex.h) */
        if (s1->shape == RO_SHAPE_HALFSPACE) {
            *n1 = s1->he;
            return vdot(s1->he, pose_tp(pos12, sm_support(s2, qrot_inv(pos12.r, vneg(s1->he))))) - s2->border;
        }
        float d = sm_distance(s1, s2, pos12, n1);
        return d < 0.0f ? d : d - sm_border_radius(s1) - sm_border_radius(s2);
    }
    if (s1->shape == RO_SHAPE_HALFSPACE) {
        const v3 n = s1->he;
        *n1 = n;
        if (s2->shape == RO_SHAPE_BALL) return vdot(n, pos12.t) - s2->radius;
        if (s2->shape == RO_SHAPE_CUBOID) {
            v3 n2 = qrot_inv(pos12.r, n);
            float ext = (fabsf(n2.x) * s2->he.x + fabsf(n2.y) * s2->he.y) + fabsf(n2.z) * s2->he.z;
            return vdot(n, pos12.t) - ext;
        }
        v3 e = vmul(capsule_axis_dir(s2->axis), s2->he.x);
        float da = vdot(n, pose_tp(pos12, vneg(e))), db = vdot(n, pose_tp(pos12, e));
        return ro_minf(da, db) - s2->radius;
    }
    if (s1->shape == RO_SHAPE_BALL) {
        if (s2->shape == RO_SHAPE_BALL) { float d = ccd_point_dir(pos12.t, n1); return d < 0.0f ? d : d - s1->radius - s2->radius; }
        if (s2->shape == RO_SHAPE_CUBOID) { /* the ball's centre against the box, in the box's frame */
            v3 dir2; float d = ccd_point_box(pos21.t, s2->he, &dir2);
            *n1 = qrot(pos12.r, vneg(dir2));
            return d < 0.0f ? d : d - s1->radius;
        }
        v3 e = vmul(capsule_axis_dir(s2->axis), s2->he.x);
        v3 p = segment_project_point(pose_tp(pos12, vneg(e)), pose_tp(pos12, e), V3(0, 0, 0));
        float d = ccd_point_dir(p, n1);
        return d < 0.0f ? d : d - s1->radius - s2->radius;
    }
    if (s1->shape == RO_SHAPE_CUBOID) {
        if (s2->shape == RO_SHAPE_BALL) { float d = ccd_point_box(pos12.t, s1->he, n1); return d < 0.0f ? d : d - s2->radius; }
        if (s2->shape == RO_SHAPE_CUBOID) {
            v3 d1, d2, d3;
            float sa = sat_normal_oneway(s1->he, s2->he, pos12, &d1);
            float sb = sat_normal_oneway(s2->he, s1->he, pos21, &d2);
            float sc = sat_edge_twoway(s1->he, s2->he, pos12, &d3);
            float sep = sa; *n1 = d1;
            if (sb > sep) { sep = sb; *n1 = qrot(pos12.r, vneg(d2)); }
            if (sc > sep) { sep = sc; *n1 = d3; }
            return sep;
        }
        v3 e = vmul(capsule_axis_dir(s2->axis), s2->he.x);
        v3 a2 = pose_tp(pos12, vneg(e)), b2 = pose_tp(pos12, e), d1, d3;
        float sa = sat_cuboid_segment_normal_oneway(s1->he, a2, b2, &d1);
        float sc = sat_cuboid_segment_edge_twoway(s1->he, a2, b2, &d3);
        float sep = sa; *n1 = d1;
        if (sc > sep) { sep = sc; *n1 = d3; }
        return sep - s2->radius;
    }
    /* target capsule */
    {
        v3 e1 = vmul(capsule_axis_dir(s1->axis), s1->he.x), a1 = vneg(e1), b1 = e1;
        if (s2->shape == RO_SHAPE_BALL) {
            v3 p = segment_project_point(a1, b1, pos12.t);
            float d = ccd_point_dir(vsub(pos12.t, p), n1);
            return d < 0.0f ? d : d - s1->radius - s2->radius;
        }
        if (s2->shape == RO_SHAPE_CUBOID) { /* the capsule's segment against the box, in the box's frame */
            v3 a = pose_tp(pos21, a1), b = pose_tp(pos21, b1), d1, d3;
            float sa = sat_cuboid_segment_normal_oneway(s2->he, a, b, &d1);
            float sc = sat_cuboid_segment_edge_twoway(s2->he, a, b, &d3);
            float sep = sa; v3 dir2 = d1;
            if (sc > sep) { sep = sc; dir2 = d3; }
            *n1 = qrot(pos12.r, vneg(dir2));
            return sep - s1->radius;
        }
        v3 e2 = vmul(capsule_axis_dir(s2->axis), s2->he.x);
        v3 a2 = pose_tp(pos12, vneg(e2)), b2 = pose_tp(pos12, e2);
        float s, t;
        closest_points_segment_segment(a1, b1, a2, b2, &s, &t);
        v3 p1 = vadd(a1, vmul(vsub(b1, a1), s)), p2 = vadd(a2, vmul(vsub(b2, a2), t));
        float d = ccd_point_dir(vsub(p2, p1), n1);
        return d < 0.0f ? d : d - s1->radius - s2->radius;
    }
}

/* One pair: the fraction in (0, max_fraction) at which the fast collider (shape s2 at sweep(t) * pos_wrt_parent) comes within
 * target + tolerance of the stationary target (shape s1 at target_pose), -1 for a miss, -2 for an initial overlap / touch (fraction 0),
 * which is "no impact" for a solid pair in 3D (sweeps.rs:268-275, :409-411) — except that a sub-shape of a composite target is then
 * tried again with the fast piece's core ball (ccd_core_of below). */
static inline float ccd_rot_radius(const CcdShape *s2, pose pos_wrt_parent, v3 local_com) { /* see the header: what rotation about the centre of mass can move */
    v3 c = vsub(pos_wrt_parent.t, local_com);
    if (s2->shape == RO_SHAPE_BALL) return vlen(c);
    if (s2->shape == RO_SHAPE_CAPSULE) {
        v3 e = qrot(pos_wrt_parent.r, vmul(capsule_axis_dir(s2->axis), s2->he.x));
        return ro_maxf(vlen(vsub(c, e)), vlen(vadd(c, e)));
    }
    return (vlen(c) + vlen(s2->he)) + s2->border;
}
static inline float ccd_cast_pair(const CcdShape *s1, pose target_pose, const CcdShape *s2, pose pos_wrt_parent, const CcdSweep *sw,
                                  float rot_radius, float max_fraction, float slop) {
    /* the impact distance of the surfaces: max(slop, total_radius - slop) between the cores, minus the radii */
    const float total_radius = sm_border_radius(s1) + sm_border_radius(s2); /* balls, capsules, round shapes */
    const float target = ro_maxf(slop, total_radius - slop) - total_radius, tol = 0.25f * slop;
    const v3 D = vsub(sw->c1, sw->c0);
    const quat dq = qmul(sw->q1, qconj(sw->q0));
    const float sv = sqrtf((dq.x * dq.x + dq.y * dq.y) + dq.z * dq.z);
    const float rot_bound = (4.0f * sv / (1.0f + fabsf(dq.w))) * rot_radius;
    float t = 0.0f;
    for (int iter = 0; iter < RO_CCD_MAX_ITERS; ++iter) {
        pose cp = pose_mul(ccd_sweep_transform_at(sw, t), pos_wrt_parent);
        pose pos12 = pose_inv_mul(target_pose, cp);
        v3 n1;
        float sep = ccd_separation(s1, s2, pos12, &n1);
        if (sep < target + tol) return iter == 0 ? -2.0f : t; /* -2: touching or overlapping at the start */
        v3 nw = qrot(target_pose.r, n1);
        float approach = -vdot(D, nw);
        if (approach < 0.0f) approach = 0.0f;
        float bound = approach + rot_bound;
        if (!(bound > 0.0f)) return -1.0f;
        t = t + (sep - target) / bound;
        if (!(t < max_fraction)) return -1.0f;
    }
    return -1.0f;
}

/* The core of a fast piece (sweeps.rs:166-173 FastSubShape::local_centroid / min_extent, handed to the composite sweep at
 * :386-391): a ball of CORE_FRACTION x the piece's smallest extent about its centre.  A piece that starts a step touching or
 * overlapping one triangle / cell / part of a composite target (a thin slab lying across a mesh that has no inside) is swept
 * again as this ball, so that its centre can never cross the sheet within a step (sweeps.rs:421-440 shows the retry for the 2D
 * proxies).  CORE_FRACTION is a parry constant (sweep_toi.rs, not under /root/reference): 0.25, the value of the Box2D v3
 * continuous pass this design follows — unpinned like the rest of the time-of-impact query.  The smallest extent is the shape's
 * ccd_thickness; the centre is the origin of the shape's own frame. */
#define RO_CCD_CORE_FRACTION 0.25f
static inline CcdShape ccd_core_of(const CcdShape *s2) {
    float th = s2->shape == RO_SHAPE_BALL ? s2->radius : s2->shape == RO_SHAPE_CAPSULE ? s2->radius : ro_minf(s2->he.x, ro_minf(s2->he.y, s2->he.z));
    if (s2->border > 0.0f) th = th + s2->border;
    CcdShape c = *s2;
    c.shape = RO_SHAPE_BALL; c.radius = RO_CCD_CORE_FRACTION * th; c.border = 0.0f; c.poly = NULL; c.he = V3(c.radius, 0.0f, 0.0f);
    return c;
}

/* conservative pre-filter: the whole swept volume of the fast BODY lies within max_extent of its centre-of-mass segment */
static inline int ccd_may_reach(v3 c0, v3 c1, float max_extent, v3 target_centre, float target_radius, float margin) {
    v3 p = segment_project_point(c0, c1, target_centre);
    float reach = (max_extent + target_radius) + margin;
    return vlen2(vsub(target_centre, p)) <= reach * reach;
}
#endif
