/* ro_convex.h — support-mapped shapes of the oracle: cylinders and cones (test infrastructure only, like the rest of oracle/).
 *
 * WHAT IT STANDS FOR.  rapier hands every pair with a Cylinder or a Cone (ColliderBuilder::cylinder / cone,
 * /root/reference/src/geometry/collider.rs:770, :789) to parry3d's DefaultQueryDispatcher::contact_manifolds
 * (call site /root/reference/src/geometry/narrow_phase/pair_update.rs:323-330), which routes them to
 *   contact_manifold_pfm_pfm        (two PolygonalFeatureMaps: cuboid, cylinder, cone, a capsule's segment + border radius):
 *                                   try_update_contacts -> GJK closest points (EPA when they overlap) -> local_support_feature of
 *                                   both shapes along the separating direction -> PolygonalFeature::contacts (2-D clipping) -> the
 *                                   GJK point pair when the clipping found nothing -> border radii -> match_contacts;
 *   contact_manifold_convex_ball    (a ball against the shape's point projection, non-solid);
 *   contact_manifold_halfspace_pfm  (the shape's support feature toward the plane).
 * parry3d 0.30.2 is NOT under /root/reference (Cargo dependency, no vendored source, no lockfile), so none of this can be
 * restated line by line: PARITY UNPINNED at the manifold level.  What follows is the crate's published algorithm as far as it is
 * known — the shapes' support functions, point projections, mass properties and feature approximations (a cylinder's cap and a
 * cone's base are SQUARES inscribed in the circle, the curved part is ONE segment), the GJK loop with its Voronoi simplex and
 * termination rules — with an expanding-polytope pass of our own (closest face by linear scan, horizon by edge cancellation,
 * fixed capacities) where parry has its heap-based EPA.  One known behaviour of 0.30.2 is restated from the reference's own
 * regression test (crates/rapier3d/tests/issue_810_cubes_thin_cylinder_tunnel.rs:1-8): the cap's square is ORIENTED TOWARD THE
 * CONTACT POINT, so that a small box landing anywhere on a wide disc gets a multi-point manifold.  Outcome-level pins: that test,
 * analytic rest heights / distances / mass properties (tests/test_convex_oracle.py).
 *
 * Every function here has a twin in rapier_amd/csrc/rp_convex.h performing the same operations in the same order: the device
 * result is compared bit for bit. */
#ifndef RO_CONVEX_H
#define RO_CONVEX_H
#include "ro_shapes.h"
#include "ro_polyhedron.h"

/* he: cuboid half extents | capsule: he.x = half height, radius, axis | ball: radius | cylinder / cone (axis Y): he = (radius,
 * half_height, radius) — the half extents of the local AABB — and radius */
typedef struct { int shape; v3 he; float radius; int axis; const RoPolyhedron *poly; /* RO_SHAPE_CONVEX_POLYHEDRON: he = the local AABB's half extents */
                 float border; /* a round shape (parry RoundShape<S>): `shape` is its inner shape, this its border radius */
                 v3 tri[3];    /* RO_SHAPE_TRIANGLE (a triangle of a TriMesh / HeightField, ro_composite): its vertices */ } SmShape;

#define RO_GJK_EPS_TOL 1.1920929e-6f          /* gjk::eps_tol() = 10 * f32::EPSILON */
#define RO_EPA_EPS_TOL 1.1920929e-5f          /* 100 * f32::EPSILON */
#define RO_GJK_REL_TOL 1.0e-5f              /* add_point: sine of the smallest angle a new vertex must add */
#define RO_GJK_MAX_ITERS 100
#define RO_EPA_MAXV 40
#define RO_EPA_MAXF 80
#define RO_EPA_MAXE 48

/* the part of a round shape that goes through GJK is its core (a ball's centre, a capsule's segment) */
static inline float sm_border_radius(const SmShape *s) { return (s->shape == RO_SHAPE_BALL || s->shape == RO_SHAPE_CAPSULE) ? s->radius : s->border; }

/* SupportMap::local_support_point of the core shape */
static inline v3 sm_support(const SmShape *s, v3 d) {
    if (s->shape == RO_SHAPE_CUBOID) return cuboid_support_point(s->he, d);
    if (s->shape == RO_SHAPE_CAPSULE) { /* Segment: a unless b is strictly further along d... (a . d > b . d ? a : b) */
        v3 e = capsule_axis_dir(s->axis);
        float c = vget(d, s->axis) * s->he.x;
        return (-c > c) ? vmul(e, -s->he.x) : vmul(e, s->he.x);
    }
    if (s->shape == RO_SHAPE_CYLINDER) {
        float n = sqrtf(d.x * d.x + d.z * d.z);
        v3 r = V3(0, 0, 0);
        if (n != 0.0f) r = V3(d.x / n * s->radius, 0.0f, d.z / n * s->radius);
        r.y = copysignf(s->he.y, d.y);
        return r;
    }
    if (s->shape == RO_SHAPE_CONE) {
        float n = sqrtf(d.x * d.x + d.z * d.z);
        if (n == 0.0f) return V3(0.0f, copysignf(s->he.y, d.y), 0.0f);
        v3 r = V3(d.x / n * s->radius, -s->he.y, d.z / n * s->radius);
        if (vdot(d, r) < d.y * s->he.y) r = V3(0.0f, s->he.y, 0.0f);
        return r;
    }
    if (s->shape == RO_SHAPE_TRIANGLE) { /* Triangle::local_support_point: the first vertex with the largest dot product */
        float da = vdot(s->tri[0], d), db = vdot(s->tri[1], d), dc = vdot(s->tri[2], d);
        if (da > db) return da > dc ? s->tri[0] : s->tri[2];
        return db > dc ? s->tri[1] : s->tri[2];
    }
    if (s->shape == RO_SHAPE_CONVEX_POLYHEDRON) { /* utils::point_cloud_support_point: the first vertex with the largest dot product */
        const RoPolyhedron *P = s->poly;
        int best = 0; float bd = vdot(P->pts[0], d);
        for (int i = 1; i < P->nv; ++i) { float x = vdot(P->pts[i], d); if (x > bd) { bd = x; best = i; } }
        return P->pts[best];
    }
    return V3(0, 0, 0); /* ball: its centre */
}

/* a point of the configuration-space obstacle shape1 - shape2 with its two origins (CSOPoint::from_shapes), all in frame 1 */
typedef struct { v3 p, o1, o2; } CsoPt;
static inline CsoPt cso_support(const SmShape *s1, const SmShape *s2, pose pos12, v3 dir) {
    CsoPt r;
    r.o1 = sm_support(s1, dir);
    r.o2 = pose_tp(pos12, sm_support(s2, qrot_inv(pos12.r, vneg(dir))));
    r.p = vsub(r.o1, r.o2);
    return r;
}

/* ---- Voronoi simplex: the origin projected on a segment / triangle / tetrahedron (Ericson, Real-Time Collision Detection 5.1) ---- */
/* each returns the mask of the vertices that carry the projection and their barycentric coordinates */
static inline int sx_proj_seg(v3 a, v3 b, float bc[4]) {
    v3 ab = vsub(b, a);
    float t = -vdot(a, ab);
    if (t <= 0.0f) { bc[0] = 1.0f; bc[1] = 0.0f; return 1; }
    float denom = vdot(ab, ab);
    if (t >= denom) { bc[0] = 0.0f; bc[1] = 1.0f; return 2; }
    t = t / denom;
    bc[0] = 1.0f - t; bc[1] = t;
    return 3;
}
static inline int sx_proj_tri(v3 a, v3 b, v3 c, float bc[4]) {
    v3 ab = vsub(b, a), ac = vsub(c, a);
    float d1 = -vdot(ab, a), d2 = -vdot(ac, a);
    bc[0] = bc[1] = bc[2] = 0.0f;
    if (d1 <= 0.0f && d2 <= 0.0f) { bc[0] = 1.0f; return 1; }
    float d3 = -vdot(ab, b), d4 = -vdot(ac, b);
    if (d3 >= 0.0f && d4 <= d3) { bc[1] = 1.0f; return 2; }
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) { float v = d1 / (d1 - d3); bc[0] = 1.0f - v; bc[1] = v; return 3; }
    float d5 = -vdot(ab, c), d6 = -vdot(ac, c);
    if (d6 >= 0.0f && d5 <= d6) { bc[2] = 1.0f; return 4; }
    float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) { float w = d2 / (d2 - d6); bc[0] = 1.0f - w; bc[2] = w; return 5; }
    float va = d3 * d6 - d5 * d4;
    if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) { float w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); bc[1] = 1.0f - w; bc[2] = w; return 6; }
    float denom = 1.0f / ((va + vb) + vc);
    float v = vb * denom, w = vc * denom;
    bc[0] = (1.0f - v) - w; bc[1] = v; bc[2] = w;
    return 7;
}

typedef struct { CsoPt v[4]; float bc[4]; int n; CsoPt pv[4]; float pbc[4]; int pn; } GjkSimplex;

static inline void sx_reset(GjkSimplex *s, CsoPt p) { s->n = 1; s->v[0] = p; s->bc[0] = 1.0f; s->pn = 1; s->pv[0] = p; s->pbc[0] = 1.0f; }
/* keep the vertices of `mask` (in order), with their barycentric coordinates */
static inline void sx_keep(GjkSimplex *s, int mask, const float bc[4]) {
    int k = 0;
    for (int i = 0; i < s->n; ++i)
        if (mask & (1 << i)) { s->v[k] = s->v[i]; s->bc[k] = bc[i]; ++k; }
    s->n = k;
}
/* VoronoiSimplex::project_origin_and_reduce; *inside = the origin lies inside a tetrahedron (dimension stays 3) */
static inline v3 sx_project_and_reduce(GjkSimplex *s, int *inside) {
    float bc[4] = {0, 0, 0, 0};
    *inside = 0;
    if (s->n == 1) { s->bc[0] = 1.0f; return s->v[0].p; }
    if (s->n == 2) {
        int mask = sx_proj_seg(s->v[0].p, s->v[1].p, bc);
        sx_keep(s, mask, bc);
    } else if (s->n == 3) {
        int mask = sx_proj_tri(s->v[0].p, s->v[1].p, s->v[2].p, bc);
        sx_keep(s, mask, bc);
    } else {
        /* the faces that see the origin on their outer side; the closest of their projections wins (ties: the first) */
        static const int F[4][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 3, 1}, {1, 2, 3, 0}};
        float best = FLT_MAX; int best_f = -1, best_mask = 0; float best_bc[4] = {0, 0, 0, 0};
        for (int f = 0; f < 4; ++f) {
            v3 a = s->v[F[f][0]].p, b = s->v[F[f][1]].p, c = s->v[F[f][2]].p, d = s->v[F[f][3]].p;
            v3 nrm = vcross(vsub(b, a), vsub(c, a));
            float sd = vdot(nrm, vsub(d, a)), so = -vdot(nrm, a);
            if (sd * so > 0.0f) continue; /* the origin is on the inner side of this face */
            float fb[4];
            int m = sx_proj_tri(a, b, c, fb);
            v3 q = vadd(vadd(vmul(a, fb[0]), vmul(b, fb[1])), vmul(c, fb[2]));
            float d2 = vlen2(q);
            if (d2 < best) {
                best = d2; best_f = f; best_mask = 0;
                for (int k = 0; k < 4; ++k) best_bc[k] = 0.0f;
                for (int k = 0; k < 3; ++k) if (m & (1 << k)) { best_mask |= 1 << F[f][k]; best_bc[F[f][k]] = fb[k]; }
            }
        }
        if (best_f < 0) { *inside = 1; return V3(0, 0, 0); }
        sx_keep(s, best_mask, best_bc);
    }
    if (s->n == 3) { /* the projection falls inside a triangle: along the triangle's normal, exactly — the barycentric sum of a large
                        triangle that passes the origin at 1e-3 has lost the direction by then (the witness points still use bc) */
        v3 nrm = vcross(vsub(s->v[1].p, s->v[0].p), vsub(s->v[2].p, s->v[0].p));
        float l2 = vlen2(nrm);
        if (l2 > 0.0f) return vmul(nrm, vdot(nrm, s->v[0].p) / l2);
    }
    v3 q = V3(0, 0, 0);
    for (int i = 0; i < s->n; ++i) q = vadd(q, vmul(s->v[i].p, s->bc[i]));
    return q;
}
/* VoronoiSimplex::add_point: refuses a point that does not raise the dimension */
static inline int sx_add_point(GjkSimplex *s, CsoPt pt) {
    s->pn = s->n;
    for (int i = 0; i < s->n; ++i) { s->pv[i] = s->v[i]; s->pbc[i] = s->bc[i]; }
    /* parry tests absolute sizes against eps_tol here (|v0 - pt|^2, |ab x ac|^2, the distance from the plane), which refuses good
     * points once the simplex has shrunk around a nearly touching configuration (distances of 1e-3: the direction GJK then stops
     * with is off by tens of degrees); the tests are relative to the simplex's own size instead */
    for (int i = 0; i < s->n; ++i) {
        v3 d = vsub(s->v[i].p, pt.p);
        if (d.x == 0.0f && d.y == 0.0f && d.z == 0.0f) return 0;
    }
    if (s->n == 2) {
        v3 ab = vsub(s->v[1].p, s->v[0].p), ac = vsub(pt.p, s->v[0].p);
        if (!(vlen2(vcross(ab, ac)) > RO_GJK_REL_TOL * RO_GJK_REL_TOL * (vlen2(ab) * vlen2(ac)))) return 0;
    } else if (s->n == 3) {
        v3 ab = vsub(s->v[1].p, s->v[0].p), ac = vsub(s->v[2].p, s->v[0].p), ap = vsub(pt.p, s->v[0].p);
        v3 nrm = vcross(ab, ac);
        float h = vdot(nrm, ap);
        if (!(h * h > RO_GJK_REL_TOL * RO_GJK_REL_TOL * (vlen2(nrm) * vlen2(ap)))) return 0;
    } else if (s->n != 1) return 0;
    s->v[s->n++] = pt;
    return 1;
}
/* gjk::result: the two closest points from the barycentric coordinates (of the previous simplex when `prev`) */
static inline void sx_result(const GjkSimplex *s, int prev, v3 *p1, v3 *p2) {
    const CsoPt *v = prev ? s->pv : s->v; const float *bc = prev ? s->pbc : s->bc; int n = prev ? s->pn : s->n;
    v3 a = V3(0, 0, 0), b = V3(0, 0, 0);
    for (int i = 0; i < n; ++i) { a = vadd(a, vmul(v[i].o1, bc[i])); b = vadd(b, vmul(v[i].o2, bc[i])); }
    *p1 = a; *p2 = b;
}

enum { RO_GJK_INTERSECTION = 0, RO_GJK_CLOSEST_POINTS = 1, RO_GJK_NO_INTERSECTION = 2 };
typedef struct { int kind; v3 p1, p2, dir; int unsure; } GjkResult; /* p1, p2 (both in frame 1), dir = unit vector from shape 1 towards shape 2;
   unsure = the loop stalled (rounding) before any direction separated the shapes: the distance is an upper bound, they may overlap */

/* gjk::closest_points(pos12, g1, g2, max_dist, exact_dist = true, simplex) */
static inline GjkResult gjk_closest_points(const SmShape *s1, const SmShape *s2, pose pos12, float max_dist, GjkSimplex *sx) {
    const float eps_tol = RO_GJK_EPS_TOL, eps_rel = 1.0918301e-3f /* sqrt(eps_tol) */;
    GjkResult r; r.kind = RO_GJK_INTERSECTION; r.p1 = r.p2 = V3(0, 0, 0); r.dir = V3(1, 0, 0); r.unsure = 0;
    float last_min_bound = -FLT_MAX;
    int inside;
    v3 proj = sx_project_and_reduce(sx, &inside);
    float plen = vlen(proj);
    if (!(plen > 0.0f)) return r;
    v3 old_dir = vmul(proj, -1.0f / plen);
    float max_bound = FLT_MAX;
    v3 dir;
    for (int niter = 0; niter < RO_GJK_MAX_ITERS; ++niter) {
        float old_max_bound = max_bound;
        plen = vlen(proj);
        if (!(plen > eps_tol)) return r; /* the origin is on the simplex */
        dir = vmul(proj, -1.0f / plen); max_bound = plen;
        if (max_bound >= old_max_bound) { /* the previous projection was better */
            r.kind = RO_GJK_CLOSEST_POINTS; sx_result(sx, 1, &r.p1, &r.p2); r.dir = old_dir; r.unsure = !(last_min_bound > 0.0f); return r;
        }
        CsoPt w = cso_support(s1, s2, pos12, dir);
        float min_bound = -vdot(dir, w.p);
        if (min_bound > max_dist) { r.kind = RO_GJK_NO_INTERSECTION; r.dir = dir; return r; }
        if (max_bound - min_bound <= eps_rel * max_bound) { r.kind = RO_GJK_CLOSEST_POINTS; sx_result(sx, 0, &r.p1, &r.p2); r.dir = dir; return r; }
        last_min_bound = min_bound;
        if (!sx_add_point(sx, w)) { r.kind = RO_GJK_CLOSEST_POINTS; sx_result(sx, 0, &r.p1, &r.p2); r.dir = dir; r.unsure = !(min_bound > 0.0f); return r; }
        old_dir = dir;
        proj = sx_project_and_reduce(sx, &inside);
        if (inside) { /* simplex.dimension() == DIM: the origin is inside the tetrahedron, or the bound says it cannot be */
            if (min_bound >= eps_tol) { r.kind = RO_GJK_CLOSEST_POINTS; sx_result(sx, 1, &r.p1, &r.p2); r.dir = old_dir; return r; }
            return r;
        }
    }
    r.kind = RO_GJK_NO_INTERSECTION; r.dir = V3(1, 0, 0);
    return r;
}

/* ---- expanding polytope: penetration depth, normal and witness points once GJK found the origin inside the CSO ---- */
typedef struct { unsigned char a, b, c, alive; v3 n; float d; } EpaFace;
typedef struct { CsoPt v[RO_EPA_MAXV]; int nv; EpaFace f[RO_EPA_MAXF]; int nf; } EpaPoly;

/* face (a, b, c), counter-clockwise seen from outside: unit normal and plane offset; 0 when it has no area */
static inline int epa_face_init(const EpaPoly *P, EpaFace *f, int a, int b, int c) {
    v3 pa = P->v[a].p;
    v3 nrm = vcross(vsub(P->v[b].p, pa), vsub(P->v[c].p, pa));
    float len = vlen(nrm);
    f->a = (unsigned char)a; f->b = (unsigned char)b; f->c = (unsigned char)c; f->alive = 1;
    if (!(len > 1.0e-18f)) { f->n = V3(0, 0, 0); f->d = FLT_MAX; return 0; }
    f->n = vmul(nrm, 1.0f / len);
    f->d = vdot(f->n, pa);
    return 1;
}
static inline int epa_add_face(EpaPoly *P, int a, int b, int c) { /* the lowest free slot; -1 = none left / degenerate */
    int slot = -1;
    for (int i = 0; i < P->nf; ++i) if (!P->f[i].alive) { slot = i; break; }
    if (slot < 0) { if (P->nf >= RO_EPA_MAXF) return -1; slot = P->nf++; }
    if (!epa_face_init(P, &P->f[slot], a, b, c)) { P->f[slot].alive = 0; return -1; }
    return slot;
}
/* raise the simplex GJK stopped with to a tetrahedron (it holds the origin on a vertex, an edge or a face) */
static inline int epa_blow_up(const SmShape *s1, const SmShape *s2, pose pos12, EpaPoly *P) {
    static const float AX[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    if (P->nv == 1) {
        for (int k = 0; k < 6 && P->nv == 1; ++k) {
            CsoPt w = cso_support(s1, s2, pos12, V3(AX[k][0], AX[k][1], AX[k][2]));
            if (vlen2(vsub(w.p, P->v[0].p)) > RO_GJK_EPS_TOL) P->v[P->nv++] = w;
        }
        if (P->nv == 1) return 0;
    }
    if (P->nv == 2) {
        v3 d = vsub(P->v[1].p, P->v[0].p);
        float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
        v3 e = (ax <= ay && ax <= az) ? V3(1, 0, 0) : (ay <= az ? V3(0, 1, 0) : V3(0, 0, 1));
        v3 u = vcross(d, e);
        float dl = vlen(d);
        v3 dn = vmul(d, 1.0f / dl);
        for (int k = 0; k < 6 && P->nv == 2; ++k) {
            CsoPt w = cso_support(s1, s2, pos12, u);
            if (vlen2(vcross(vsub(w.p, P->v[0].p), d)) > RO_GJK_EPS_TOL * vlen2(d)) P->v[P->nv++] = w;
            /* next direction: u turned by 60 degrees about d */
            u = vadd(vmul(u, 0.5f), vmul(vcross(dn, u), 0.86602540378f));
        }
        if (P->nv == 2) return 0;
    }
    if (P->nv == 3) {
        v3 nrm = vcross(vsub(P->v[1].p, P->v[0].p), vsub(P->v[2].p, P->v[0].p));
        float len = vlen(nrm);
        if (!(len > 0.0f)) return 0;
        nrm = vmul(nrm, 1.0f / len);
        CsoPt w = cso_support(s1, s2, pos12, nrm);
        if (!(fabsf(vdot(vsub(w.p, P->v[0].p), nrm)) > RO_GJK_EPS_TOL)) {
            w = cso_support(s1, s2, pos12, vneg(nrm));
            if (!(fabsf(vdot(vsub(w.p, P->v[0].p), nrm)) > RO_GJK_EPS_TOL)) return 0;
        }
        P->v[P->nv++] = w;
    }
    return 1;
}
/* returns 1 with the witness points (frame 1) and the unit normal from shape 1 towards shape 2 (the direction shape 2 must be
 * pushed to separate); 0 when no polytope could be built */
static inline int epa_closest_points(const SmShape *s1, const SmShape *s2, pose pos12, const GjkSimplex *sx, v3 *p1, v3 *p2, v3 *normal) {
    EpaPoly P;
    P.nv = sx->n; P.nf = 0;
    for (int i = 0; i < sx->n; ++i) P.v[i] = sx->v[i];
    if (P.nv < 4 && !epa_blow_up(s1, s2, pos12, &P)) return 0;
    {   /* orient: vertex 3 on the inner side of (0, 1, 2) */
        v3 a = P.v[0].p;
        float vol = vdot(vcross(vsub(P.v[1].p, a), vsub(P.v[2].p, a)), vsub(P.v[3].p, a));
        if (vol == 0.0f) return 0;
        if (vol > 0.0f) { CsoPt t = P.v[1]; P.v[1] = P.v[2]; P.v[2] = t; }
        if (epa_add_face(&P, 0, 1, 2) < 0 || epa_add_face(&P, 0, 2, 3) < 0 || epa_add_face(&P, 0, 3, 1) < 0 || epa_add_face(&P, 1, 3, 2) < 0) return 0;
    }
    int best = 0;
    EpaFace good = P.f[0]; good.alive = 0; /* the closest face of the last sound polytope */
    for (int iter = 0; iter < 64; ++iter) {
        best = -1;
        for (int i = 0; i < P.nf; ++i) if (P.f[i].alive && (best < 0 || P.f[i].d < P.f[best].d)) best = i;
        if (best < 0) return 0;
        /* a growing convex polytope's distance to the origin cannot shrink: when it does, the last expansion went through a sliver
         * whose rounding turned a face inside out — answer with the closest face from before it */
        if (good.alive && P.f[best].d < good.d - 1.0e-6f) break;
        good = P.f[best];
        EpaFace bf = P.f[best];
        CsoPt w = cso_support(s1, s2, pos12, bf.n);
        if (vdot(w.p, bf.n) - bf.d < RO_EPA_EPS_TOL || P.nv >= RO_EPA_MAXV) break;
        /* faces that see the new point die; their edges that are not shared by two dying faces form the horizon */
        unsigned char ea[RO_EPA_MAXE], eb[RO_EPA_MAXE]; int ne = 0, overflow = 0, removed = 0;
        for (int i = 0; i < P.nf; ++i) {
            EpaFace *g = &P.f[i];
            if (!g->alive || !(vdot(g->n, vsub(w.p, P.v[g->a].p)) > 0.0f)) continue;
            g->alive = 0; ++removed;
            unsigned char va[3] = {g->a, g->b, g->c}, vb[3] = {g->b, g->c, g->a};
            for (int k = 0; k < 3; ++k) {
                int found = -1;
                for (int q = 0; q < ne; ++q) if (ea[q] == vb[k] && eb[q] == va[k]) { found = q; break; }
                if (found >= 0) { ea[found] = ea[ne - 1]; eb[found] = eb[ne - 1]; --ne; }
                else if (ne < RO_EPA_MAXE) { ea[ne] = va[k]; eb[ne] = vb[k]; ++ne; }
                else overflow = 1;
            }
        }
        if (removed == 0 || overflow) break; /* (numerically) on the hull already, or out of room: the closest face stands */
        int k = P.nv; P.v[P.nv++] = w;
        int failed = 0;
        for (int q = 0; q < ne; ++q) if (epa_add_face(&P, ea[q], eb[q], k) < 0) failed = 1;
        if (failed) break; /* out of face slots or a face without area: the closest face from before stands */
    }
    EpaFace bf = good;
    /* the origin projected on the closest face: barycentric coordinates, clamped to the triangle */
    float bc[4];
    v3 a = vsub(P.v[bf.a].p, vmul(bf.n, bf.d)), b = vsub(P.v[bf.b].p, vmul(bf.n, bf.d)), c = vsub(P.v[bf.c].p, vmul(bf.n, bf.d));
    sx_proj_tri(a, b, c, bc);
    *p1 = vadd(vadd(vmul(P.v[bf.a].o1, bc[0]), vmul(P.v[bf.b].o1, bc[1])), vmul(P.v[bf.c].o1, bc[2]));
    *p2 = vadd(vadd(vmul(P.v[bf.a].o2, bc[0]), vmul(P.v[bf.b].o2, bc[1])), vmul(P.v[bf.c].o2, bc[2]));
    *normal = bf.n;
    return 1;
}

/* query::details::contact_support_map_support_map_with_params: 1 = a point pair (closest points within `prediction`, or the
 * penetration's witness points) with the unit normal from 1 to 2; 0 = further apart than `prediction` (*normal = the direction GJK
 * stopped with: a cache for the next call) */
static inline int sm_contact(const SmShape *s1, const SmShape *s2, pose pos12, float prediction, v3 init_dir, v3 *p1, v3 *p2, v3 *normal) {
    v3 dir = init_dir;
    float dl = vlen(dir);
    if (dl > FLT_EPSILON) dir = vmul(dir, 1.0f / dl);
    else {
        float tl = vlen(pos12.t);
        dir = tl > FLT_EPSILON ? vmul(pos12.t, 1.0f / tl) : V3(1, 0, 0);
    }
    GjkSimplex sx;
    sx_reset(&sx, cso_support(s1, s2, pos12, dir));
    GjkResult r = gjk_closest_points(s1, s2, pos12, prediction, &sx);
    if (r.kind == RO_GJK_CLOSEST_POINTS && r.unsure) {
        /* GJK stalled within rounding of the origin without a separating direction: the polytope pass settles it either way (its
         * closest face carries a signed distance: negative plane offsets are separations) */
        v3 q1, q2, qn;
        if (epa_closest_points(s1, s2, pos12, &sx, &q1, &q2, &qn)) {
            if (vdot(vsub(q2, q1), qn) > prediction) { *normal = qn; return 0; }
            *p1 = q1; *p2 = q2; *normal = qn; return 1;
        }
    }
    if (r.kind == RO_GJK_CLOSEST_POINTS) { *p1 = r.p1; *p2 = r.p2; *normal = r.dir; return 1; }
    if (r.kind == RO_GJK_NO_INTERSECTION) { *normal = r.dir; return 0; }
    if (epa_closest_points(s1, s2, pos12, &sx, p1, p2, normal)) return 1;
    *normal = V3(1, 0, 0);
    return 0;
}
/* distance between the core shapes (0 when they overlap) and the unit direction from 1 to 2: the lower bound the CCD pass advances on */
static inline float sm_distance(const SmShape *s1, const SmShape *s2, pose pos12, v3 *n1) {
    v3 p1, p2;
    float tl = vlen(pos12.t);
    v3 dir = tl > FLT_EPSILON ? vmul(pos12.t, 1.0f / tl) : V3(1, 0, 0);
    GjkSimplex sx;
    sx_reset(&sx, cso_support(s1, s2, pos12, dir));
    GjkResult r = gjk_closest_points(s1, s2, pos12, FLT_MAX, &sx);
    if (r.kind != RO_GJK_CLOSEST_POINTS) { *n1 = V3(0, 1, 0); return -1.0f; }
    p1 = r.p1; p2 = r.p2; *n1 = r.dir;
    return vdot(vsub(p2, p1), r.dir);
}
/* intersection_test of two support-mapped shapes (sensor pairs): the cores within the sum of the border radii */
static inline int sm_intersects(const SmShape *s1, const SmShape *s2, pose pos12) {
    v3 n;
    float d = sm_distance(s1, s2, pos12, &n);
    return d <= sm_border_radius(s1) + sm_border_radius(s2);
}

/* ---- polygonal feature maps ---- */
typedef struct { v3 v[4]; uint32_t vid[4], eid[4], fid; int nv; } PolyFeat;

/* the direction a cap's square is turned to: towards `hint` (the contact point on this shape) when it is off the axis, else along
 * the horizontal part of `dir`, else +x */
static inline void cap_dir2(v3 dir, v3 hint, float *cx, float *cz) {
    float hn = sqrtf(hint.x * hint.x + hint.z * hint.z);
    if (hn > 1.0e-6f) { *cx = hint.x / hn; *cz = hint.z / hn; return; }
    float dn = sqrtf(dir.x * dir.x + dir.z * dir.z);
    if (dn > FLT_EPSILON) { *cx = dir.x / dn; *cz = dir.z / dn; return; }
    *cx = 1.0f; *cz = 0.0f;
}
/* PolygonalFeatureMap::local_support_feature (feature ids as in parry's cylinder.rs / cone.rs: curved part = segment 0 with end
 * points 1 and 11; bottom cap: vertices 1, 3, 5, 7, edges 2, 4, 6, 8, face 9; top cap: the same + 10) */
static inline void sm_support_feature(const SmShape *s, v3 dir, v3 hint, PolyFeat *out) {
    if (s->shape == RO_SHAPE_CUBOID) {
        PolyFace f = cuboid_support_face(s->he, dir);
        for (int i = 0; i < 4; ++i) { out->v[i] = f.vertices[i]; out->vid[i] = f.vids[i]; out->eid[i] = f.eids[i]; }
        out->fid = f.fid; out->nv = 4;
        return;
    }
    if (s->shape == RO_SHAPE_CAPSULE) { /* Segment: the segment itself */
        v3 e = capsule_axis_dir(s->axis);
        out->v[0] = vmul(e, -s->he.x); out->v[1] = vmul(e, s->he.x); out->v[2] = out->v[1]; out->v[3] = out->v[1];
        out->vid[0] = 0; out->vid[1] = 2; out->vid[2] = 2; out->vid[3] = 2;
        for (int i = 0; i < 4; ++i) out->eid[i] = 1;
        out->fid = 0; out->nv = 2;
        return;
    }
    if (s->shape == RO_SHAPE_TRIANGLE) { /* Triangle: PolygonalFeature::from(triangle) — the face itself whatever the direction (vertex ids 0, 2, 4, edge ids 1, 3, 5) */
        for (int i = 0; i < 4; ++i) { int k = i < 3 ? i : 2; out->v[i] = s->tri[k]; out->vid[i] = 2u * (uint32_t)k; out->eid[i] = 2u * (uint32_t)k + 1u; }
        out->fid = 0; out->nv = 3;
        return;
    }
    if (s->shape == RO_SHAPE_CONVEX_POLYHEDRON) { /* ConvexPolyhedron: the face whose normal is closest to dir (the first one), its first four vertices */
        const RoPolyhedron *P = s->poly;
        int best = 0; float bd = vdot(P->fnormal[0], dir);
        for (int f = 1; f < P->nf; ++f) { float x = vdot(P->fnormal[f], dir); if (x > bd) { bd = x; best = f; } }
        int cnt = P->fcount[best] < 4 ? P->fcount[best] : 4, first = P->ffirst[best];
        for (int i = 0; i < 4; ++i) {
            int k = first + (i < cnt ? i : cnt - 1);
            out->v[i] = P->pts[P->loop_v[k]]; out->vid[i] = (uint32_t)P->loop_v[k]; out->eid[i] = RO_FID_EDGE | (uint32_t)P->loop_e[k];
        }
        out->fid = RO_FID_FACE | (uint32_t)best; out->nv = cnt;
        return;
    }
    float r = s->radius, hh = s->he.y;
    int curved = s->shape == RO_SHAPE_CYLINDER ? (fabsf(dir.y) < 0.5f) : (dir.y > 0.0f);
    if (curved) {
        float dn = sqrtf(dir.x * dir.x + dir.z * dir.z), cx = 1.0f, cz = 0.0f;
        if (dn > FLT_EPSILON) { cx = dir.x / dn; cz = dir.z / dn; }
        out->v[0] = V3(cx * r, -hh, cz * r);
        out->v[1] = s->shape == RO_SHAPE_CYLINDER ? V3(cx * r, hh, cz * r) : V3(0.0f, hh, 0.0f);
        out->v[2] = out->v[1]; out->v[3] = out->v[1];
        out->vid[0] = 1; out->vid[1] = 11; out->vid[2] = 11; out->vid[3] = 11;
        for (int i = 0; i < 4; ++i) out->eid[i] = 0;
        out->fid = 0; out->nv = 2;
        return;
    }
    float cx, cz;
    cap_dir2(dir, hint, &cx, &cz);
    float y = s->shape == RO_SHAPE_CYLINDER ? copysignf(hh, dir.y) : -hh;
    out->v[0] = V3(cx * r, y, cz * r);
    out->v[1] = V3(-cz * r, y, cx * r);
    out->v[2] = V3(-cx * r, y, -cz * r);
    out->v[3] = V3(cz * r, y, -cx * r);
    uint32_t base = y < 0.0f ? 0u : 10u;
    for (int i = 0; i < 4; ++i) { out->vid[i] = base + 1u + 2u * (uint32_t)i; out->eid[i] = base + 2u + 2u * (uint32_t)i; }
    out->fid = base + 9u; out->nv = 4;
}

/* query::details::clip_segment_segment: the overlap of two (nearly parallel) segments as two point pairs */
static inline int clip_segment_segment(v3 a1, v3 b1, v3 a2, v3 b2, v3 out[4]) {
    v3 t1 = vsub(b1, a1);
    float sq = vlen2(t1);
    float r20 = vdot(vsub(a2, a1), t1), r21 = vdot(vsub(b2, a1), t1);
    if (r21 < r20) { float t = r20; r20 = r21; r21 = t; v3 p = a2; a2 = b2; b2 = p; }
    if (r20 > sq || 0.0f > r21) return 0;
    v3 d2 = vsub(b2, a2);
    if (r20 > 0.0f) { out[0] = vadd(a1, vmul(t1, r20 * ro_inv(sq))); out[1] = a2; }
    else { out[0] = a1; out[1] = vadd(a2, vmul(d2, (0.0f - r20) * ro_inv(r21 - r20))); }
    if (r21 < sq) { out[2] = vadd(a1, vmul(t1, r21 * ro_inv(sq))); out[3] = b2; }
    else { out[2] = b1; out[3] = vadd(a2, vmul(d2, (sq - r20) * ro_inv(r21 - r20))); }
    return 1;
}

/* PolygonalFeature::contacts: f2 is expressed in frame 1 already; points are appended to m */
static inline void contacts_features(pose pos12, const PolyFeat *f1, v3 sep, const PolyFeat *f2, Manifold *m) {
    v3 basis[2]; orthonormal_basis(sep, basis);
    float q1[4][2], q2[4][2];
    for (int i = 0; i < 4; ++i) {
        q1[i][0] = vdot(f1->v[i], basis[0]); q1[i][1] = vdot(f1->v[i], basis[1]);
        q2[i][0] = vdot(f2->v[i], basis[0]); q2[i][1] = vdot(f2->v[i], basis[1]);
    }
    if (f1->nv == 2 && f2->nv == 2) { /* contacts_edge_edge */
        float t1x = q1[1][0] - q1[0][0], t1y = q1[1][1] - q1[0][1], t2x = q2[1][0] - q2[0][0], t2y = q2[1][1] - q2[0][1];
        float l1 = sqrtf(t1x * t1x + t1y * t1y), l2 = sqrtf(t2x * t2x + t2y * t2y);
        if (l1 > FLT_EPSILON && l2 > FLT_EPSILON) {
            float c = (t1x / l1) * (t2x / l2) + (t1y / l1) * (t2y / l2);
            if (!(fabsf(c) >= 0.92387953251f)) { /* not parallel (COS_FRAC_PI_8): the closest points of the two segments */
                float s, t;
                closest_points_segment_segment(V3(q1[0][0], q1[0][1], 0.0f), V3(q1[1][0], q1[1][1], 0.0f), V3(q2[0][0], q2[0][1], 0.0f), V3(q2[1][0], q2[1][1], 0.0f), &s, &t);
                v3 p1 = vadd(vmul(f1->v[0], 1.0f - s), vmul(f1->v[1], s));
                v3 p2 = vadd(vmul(f2->v[0], 1.0f - t), vmul(f2->v[1], t));
                manifold_push(m, p1, pose_itp(pos12, p2), f1->eid[0], f2->eid[0], vdot(vsub(p2, p1), sep));
                return;
            }
        }
        v3 c4[4];
        if (clip_segment_segment(f1->v[0], f1->v[1], f2->v[0], f2->v[1], c4)) {
            manifold_push(m, c4[0], pose_itp(pos12, c4[1]), f1->vid[0], f2->vid[0], vdot(vsub(c4[1], c4[0]), sep));
            manifold_push(m, c4[2], pose_itp(pos12, c4[3]), f1->vid[1], f2->vid[1], vdot(vsub(c4[3], c4[2]), sep));
        }
        return;
    }
#define PERP(ax, ay, bx, by) ((ax) * (by) - (ay) * (bx))
    if (f2->nv > 2) { /* vertices of feature 1 inside feature 2 */
        v3 normal2_1 = vcross(vsub(f2->v[2], f2->v[1]), vsub(f2->v[0], f2->v[1]));
        float denom = vdot(normal2_1, sep);
        if (!ro_relative_eq0(denom)) {
            int last = f2->nv - 1;
            for (int i = 0; i < f1->nv; ++i) {
                float px = q1[i][0], py = q1[i][1];
                float sign = PERP(q2[0][0] - q2[last][0], q2[0][1] - q2[last][1], px - q2[last][0], py - q2[last][1]);
                int outside = 0;
                for (int j = 0; j < last; ++j) {
                    float ns = PERP(q2[j + 1][0] - q2[j][0], q2[j + 1][1] - q2[j][1], px - q2[j][0], py - q2[j][1]);
                    if (sign == 0.0f) sign = ns;
                    else if (sign * ns < 0.0f) { outside = 1; break; }
                }
                if (outside) continue;
                float dist = vdot(vsub(f2->v[0], f1->v[i]), normal2_1) / denom;
                manifold_push(m, f1->v[i], pose_itp(pos12, vadd(f1->v[i], vmul(sep, dist))), f1->vid[i], f2->fid, dist);
            }
        }
    }
    if (f1->nv > 2) { /* vertices of feature 2 inside feature 1 */
        v3 normal1 = vcross(vsub(f1->v[2], f1->v[1]), vsub(f1->v[0], f1->v[1]));
        float denom = -vdot(normal1, sep);
        if (!ro_relative_eq0(denom)) {
            int last = f1->nv - 1;
            for (int i = 0; i < f2->nv; ++i) {
                float px = q2[i][0], py = q2[i][1];
                float sign = PERP(q1[0][0] - q1[last][0], q1[0][1] - q1[last][1], px - q1[last][0], py - q1[last][1]);
                int outside = 0;
                for (int j = 0; j < last; ++j) {
                    float ns = PERP(q1[j + 1][0] - q1[j][0], q1[j + 1][1] - q1[j][1], px - q1[j][0], py - q1[j][1]);
                    if (sign == 0.0f) sign = ns;
                    else if (sign * ns < 0.0f) { outside = 1; break; }
                }
                if (outside) continue;
                float dist = vdot(vsub(f1->v[0], f2->v[i]), normal1) / denom;
                manifold_push(m, vsub(f2->v[i], vmul(sep, dist)), pose_itp(pos12, f2->v[i]), f1->fid, f2->vid[i], dist);
            }
        }
    }
#undef PERP
    int ne1 = f1->nv == 2 ? 1 : f1->nv, ne2 = f2->nv == 2 ? 1 : f2->nv; /* a segment is one edge */
    for (int j = 0; j < ne2; ++j) {
        int j1 = (j + 1) % f2->nv;
        float e2[2][2] = {{q2[j][0], q2[j][1]}, {q2[j1][0], q2[j1][1]}};
        for (int i = 0; i < ne1; ++i) {
            int i1 = (i + 1) % f1->nv;
            float e1[2][2] = {{q1[i][0], q1[i][1]}, {q1[i1][0], q1[i1][1]}};
            float s, t;
            if (closest_points_line2d(e1, e2, &s, &t) && s > 0.0f && s < 1.0f && t > 0.0f && t < 1.0f) {
                v3 p1 = vadd(vmul(f1->v[i], 1.0f - s), vmul(f1->v[i1], s));
                v3 p2 = vadd(vmul(f2->v[j], 1.0f - t), vmul(f2->v[j1], t));
                manifold_push(m, p1, pose_itp(pos12, p2), f1->eid[i], f2->eid[j], vdot(vsub(p2, p1), sep));
            }
        }
    }
}

static inline void manifold_match_contacts(Manifold *m, const TrackedContact *old, int nold) {
    for (int i = 0; i < m->npoints; ++i)
        for (int j = 0; j < nold; ++j)
            if (m->points[i].fid1 == old[j].fid1 && m->points[i].fid2 == old[j].fid2)
                m->points[i].data = old[j].data;
}

/* contact_manifold_pfm_pfm */
static inline void manifold_pfm_pfm(pose pos12, const SmShape *s1, const SmShape *s2, float prediction, Manifold *m) {
    if (manifold_try_update_contacts(m, pos12)) return;
    float b1 = sm_border_radius(s1), b2 = sm_border_radius(s2);
    v3 p1, p2, n1;
    int hit = sm_contact(s1, s2, pos12, prediction + b1 + b2, m->local_n1, &p1, &p2, &n1);
    TrackedContact old[RO_MAX_MANIFOLD_PTS]; int nold = m->npoints;
    memcpy(old, m->points, sizeof(old));
    m->npoints = 0;
    if (!hit) { m->local_n1 = n1; return; } /* the separating direction is kept as the next call's first guess */
    v3 n2 = qrot_inv(pos12.r, vneg(n1));
    float dist = vdot(vsub(p2, p1), n1);
    PolyFeat f1, f2;
    sm_support_feature(s1, n1, p1, &f1);
    sm_support_feature(s2, n2, pose_itp(pos12, p2), &f2);
    for (int i = 0; i < 4; ++i) f2.v[i] = pose_tp(pos12, f2.v[i]);
    contacts_features(pos12, &f1, n1, &f2, m);
    if (m->npoints == 0) manifold_push(m, p1, pose_itp(pos12, p2), RO_FID_UNKNOWN, RO_FID_UNKNOWN, dist);
    if (b1 != 0.0f || b2 != 0.0f)
        for (int i = 0; i < m->npoints; ++i) {
            m->points[i].local_p1 = vadd(m->points[i].local_p1, vmul(n1, b1));
            m->points[i].local_p2 = vadd(m->points[i].local_p2, vmul(n2, b2));
            m->points[i].dist -= b1 + b2;
        }
    m->local_n1 = n1; m->local_n2 = n2;
    manifold_match_contacts(m, old, nold);
}

/* PointQuery::project_local_point(pt, solid = false) of a cylinder / cone: the projection and whether pt is inside */
static inline v3 sm_project_point(const SmShape *s, v3 pt, int *inside) {
    float r = s->radius, hh = s->he.y;
    float pd = sqrtf(pt.x * pt.x + pt.z * pt.z);
    float dx = 1.0f, dz = 0.0f;
    if (pd > FLT_EPSILON) { dx = pt.x / pd; dz = pt.z / pd; }
    float sx = dx * r, sz = dz * r;
    *inside = 0;
    if (s->shape == RO_SHAPE_CYLINDER) {
        if (pt.y >= -hh && pt.y <= hh && pd <= r) {
            *inside = 1;
            float to_top = hh - pt.y, to_bottom = pt.y - (-hh), to_side = r - pd;
            if (to_top < to_bottom && to_top < to_side) return V3(pt.x, hh, pt.z);
            if (to_bottom < to_top && to_bottom < to_side) return V3(pt.x, -hh, pt.z);
            return V3(sx, pt.y, sz);
        }
        if (pt.y > hh) return pd <= r ? V3(pt.x, hh, pt.z) : V3(sx, hh, sz);
        if (pt.y < -hh) return pd <= r ? V3(pt.x, -hh, pt.z) : V3(sx, -hh, sz);
        return V3(sx, pt.y, sz);
    }
    /* cone */
    v3 on_basis = V3(pt.x, -hh, pt.z);
    if (pt.y < -hh && pd <= r) return on_basis;
    v3 apex = V3(0.0f, hh, 0.0f), rim = V3(sx, -hh, sz);
    v3 sd = vsub(rim, apex);
    v3 proj = vadd(apex, vmul(sd, ro_clampf(vdot(vsub(pt, apex), sd) / vdot(sd, sd), 0.0f, 1.0f))); /* Segment::project_local_point */
    v3 apex_to_centre = V3(0.0f, -2.0f * hh, 0.0f);
    if (pt.y >= -hh && pt.y <= hh && vdot(vcross(sd, vsub(pt, apex)), vcross(sd, apex_to_centre)) >= 0.0f) {
        *inside = 1;
        if (vlen2(vsub(proj, pt)) > vlen2(vsub(on_basis, pt))) return on_basis;
        return proj;
    }
    return proj;
}
/* contact_manifold_convex_ball with shape1 = a cylinder / cone / convex polyhedron; flipped = the ball is collider 1 */
static inline void manifold_sm_ball(pose pos12, const SmShape *s1, float r2, float prediction, Manifold *m, int flipped) {
    v3 pt = pos12.t;
    if (s1->shape == RO_SHAPE_CONVEX_POLYHEDRON || s1->shape == RO_SHAPE_TRIANGLE || s1->border > 0.0f) {
        /* ConvexPolyhedron / Triangle / RoundShape::project_local_point = local_point_projection_on_support_map: GJK against the point, the polytope
         * pass when the point is inside — the support-mapped contact query with the ball's centre as second shape; a round shape's
         * border moves the projection out along the normal */
        SmShape centre; centre.shape = RO_SHAPE_BALL; centre.he = V3(0, 0, 0); centre.radius = 0.0f; centre.axis = 1; centre.poly = NULL; centre.border = 0.0f;
        const float b1 = s1->border;
        v3 p1, p2, n1;
        if (!sm_contact(s1, &centre, pos12, (r2 + prediction) + b1, V3(0, 0, 0), &p1, &p2, &n1)) { m->npoints = 0; return; }
        float dist = vdot(vsub(p2, p1), n1);
        if (b1 != 0.0f) { p1 = vadd(p1, vmul(n1, b1)); dist = dist - b1; }
        if (dist <= r2 + prediction) {
            v3 n2 = qrot_inv(pos12.r, vneg(n1));
            v3 q2 = vmul(n2, r2);
            float d = dist - r2;
            v3 a = flipped ? q2 : p1, b = flipped ? p1 : q2;
            if (m->npoints != 1) { m->npoints = 0; manifold_push(m, a, b, RO_FID_UNKNOWN, RO_FID_UNKNOWN, d); }
            else { m->points[0].local_p1 = a; m->points[0].local_p2 = b; m->points[0].dist = d; }
            if (flipped) { m->local_n1 = n2; m->local_n2 = n1; } else { m->local_n1 = n1; m->local_n2 = n2; }
        } else m->npoints = 0;
        return;
    }
    int inside;
    v3 proj = sm_project_point(s1, pt, &inside);
    v3 dpos = vsub(pt, proj);
    float dist = vlen(dpos);
    if (!(dist > 0.0f)) return;
    v3 n1 = vmul(dpos, 1.0f / dist);
    if (inside) { n1 = vneg(n1); dist = -dist; }
    if (dist <= r2 + prediction) {
        v3 n2 = qrot_inv(pos12.r, vneg(n1));
        v3 p2 = vmul(n2, r2);
        float d = dist - r2;
        v3 a = flipped ? p2 : proj, b = flipped ? proj : p2;
        if (m->npoints != 1) { m->npoints = 0; manifold_push(m, a, b, RO_FID_UNKNOWN, RO_FID_UNKNOWN, d); }
        else { m->points[0].local_p1 = a; m->points[0].local_p2 = b; m->points[0].dist = d; }
        if (flipped) { m->local_n1 = n2; m->local_n2 = n1; } else { m->local_n1 = n1; m->local_n2 = n2; }
    } else m->npoints = 0;
}
/* contact_manifold_halfspace_pfm with shape 2 = a cylinder / cone: its support feature toward the plane */
static inline void manifold_halfspace_sm(pose pos12, v3 normal1, const SmShape *s2, float prediction, Manifold *m, int flipped) {
    v3 normal1_2 = qrot_inv(pos12.r, normal1);
    PolyFeat f;
    sm_support_feature(s2, vneg(normal1_2), V3(0, 0, 0), &f);
    const float border = s2->border;
    TrackedContact old[RO_MAX_MANIFOLD_PTS]; int nold = m->npoints;
    memcpy(old, m->points, sizeof(old));
    m->npoints = 0;
    for (int i = 0; i < f.nv; ++i) {
        v3 vtx2_1 = pose_tp(pos12, f.v[i]);
        float dist_to_plane = vdot(vtx2_1, normal1);
        if (dist_to_plane - border <= prediction) {
            v3 p1 = vsub(vtx2_1, vmul(normal1, dist_to_plane));
            v3 p2 = border != 0.0f ? vsub(f.v[i], vmul(normal1_2, border)) : f.v[i];
            if (flipped) manifold_push(m, p2, p1, f.vid[i], 0u, dist_to_plane - border);
            else manifold_push(m, p1, p2, 0u, f.vid[i], dist_to_plane - border);
        }
    }
    if (flipped) { m->local_n1 = vneg(normal1_2); m->local_n2 = normal1; }
    else { m->local_n1 = normal1; m->local_n2 = vneg(normal1_2); }
    manifold_match_contacts(m, old, nold);
}
#endif
