// rp_convex.h — support-mapped shapes on the device: cylinders and cones (included by rp_narrowphase.hip after the cuboid / capsule
// generators, whose helpers it reuses).
//
// rapier hands every pair with a Cylinder or a Cone (ColliderBuilder::cylinder / cone, /root/reference/src/geometry/collider.rs:770,
// :789) to parry3d's DefaultQueryDispatcher::contact_manifolds (call site /root/reference/src/geometry/narrow_phase/pair_update.rs:
// 323-330): contact_manifold_pfm_pfm (try_update_contacts -> GJK closest points, an expanding-polytope pass when the shapes overlap ->
// local_support_feature of both shapes -> PolygonalFeature::contacts -> border radii -> match_contacts), contact_manifold_convex_ball
// (point projection) and contact_manifold_halfspace_pfm.  parry3d 0.30.2 is not under /root/reference: what is implemented is the
// crate's published algorithm (support functions, projections, the feature approximations — a cap is a square inscribed in the
// circle, the curved part one segment —, the GJK loop and its Voronoi simplex), the polytope pass is our own, and the one behaviour
// pinned by the reference's tests is kept: a cap's square turns toward the contact point
// (crates/rapier3d/tests/issue_810_cubes_thin_cylinder_tunnel.rs:1-8).  Manifold-level parity with parry is UNPINNED.
//
// One lane runs the whole query for its pair: GJK's simplex lives in registers, the polytope (40 vertices, 80 faces) in the lane's
// scratch memory — which is why these functions are only compiled into the CONVEX instantiations of the narrow-phase, sensor and
// continuous-collision kernels, launched for worlds that hold such a shape (DevWorld::has_convex): every other world keeps the
// kernels it had.  Every function performs the operations of the checker's restatement in the same order (compared bit for bit).
#pragma once

#define RP_GJK_EPS_TOL 1.1920929e-6f // gjk::eps_tol() = 10 * f32::EPSILON
#define RP_EPA_EPS_TOL 1.1920929e-5f
#define RP_GJK_REL_TOL 1.0e-5f       // add_point: sine of the smallest angle a new vertex must add
#define RP_GJK_MAX_ITERS 100
#define RP_EPA_MAXV 40
#define RP_EPA_MAXF 80
#define RP_EPA_MAXE 48

// c_shape / c_he of rp_world.h: cuboid half extents | capsule (half height, radius, axis) | ball radius | cylinder / cone (axis Y):
// he = (radius, half_height, radius) = the half extents of the local AABB
// a convex polyhedron (c_he = its local box's half extents, w = its row in cv_hdr as bits): points, face normals, per face {first loop
// entry, entries}, loop entries {vertex, edge} — pointers into the world's cv_* tables
// a round shape (RP_SHAPE_ROUND_*, parry RoundShape<S>): `shape` is its inner shape, `border` its border radius (c_mat.w)
struct SmShape { int shape; V3 he; float radius; int axis; const float4 *pts; const float4 *fn; const int2 *fl; const int2 *loop; int npts, nfaces; float border;
                 V3 tri[3]; }; // tri: RP_SHAPE_TRIANGLE (a triangle of a mesh collider, rp_composite.h): its vertices
RP_DEV int sm_core_shape(int sh) { return (sh >= RP_SHAPE_ROUND_CUBOID && sh <= RP_SHAPE_ROUND_CONVEX_POLYHEDRON) ? (sh == RP_SHAPE_ROUND_CUBOID ? RP_SHAPE_CUBOID : sh - RP_SHAPE_ROUND_CYLINDER + RP_SHAPE_CYLINDER) : sh; }
RP_DEV SmShape sm_shape_of(const DevWorld &w, int sh_in, float4 he, float border_in) {
    const int sh = sm_core_shape(sh_in);
    SmShape s; s.shape = sh; s.he = v3(he); s.axis = 1; s.border = (sh_in >= RP_SHAPE_ROUND_CUBOID && sh_in <= RP_SHAPE_ROUND_CONVEX_POLYHEDRON) ? border_in : 0.0f;
    s.tri[0] = v3(0, 0, 0); s.tri[1] = s.tri[0]; s.tri[2] = s.tri[0];
    s.radius = sh == RP_SHAPE_CAPSULE ? he.y : he.x;
    if (sh == RP_SHAPE_CAPSULE) s.axis = (int)he.z;
    s.pts = nullptr; s.fn = nullptr; s.fl = nullptr; s.loop = nullptr; s.npts = 0; s.nfaces = 0;
    if (sh == RP_SHAPE_CONVEX_POLYHEDRON) {
        const int4 h = w.cv_hdr[__float_as_int(he.w)];
        s.pts = w.cv_pts + h.x; s.npts = h.y; s.fn = w.cv_fn + h.z; s.fl = w.cv_fl + h.z; s.nfaces = h.w; s.loop = w.cv_loop; s.radius = 0.0f;
    }
    return s;
}
RP_DEV SmShape sm_point_shape() { // a ball's centre as second shape of a query
    SmShape s; s.shape = RP_SHAPE_BALL; s.he = v3(0, 0, 0); s.radius = 0.0f; s.axis = 1;
    s.pts = nullptr; s.fn = nullptr; s.fl = nullptr; s.loop = nullptr; s.npts = 0; s.nfaces = 0; s.border = 0.0f;
    s.tri[0] = v3(0, 0, 0); s.tri[1] = s.tri[0]; s.tri[2] = s.tri[0];
    return s;
}
RP_DEV float sm_border_radius(const SmShape &s) { return (s.shape == RP_SHAPE_BALL || s.shape == RP_SHAPE_CAPSULE) ? s.radius : s.border; }

// SupportMap::local_support_point of the core shape (a ball's centre, a capsule's segment)
__device__ V3 sm_support(const SmShape &s, V3 d) {
    if (s.shape == RP_SHAPE_CUBOID) return cuboid_support_point(s.he, d);
    if (s.shape == RP_SHAPE_CAPSULE) {
        V3 e = capsule_axis_dir(s.axis);
        float c = comp(d, s.axis) * s.he.x;
        return (-c > c) ? e * -s.he.x : e * s.he.x;
    }
    if (s.shape == RP_SHAPE_CYLINDER) {
        float n = sqrtf(d.x * d.x + d.z * d.z);
        V3 r = v3(0, 0, 0);
        if (n != 0.0f) r = v3(d.x / n * s.radius, 0.0f, d.z / n * s.radius);
        r.y = copysignf(s.he.y, d.y);
        return r;
    }
    if (s.shape == RP_SHAPE_CONE) {
        float n = sqrtf(d.x * d.x + d.z * d.z);
        if (n == 0.0f) return v3(0.0f, copysignf(s.he.y, d.y), 0.0f);
        V3 r = v3(d.x / n * s.radius, -s.he.y, d.z / n * s.radius);
        if (dot(d, r) < d.y * s.he.y) r = v3(0.0f, s.he.y, 0.0f);
        return r;
    }
    if (s.shape == RP_SHAPE_TRIANGLE) { // Triangle::local_support_point: the first vertex with the largest dot product
        float da = dot(s.tri[0], d), db = dot(s.tri[1], d), dc = dot(s.tri[2], d);
        if (da > db) return da > dc ? s.tri[0] : s.tri[2];
        return db > dc ? s.tri[1] : s.tri[2];
    }
    if (s.shape == RP_SHAPE_CONVEX_POLYHEDRON) { // utils::point_cloud_support_point: the first vertex with the largest dot product
        int best = 0; float bd = dot(v3(s.pts[0]), d);
        for (int i = 1; i < s.npts; ++i) { float x = dot(v3(s.pts[i]), d); if (x > bd) { bd = x; best = i; } }
        return v3(s.pts[best]);
    }
    return v3(0, 0, 0);
}

struct CsoPt { V3 p, o1, o2; };
__device__ CsoPt cso_support(const SmShape &s1, const SmShape &s2, Pose pos12, V3 dir) {
    CsoPt r;
    r.o1 = sm_support(s1, dir);
    r.o2 = pose_tp(pos12, sm_support(s2, qrot_inv(pos12.r, -dir)));
    r.p = r.o1 - r.o2;
    return r;
}

// ---- Voronoi simplex (Ericson 5.1): mask of the vertices that carry the origin's projection + barycentric coordinates ----
RP_DEV int sx_proj_seg(V3 a, V3 b, float bc[4]) {
    V3 ab = b - a;
    float t = -dot(a, ab);
    if (t <= 0.0f) { bc[0] = 1.0f; bc[1] = 0.0f; return 1; }
    float denom = dot(ab, ab);
    if (t >= denom) { bc[0] = 0.0f; bc[1] = 1.0f; return 2; }
    t = t / denom;
    bc[0] = 1.0f - t; bc[1] = t;
    return 3;
}
__device__ int sx_proj_tri(V3 a, V3 b, V3 c, float bc[4]) {
    V3 ab = b - a, ac = c - a;
    float d1 = -dot(ab, a), d2 = -dot(ac, a);
    bc[0] = bc[1] = bc[2] = 0.0f;
    if (d1 <= 0.0f && d2 <= 0.0f) { bc[0] = 1.0f; return 1; }
    float d3 = -dot(ab, b), d4 = -dot(ac, b);
    if (d3 >= 0.0f && d4 <= d3) { bc[1] = 1.0f; return 2; }
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) { float v = d1 / (d1 - d3); bc[0] = 1.0f - v; bc[1] = v; return 3; }
    float d5 = -dot(ab, c), d6 = -dot(ac, c);
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

struct GjkSimplex { CsoPt v[4]; float bc[4]; int n; CsoPt pv[4]; float pbc[4]; int pn; };
RP_DEV void sx_reset(GjkSimplex &s, CsoPt p) { s.n = 1; s.v[0] = p; s.bc[0] = 1.0f; s.pn = 1; s.pv[0] = p; s.pbc[0] = 1.0f; }
RP_DEV void sx_keep(GjkSimplex &s, int mask, const float bc[4]) {
    int k = 0;
    for (int i = 0; i < s.n; ++i)
        if (mask & (1 << i)) { s.v[k] = s.v[i]; s.bc[k] = bc[i]; ++k; }
    s.n = k;
}
// the vertices of face f of a tetrahedron and the vertex opposite: (0,1,2|3) (0,1,3|2) (0,2,3|1) (1,2,3|0)
RP_DEV int tet_face_vertex(int f, int k) { return k == 3 ? 3 - f : (f == 3 ? k + 1 : (k == 0 ? 0 : (k == 1 ? (f == 2 ? 2 : 1) : (f == 0 ? 2 : 3)))); }
__device__ V3 sx_project_and_reduce(GjkSimplex &s, int &inside) {
    float bc[4] = {0, 0, 0, 0};
    inside = 0;
    if (s.n == 1) { s.bc[0] = 1.0f; return s.v[0].p; }
    if (s.n == 2) {
        int mask = sx_proj_seg(s.v[0].p, s.v[1].p, bc);
        sx_keep(s, mask, bc);
    } else if (s.n == 3) {
        int mask = sx_proj_tri(s.v[0].p, s.v[1].p, s.v[2].p, bc);
        sx_keep(s, mask, bc);
    } else {
        float best = FLT_MAX; int best_f = -1, best_mask = 0; float best_bc[4] = {0, 0, 0, 0};
        for (int f = 0; f < 4; ++f) {
            const int i0 = tet_face_vertex(f, 0), i1 = tet_face_vertex(f, 1), i2 = tet_face_vertex(f, 2), i3 = tet_face_vertex(f, 3);
            V3 a = s.v[i0].p, b = s.v[i1].p, c = s.v[i2].p, d = s.v[i3].p;
            V3 nrm = cross(b - a, c - a);
            float sd = dot(nrm, d - a), so = -dot(nrm, a);
            if (sd * so > 0.0f) continue;
            float fb[4];
            int m = sx_proj_tri(a, b, c, fb);
            V3 q = a * fb[0] + b * fb[1] + c * fb[2];
            float d2 = len2(q);
            if (d2 < best) {
                best = d2; best_f = f; best_mask = 0;
                for (int k = 0; k < 4; ++k) best_bc[k] = 0.0f;
                if (m & 1) { best_mask |= 1 << i0; best_bc[i0] = fb[0]; }
                if (m & 2) { best_mask |= 1 << i1; best_bc[i1] = fb[1]; }
                if (m & 4) { best_mask |= 1 << i2; best_bc[i2] = fb[2]; }
            }
        }
        if (best_f < 0) { inside = 1; return v3(0, 0, 0); }
        sx_keep(s, best_mask, best_bc);
    }
    if (s.n == 3) { // inside a triangle: along the triangle's normal, exactly
        V3 nrm = cross(s.v[1].p - s.v[0].p, s.v[2].p - s.v[0].p);
        float l2 = len2(nrm);
        if (l2 > 0.0f) return nrm * (dot(nrm, s.v[0].p) / l2);
    }
    V3 q = v3(0, 0, 0);
    for (int i = 0; i < s.n; ++i) q = q + s.v[i].p * s.bc[i];
    return q;
}
__device__ int sx_add_point(GjkSimplex &s, CsoPt pt) {
    s.pn = s.n;
    for (int i = 0; i < s.n; ++i) { s.pv[i] = s.v[i]; s.pbc[i] = s.bc[i]; }
    for (int i = 0; i < s.n; ++i) {
        V3 d = s.v[i].p - pt.p;
        if (d.x == 0.0f && d.y == 0.0f && d.z == 0.0f) return 0;
    }
    if (s.n == 2) {
        V3 ab = s.v[1].p - s.v[0].p, ac = pt.p - s.v[0].p;
        if (!(len2(cross(ab, ac)) > RP_GJK_REL_TOL * RP_GJK_REL_TOL * (len2(ab) * len2(ac)))) return 0;
    } else if (s.n == 3) {
        V3 ab = s.v[1].p - s.v[0].p, ac = s.v[2].p - s.v[0].p, ap = pt.p - s.v[0].p;
        V3 nrm = cross(ab, ac);
        float h = dot(nrm, ap);
        if (!(h * h > RP_GJK_REL_TOL * RP_GJK_REL_TOL * (len2(nrm) * len2(ap)))) return 0;
    } else if (s.n != 1) return 0;
    s.v[s.n++] = pt;
    return 1;
}
RP_DEV void sx_result(const GjkSimplex &s, int prev, V3 &p1, V3 &p2) {
    V3 a = v3(0, 0, 0), b = v3(0, 0, 0);
    if (prev) { for (int i = 0; i < s.pn; ++i) { a = a + s.pv[i].o1 * s.pbc[i]; b = b + s.pv[i].o2 * s.pbc[i]; } }
    else { for (int i = 0; i < s.n; ++i) { a = a + s.v[i].o1 * s.bc[i]; b = b + s.v[i].o2 * s.bc[i]; } }
    p1 = a; p2 = b;
}

enum { RP_GJK_INTERSECTION = 0, RP_GJK_CLOSEST_POINTS = 1, RP_GJK_NO_INTERSECTION = 2 };
struct GjkResult { int kind; V3 p1, p2, dir; int unsure; };

// gjk::closest_points(pos12, g1, g2, max_dist, exact_dist = true, simplex)
__device__ GjkResult gjk_closest_points(const SmShape &s1, const SmShape &s2, Pose pos12, float max_dist, GjkSimplex &sx) {
    const float eps_tol = RP_GJK_EPS_TOL, eps_rel = 1.0918301e-3f;
    GjkResult r; r.kind = RP_GJK_INTERSECTION; r.p1 = r.p2 = v3(0, 0, 0); r.dir = v3(1, 0, 0); r.unsure = 0;
    float last_min_bound = -FLT_MAX;
    int inside;
    V3 proj = sx_project_and_reduce(sx, inside);
    float plen = len(proj);
    if (!(plen > 0.0f)) return r;
    V3 old_dir = proj * (-1.0f / plen);
    float max_bound = FLT_MAX;
    V3 dir;
    for (int niter = 0; niter < RP_GJK_MAX_ITERS; ++niter) {
        float old_max_bound = max_bound;
        plen = len(proj);
        if (!(plen > eps_tol)) return r;
        dir = proj * (-1.0f / plen); max_bound = plen;
        if (max_bound >= old_max_bound) {
            r.kind = RP_GJK_CLOSEST_POINTS; sx_result(sx, 1, r.p1, r.p2); r.dir = old_dir; r.unsure = !(last_min_bound > 0.0f); return r;
        }
        CsoPt w = cso_support(s1, s2, pos12, dir);
        float min_bound = -dot(dir, w.p);
        if (min_bound > max_dist) { r.kind = RP_GJK_NO_INTERSECTION; r.dir = dir; return r; }
        if (max_bound - min_bound <= eps_rel * max_bound) { r.kind = RP_GJK_CLOSEST_POINTS; sx_result(sx, 0, r.p1, r.p2); r.dir = dir; return r; }
        last_min_bound = min_bound;
        if (!sx_add_point(sx, w)) { r.kind = RP_GJK_CLOSEST_POINTS; sx_result(sx, 0, r.p1, r.p2); r.dir = dir; r.unsure = !(min_bound > 0.0f); return r; }
        old_dir = dir;
        proj = sx_project_and_reduce(sx, inside);
        if (inside) {
            if (min_bound >= eps_tol) { r.kind = RP_GJK_CLOSEST_POINTS; sx_result(sx, 1, r.p1, r.p2); r.dir = old_dir; return r; }
            return r;
        }
    }
    r.kind = RP_GJK_NO_INTERSECTION; r.dir = v3(1, 0, 0);
    return r;
}

// ---- expanding polytope ----
struct EpaFace { unsigned char a, b, c, alive; V3 n; float d; };
struct EpaPoly { CsoPt v[RP_EPA_MAXV]; int nv; EpaFace f[RP_EPA_MAXF]; int nf; };

__device__ int epa_face_init(const EpaPoly &P, EpaFace &f, int a, int b, int c) {
    V3 pa = P.v[a].p;
    V3 nrm = cross(P.v[b].p - pa, P.v[c].p - pa);
    float l = len(nrm);
    f.a = (unsigned char)a; f.b = (unsigned char)b; f.c = (unsigned char)c; f.alive = 1;
    if (!(l > 1.0e-18f)) { f.n = v3(0, 0, 0); f.d = FLT_MAX; return 0; }
    f.n = nrm * (1.0f / l);
    f.d = dot(f.n, pa);
    return 1;
}
__device__ int epa_add_face(EpaPoly &P, int a, int b, int c) {
    int slot = -1;
    for (int i = 0; i < P.nf; ++i) if (!P.f[i].alive) { slot = i; break; }
    if (slot < 0) { if (P.nf >= RP_EPA_MAXF) return -1; slot = P.nf++; }
    if (!epa_face_init(P, P.f[slot], a, b, c)) { P.f[slot].alive = 0; return -1; }
    return slot;
}
RP_DEV V3 epa_axis(int k) { return v3(k == 0 ? 1.0f : (k == 1 ? -1.0f : 0.0f), k == 2 ? 1.0f : (k == 3 ? -1.0f : 0.0f), k == 4 ? 1.0f : (k == 5 ? -1.0f : 0.0f)); }
__device__ int epa_blow_up(const SmShape &s1, const SmShape &s2, Pose pos12, EpaPoly &P) {
    if (P.nv == 1) {
        for (int k = 0; k < 6 && P.nv == 1; ++k) {
            CsoPt w = cso_support(s1, s2, pos12, epa_axis(k));
            if (len2(w.p - P.v[0].p) > RP_GJK_EPS_TOL) P.v[P.nv++] = w;
        }
        if (P.nv == 1) return 0;
    }
    if (P.nv == 2) {
        V3 d = P.v[1].p - P.v[0].p;
        float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
        V3 e = (ax <= ay && ax <= az) ? v3(1, 0, 0) : (ay <= az ? v3(0, 1, 0) : v3(0, 0, 1));
        V3 u = cross(d, e);
        float dl = len(d);
        V3 dn = d * (1.0f / dl);
        for (int k = 0; k < 6 && P.nv == 2; ++k) {
            CsoPt w = cso_support(s1, s2, pos12, u);
            if (len2(cross(w.p - P.v[0].p, d)) > RP_GJK_EPS_TOL * len2(d)) P.v[P.nv++] = w;
            u = u * 0.5f + cross(dn, u) * 0.86602540378f;
        }
        if (P.nv == 2) return 0;
    }
    if (P.nv == 3) {
        V3 nrm = cross(P.v[1].p - P.v[0].p, P.v[2].p - P.v[0].p);
        float l = len(nrm);
        if (!(l > 0.0f)) return 0;
        nrm = nrm * (1.0f / l);
        CsoPt w = cso_support(s1, s2, pos12, nrm);
        if (!(fabsf(dot(w.p - P.v[0].p, nrm)) > RP_GJK_EPS_TOL)) {
            w = cso_support(s1, s2, pos12, -nrm);
            if (!(fabsf(dot(w.p - P.v[0].p, nrm)) > RP_GJK_EPS_TOL)) return 0;
        }
        P.v[P.nv++] = w;
    }
    return 1;
}
RP_DEV int epa_closest_face(const EpaPoly &P) {
    int best = -1;
    for (int i = 0; i < P.nf; ++i) if (P.f[i].alive && (best < 0 || P.f[i].d < P.f[best].d)) best = i;
    return best;
}
__device__ __noinline__ int epa_closest_points(const SmShape &s1, const SmShape &s2, Pose pos12, const GjkSimplex &sx, V3 &p1, V3 &p2, V3 &normal) {
    EpaPoly P;
    P.nv = sx.n; P.nf = 0;
    for (int i = 0; i < sx.n; ++i) P.v[i] = sx.v[i];
    if (P.nv < 4 && !epa_blow_up(s1, s2, pos12, P)) return 0;
    {
        V3 a = P.v[0].p;
        float vol = dot(cross(P.v[1].p - a, P.v[2].p - a), P.v[3].p - a);
        if (vol == 0.0f) return 0;
        if (vol > 0.0f) { CsoPt t = P.v[1]; P.v[1] = P.v[2]; P.v[2] = t; }
        if (epa_add_face(P, 0, 1, 2) < 0 || epa_add_face(P, 0, 2, 3) < 0 || epa_add_face(P, 0, 3, 1) < 0 || epa_add_face(P, 1, 3, 2) < 0) return 0;
    }
    EpaFace good = P.f[0]; good.alive = 0;
    for (int iter = 0; iter < 64; ++iter) {
        int best = epa_closest_face(P);
        if (best < 0) return 0;
        if (good.alive && P.f[best].d < good.d - 1.0e-6f) break;
        good = P.f[best];
        EpaFace bf = P.f[best];
        CsoPt w = cso_support(s1, s2, pos12, bf.n);
        if (dot(w.p, bf.n) - bf.d < RP_EPA_EPS_TOL || P.nv >= RP_EPA_MAXV) break;
        unsigned char ea[RP_EPA_MAXE], eb[RP_EPA_MAXE]; int ne = 0, overflow = 0, removed = 0;
        for (int i = 0; i < P.nf; ++i) {
            EpaFace &g = P.f[i];
            if (!g.alive || !(dot(g.n, w.p - P.v[g.a].p) > 0.0f)) continue;
            g.alive = 0; ++removed;
            for (int k = 0; k < 3; ++k) {
                const unsigned char va = k == 0 ? g.a : (k == 1 ? g.b : g.c), vb = k == 0 ? g.b : (k == 1 ? g.c : g.a);
                int found = -1;
                for (int q = 0; q < ne; ++q) if (ea[q] == vb && eb[q] == va) { found = q; break; }
                if (found >= 0) { ea[found] = ea[ne - 1]; eb[found] = eb[ne - 1]; --ne; }
                else if (ne < RP_EPA_MAXE) { ea[ne] = va; eb[ne] = vb; ++ne; }
                else overflow = 1;
            }
        }
        if (removed == 0 || overflow) break;
        int k = P.nv; P.v[P.nv++] = w;
        int failed = 0;
        for (int q = 0; q < ne; ++q) if (epa_add_face(P, ea[q], eb[q], k) < 0) failed = 1;
        if (failed) break;
    }
    EpaFace bf = good;
    float bc[4];
    V3 a = P.v[bf.a].p - bf.n * bf.d, b = P.v[bf.b].p - bf.n * bf.d, c = P.v[bf.c].p - bf.n * bf.d;
    sx_proj_tri(a, b, c, bc);
    p1 = P.v[bf.a].o1 * bc[0] + P.v[bf.b].o1 * bc[1] + P.v[bf.c].o1 * bc[2];
    p2 = P.v[bf.a].o2 * bc[0] + P.v[bf.b].o2 * bc[1] + P.v[bf.c].o2 * bc[2];
    normal = bf.n;
    return 1;
}

// contact_support_map_support_map_with_params: 1 = a point pair with the unit normal from 1 to 2; 0 = further apart than `prediction`
__device__ int sm_contact(const SmShape &s1, const SmShape &s2, Pose pos12, float prediction, V3 init_dir, V3 &p1, V3 &p2, V3 &normal) {
    V3 dir = init_dir;
    float dl = len(dir);
    if (dl > FLT_EPSILON) dir = dir * (1.0f / dl);
    else {
        float tl = len(pos12.t);
        dir = tl > FLT_EPSILON ? pos12.t * (1.0f / tl) : v3(1, 0, 0);
    }
    GjkSimplex sx;
    sx_reset(sx, cso_support(s1, s2, pos12, dir));
    GjkResult r = gjk_closest_points(s1, s2, pos12, prediction, sx);
    if (r.kind == RP_GJK_CLOSEST_POINTS && r.unsure) {
        V3 q1, q2, qn;
        if (epa_closest_points(s1, s2, pos12, sx, q1, q2, qn)) {
            if (dot(q2 - q1, qn) > prediction) { normal = qn; return 0; }
            p1 = q1; p2 = q2; normal = qn; return 1;
        }
    }
    if (r.kind == RP_GJK_CLOSEST_POINTS) { p1 = r.p1; p2 = r.p2; normal = r.dir; return 1; }
    if (r.kind == RP_GJK_NO_INTERSECTION) { normal = r.dir; return 0; }
    if (epa_closest_points(s1, s2, pos12, sx, p1, p2, normal)) return 1;
    normal = v3(1, 0, 0);
    return 0;
}
// distance between the core shapes (-1 when they overlap) and the unit direction from 1 to 2
__device__ float sm_distance(const SmShape &s1, const SmShape &s2, Pose pos12, V3 &n1) {
    float tl = len(pos12.t);
    V3 dir = tl > FLT_EPSILON ? pos12.t * (1.0f / tl) : v3(1, 0, 0);
    GjkSimplex sx;
    sx_reset(sx, cso_support(s1, s2, pos12, dir));
    GjkResult r = gjk_closest_points(s1, s2, pos12, FLT_MAX, sx);
    if (r.kind != RP_GJK_CLOSEST_POINTS) { n1 = v3(0, 1, 0); return -1.0f; }
    n1 = r.dir;
    return dot(r.p2 - r.p1, r.dir);
}
RP_DEV bool sm_intersects(const SmShape &s1, const SmShape &s2, Pose pos12) {
    V3 n;
    float d = sm_distance(s1, s2, pos12, n);
    return d <= sm_border_radius(s1) + sm_border_radius(s2);
}

// ---- polygonal feature maps ----
struct PolyFeat { V3 v[4]; unsigned vid[4], eid[4], fid; int nv; };

RP_DEV void cap_dir2(V3 dir, V3 hint, float &cx, float &cz) {
    float hn = sqrtf(hint.x * hint.x + hint.z * hint.z);
    if (hn > 1.0e-6f) { cx = hint.x / hn; cz = hint.z / hn; return; }
    float dn = sqrtf(dir.x * dir.x + dir.z * dir.z);
    if (dn > FLT_EPSILON) { cx = dir.x / dn; cz = dir.z / dn; return; }
    cx = 1.0f; cz = 0.0f;
}
// PolygonalFeatureMap::local_support_feature (curved part = segment 0 with end points 1 and 11; bottom cap: vertices 1, 3, 5, 7, edges
// 2, 4, 6, 8, face 9; top cap: the same + 10)
__device__ void sm_support_feature(const SmShape &s, V3 dir, V3 hint, PolyFeat &out) {
    if (s.shape == RP_SHAPE_CUBOID) {
        Face f = cuboid_support_face(s.he, dir);
        for (int i = 0; i < 4; ++i) { out.v[i] = f.v[i]; out.vid[i] = f.vid[i]; out.eid[i] = f.eid[i]; }
        out.fid = f.fid; out.nv = 4;
        return;
    }
    if (s.shape == RP_SHAPE_CAPSULE) {
        V3 e = capsule_axis_dir(s.axis);
        out.v[0] = e * -s.he.x; out.v[1] = e * s.he.x; out.v[2] = out.v[1]; out.v[3] = out.v[1];
        out.vid[0] = 0; out.vid[1] = 2; out.vid[2] = 2; out.vid[3] = 2;
        for (int i = 0; i < 4; ++i) out.eid[i] = 1;
        out.fid = 0; out.nv = 2;
        return;
    }
    if (s.shape == RP_SHAPE_TRIANGLE) { // Triangle: PolygonalFeature::from(triangle) — the face itself whatever the direction (vertex ids 0, 2, 4, edge ids 1, 3, 5)
        for (int i = 0; i < 4; ++i) { const int k = i < 3 ? i : 2; out.v[i] = s.tri[k]; out.vid[i] = 2u * (unsigned)k; out.eid[i] = 2u * (unsigned)k + 1u; }
        out.fid = 0; out.nv = 3;
        return;
    }
    if (s.shape == RP_SHAPE_CONVEX_POLYHEDRON) { // the face whose normal is closest to dir (the first one), its first four vertices
        int best = 0; float bd = dot(v3(s.fn[0]), dir);
        for (int f = 1; f < s.nfaces; ++f) { float x = dot(v3(s.fn[f]), dir); if (x > bd) { bd = x; best = f; } }
        const int2 fl = s.fl[best];
        const int cnt = fl.y < 4 ? fl.y : 4;
        for (int i = 0; i < 4; ++i) {
            const int2 le = s.loop[fl.x + (i < cnt ? i : cnt - 1)];
            out.v[i] = v3(s.pts[le.x]); out.vid[i] = (unsigned)le.x; out.eid[i] = 0x4000u | (unsigned)le.y;
        }
        out.fid = 0x8000u | (unsigned)best; out.nv = cnt;
        return;
    }
    float r = s.radius, hh = s.he.y;
    bool curved = s.shape == RP_SHAPE_CYLINDER ? (fabsf(dir.y) < 0.5f) : (dir.y > 0.0f);
    if (curved) {
        float dn = sqrtf(dir.x * dir.x + dir.z * dir.z), cx = 1.0f, cz = 0.0f;
        if (dn > FLT_EPSILON) { cx = dir.x / dn; cz = dir.z / dn; }
        out.v[0] = v3(cx * r, -hh, cz * r);
        out.v[1] = s.shape == RP_SHAPE_CYLINDER ? v3(cx * r, hh, cz * r) : v3(0.0f, hh, 0.0f);
        out.v[2] = out.v[1]; out.v[3] = out.v[1];
        out.vid[0] = 1; out.vid[1] = 11; out.vid[2] = 11; out.vid[3] = 11;
        for (int i = 0; i < 4; ++i) out.eid[i] = 0;
        out.fid = 0; out.nv = 2;
        return;
    }
    float cx, cz;
    cap_dir2(dir, hint, cx, cz);
    float y = s.shape == RP_SHAPE_CYLINDER ? copysignf(hh, dir.y) : -hh;
    out.v[0] = v3(cx * r, y, cz * r);
    out.v[1] = v3(-cz * r, y, cx * r);
    out.v[2] = v3(-cx * r, y, -cz * r);
    out.v[3] = v3(cz * r, y, -cx * r);
    unsigned base = y < 0.0f ? 0u : 10u;
    for (int i = 0; i < 4; ++i) { out.vid[i] = base + 1u + 2u * (unsigned)i; out.eid[i] = base + 2u + 2u * (unsigned)i; }
    out.fid = base + 9u; out.nv = 4;
}

// query::details::clip_segment_segment
RP_DEV int clip_segment_segment(V3 a1, V3 b1, V3 a2, V3 b2, V3 out[4]) {
    V3 t1 = b1 - a1;
    float sq = len2(t1);
    float r20 = dot(a2 - a1, t1), r21 = dot(b2 - a1, t1);
    if (r21 < r20) { float t = r20; r20 = r21; r21 = t; V3 p = a2; a2 = b2; b2 = p; }
    if (r20 > sq || 0.0f > r21) return 0;
    V3 d2 = b2 - a2;
    if (r20 > 0.0f) { out[0] = a1 + t1 * (r20 * rp_inv(sq)); out[1] = a2; }
    else { out[0] = a1; out[1] = a2 + d2 * ((0.0f - r20) * rp_inv(r21 - r20)); }
    if (r21 < sq) { out[2] = a1 + t1 * (r21 * rp_inv(sq)); out[3] = b2; }
    else { out[2] = b1; out[3] = a2 + d2 * ((sq - r20) * rp_inv(r21 - r20)); }
    return 1;
}

#define PERP(ax, ay, bx, by) ((ax) * (by) - (ay) * (bx))
// PolygonalFeature::contacts: f2 is expressed in frame 1 already
__device__ void contacts_features(Pose pos12, const PolyFeat &f1, V3 sep, const PolyFeat &f2, LocalManifold &m) {
    V3 b0, b1; orthonormal_basis(sep, b0, b1);
    float p1x[4], p1y[4], p2x[4], p2y[4];
    for (int i = 0; i < 4; ++i) { p1x[i] = dot(f1.v[i], b0); p1y[i] = dot(f1.v[i], b1); p2x[i] = dot(f2.v[i], b0); p2y[i] = dot(f2.v[i], b1); }
    if (f1.nv == 2 && f2.nv == 2) { // contacts_edge_edge
        float t1x = p1x[1] - p1x[0], t1y = p1y[1] - p1y[0], t2x = p2x[1] - p2x[0], t2y = p2y[1] - p2y[0];
        float l1 = sqrtf(t1x * t1x + t1y * t1y), l2 = sqrtf(t2x * t2x + t2y * t2y);
        if (l1 > FLT_EPSILON && l2 > FLT_EPSILON) {
            float c = (t1x / l1) * (t2x / l2) + (t1y / l1) * (t2y / l2);
            if (!(fabsf(c) >= 0.92387953251f)) {
                float s, t;
                closest_points_segment_segment(v3(p1x[0], p1y[0], 0.0f), v3(p1x[1], p1y[1], 0.0f), v3(p2x[0], p2y[0], 0.0f), v3(p2x[1], p2y[1], 0.0f), s, t);
                V3 q1 = f1.v[0] * (1.0f - s) + f1.v[1] * s;
                V3 q2 = f2.v[0] * (1.0f - t) + f2.v[1] * t;
                lm_push(m, q1, pose_itp(pos12, q2), f1.eid[0], f2.eid[0], dot(q2 - q1, sep));
                return;
            }
        }
        V3 c4[4];
        if (clip_segment_segment(f1.v[0], f1.v[1], f2.v[0], f2.v[1], c4)) {
            lm_push(m, c4[0], pose_itp(pos12, c4[1]), f1.vid[0], f2.vid[0], dot(c4[1] - c4[0], sep));
            lm_push(m, c4[2], pose_itp(pos12, c4[3]), f1.vid[1], f2.vid[1], dot(c4[3] - c4[2], sep));
        }
        return;
    }
    if (f2.nv > 2) {
        V3 normal2_1 = cross(f2.v[2] - f2.v[1], f2.v[0] - f2.v[1]);
        float denom = dot(normal2_1, sep);
        if (!(fabsf(denom) <= FLT_EPSILON)) {
            const int last = f2.nv - 1;
            for (int i = 0; i < f1.nv; ++i) {
                float px = p1x[i], py = p1y[i];
                float sign = PERP(p2x[0] - p2x[last], p2y[0] - p2y[last], px - p2x[last], py - p2y[last]);
                bool outside = false;
                for (int j = 0; j < last; ++j) {
                    float ns = PERP(p2x[j + 1] - p2x[j], p2y[j + 1] - p2y[j], px - p2x[j], py - p2y[j]);
                    if (sign == 0.0f) sign = ns; else if (sign * ns < 0.0f) { outside = true; break; }
                }
                if (outside) continue;
                float dist = dot(f2.v[0] - f1.v[i], normal2_1) / denom;
                lm_push(m, f1.v[i], pose_itp(pos12, f1.v[i] + sep * dist), f1.vid[i], f2.fid, dist);
            }
        }
    }
    if (f1.nv > 2) {
        V3 normal1 = cross(f1.v[2] - f1.v[1], f1.v[0] - f1.v[1]);
        float denom = -dot(normal1, sep);
        if (!(fabsf(denom) <= FLT_EPSILON)) {
            const int last = f1.nv - 1;
            for (int i = 0; i < f2.nv; ++i) {
                float px = p2x[i], py = p2y[i];
                float sign = PERP(p1x[0] - p1x[last], p1y[0] - p1y[last], px - p1x[last], py - p1y[last]);
                bool outside = false;
                for (int j = 0; j < last; ++j) {
                    float ns = PERP(p1x[j + 1] - p1x[j], p1y[j + 1] - p1y[j], px - p1x[j], py - p1y[j]);
                    if (sign == 0.0f) sign = ns; else if (sign * ns < 0.0f) { outside = true; break; }
                }
                if (outside) continue;
                float dist = dot(f1.v[0] - f2.v[i], normal1) / denom;
                lm_push(m, f2.v[i] - sep * dist, pose_itp(pos12, f2.v[i]), f1.fid, f2.vid[i], dist);
            }
        }
    }
    const int ne1 = f1.nv == 2 ? 1 : f1.nv, ne2 = f2.nv == 2 ? 1 : f2.nv;
    for (int j = 0; j < ne2; ++j) {
        int j1 = (j + 1) % f2.nv;
        for (int i = 0; i < ne1; ++i) {
            int i1 = (i + 1) % f1.nv;
            float s, t;
            if (closest_points_line2d(p1x[i], p1y[i], p1x[i1], p1y[i1], p2x[j], p2y[j], p2x[j1], p2y[j1], s, t) && s > 0.0f && s < 1.0f && t > 0.0f && t < 1.0f) {
                V3 q1 = f1.v[i] * (1.0f - s) + f1.v[i1] * s;
                V3 q2 = f2.v[j] * (1.0f - t) + f2.v[j1] * t;
                lm_push(m, q1, pose_itp(pos12, q2), f1.eid[i], f2.eid[j], dot(q2 - q1, sep));
            }
        }
    }
}
#undef PERP

RP_DEV void lm_match_contacts(LocalManifold &m, int nold) {
    for (int i = 0; i < m.n; ++i)
        for (int j = 0; j < nold; ++j)
            if (m.fid[i] == m.old_fid(j)) m.src[i] = j;
}

// contact_manifold_pfm_pfm
__device__ void manifold_pfm_pfm(Pose pos12, const SmShape &s1, const SmShape &s2, float prediction, LocalManifold &m) {
    if (try_update_contacts(m, pos12)) return;
    float b1 = sm_border_radius(s1), b2 = sm_border_radius(s2);
    V3 p1, p2, n1;
    int hit = sm_contact(s1, s2, pos12, prediction + b1 + b2, m.ln1, p1, p2, n1);
    int nold = m.n;
    for (int i = 0; i < nold; ++i) m.old_fid_set(i, m.fid[i]);
    m.n = 0;
    if (!hit) { m.ln1 = n1; return; }
    V3 n2 = qrot_inv(pos12.r, -n1);
    float dist = dot(p2 - p1, n1);
    PolyFeat f1, f2;
    sm_support_feature(s1, n1, p1, f1);
    sm_support_feature(s2, n2, pose_itp(pos12, p2), f2);
    for (int i = 0; i < 4; ++i) f2.v[i] = pose_tp(pos12, f2.v[i]);
    contacts_features(pos12, f1, n1, f2, m);
    if (m.n == 0) lm_push(m, p1, pose_itp(pos12, p2), RP_FID_UNKNOWN, RP_FID_UNKNOWN, dist);
    if (b1 != 0.0f || b2 != 0.0f)
        for (int i = 0; i < m.n; ++i) {
            m.lp1[i] = m.lp1[i] + n1 * b1;
            m.lp2[i] = m.lp2[i] + n2 * b2;
            m.dist[i] = m.dist[i] - (b1 + b2);
        }
    m.ln1 = n1; m.ln2 = n2;
    lm_match_contacts(m, nold);
}

// PointQuery::project_local_point(pt, solid = false) of a cylinder / cone
__device__ V3 sm_project_point(const SmShape &s, V3 pt, bool &inside) {
    float r = s.radius, hh = s.he.y;
    float pd = sqrtf(pt.x * pt.x + pt.z * pt.z);
    float dx = 1.0f, dz = 0.0f;
    if (pd > FLT_EPSILON) { dx = pt.x / pd; dz = pt.z / pd; }
    float sx = dx * r, sz = dz * r;
    inside = false;
    if (s.shape == RP_SHAPE_CYLINDER) {
        if (pt.y >= -hh && pt.y <= hh && pd <= r) {
            inside = true;
            float to_top = hh - pt.y, to_bottom = pt.y - (-hh), to_side = r - pd;
            if (to_top < to_bottom && to_top < to_side) return v3(pt.x, hh, pt.z);
            if (to_bottom < to_top && to_bottom < to_side) return v3(pt.x, -hh, pt.z);
            return v3(sx, pt.y, sz);
        }
        if (pt.y > hh) return pd <= r ? v3(pt.x, hh, pt.z) : v3(sx, hh, sz);
        if (pt.y < -hh) return pd <= r ? v3(pt.x, -hh, pt.z) : v3(sx, -hh, sz);
        return v3(sx, pt.y, sz);
    }
    V3 on_basis = v3(pt.x, -hh, pt.z);
    if (pt.y < -hh && pd <= r) return on_basis;
    V3 apex = v3(0.0f, hh, 0.0f), rim = v3(sx, -hh, sz);
    V3 sd = rim - apex;
    V3 proj = apex + sd * rp_clamp(dot(pt - apex, sd) / dot(sd, sd), 0.0f, 1.0f);
    V3 apex_to_centre = v3(0.0f, -2.0f * hh, 0.0f);
    if (pt.y >= -hh && pt.y <= hh && dot(cross(sd, pt - apex), cross(sd, apex_to_centre)) >= 0.0f) {
        inside = true;
        if (len2(proj - pt) > len2(on_basis - pt)) return on_basis;
        return proj;
    }
    return proj;
}
// contact_manifold_convex_ball with shape1 = a cylinder / cone; flipped = the ball is collider 1
__device__ void manifold_sm_ball(Pose pos12, const SmShape &s1, float r2, float prediction, LocalManifold &m, bool flipped) {
    V3 pt = pos12.t;
    if (s1.shape == RP_SHAPE_CONVEX_POLYHEDRON || s1.shape == RP_SHAPE_TRIANGLE || s1.border > 0.0f) { // ConvexPolyhedron / Triangle / RoundShape::project_local_point = GJK / polytope pass against the point
        const SmShape centre = sm_point_shape();
        const float b1 = s1.border;
        V3 p1, p2, n1;
        if (!sm_contact(s1, centre, pos12, (r2 + prediction) + b1, v3(0, 0, 0), p1, p2, n1)) { m.n = 0; return; }
        float dist = dot(p2 - p1, n1);
        if (b1 != 0.0f) { p1 = p1 + n1 * b1; dist = dist - b1; }
        if (dist <= r2 + prediction) {
            V3 n2 = qrot_inv(pos12.r, -n1);
            V3 q2 = n2 * r2;
            int keep = m.n == 1 ? 0 : -1;
            m.n = 1;
            m.lp1[0] = flipped ? q2 : p1; m.lp2[0] = flipped ? p1 : q2; m.dist[0] = dist - r2;
            if (keep < 0) m.fid[0] = RP_FID_UNKNOWN | (RP_FID_UNKNOWN << 16);
            m.src[0] = keep;
            if (flipped) { m.ln1 = n2; m.ln2 = n1; } else { m.ln1 = n1; m.ln2 = n2; }
        } else m.n = 0;
        return;
    }
    bool inside;
    V3 proj = sm_project_point(s1, pt, inside);
    convex_ball_finish(pos12, pt, proj, inside, r2, prediction, m, flipped);
}
// contact_manifold_halfspace_pfm with shape 2 = a cylinder / cone
__device__ void manifold_halfspace_sm(Pose pos12, V3 normal1, const SmShape &s2, float prediction, LocalManifold &m, bool flipped) {
    V3 normal1_2 = qrot_inv(pos12.r, normal1);
    PolyFeat f;
    sm_support_feature(s2, -normal1_2, v3(0, 0, 0), f);
    const float border = s2.border;
    int nold = m.n;
    for (int i = 0; i < nold; ++i) m.old_fid_set(i, m.fid[i]);
    m.n = 0;
    for (int i = 0; i < f.nv; ++i) {
        V3 vtx2_1 = pose_tp(pos12, f.v[i]);
        float dist_to_plane = dot(vtx2_1, normal1);
        if (dist_to_plane - border <= prediction) {
            V3 q1 = vtx2_1 - normal1 * dist_to_plane;
            V3 q2 = border != 0.0f ? f.v[i] - normal1_2 * border : f.v[i];
            if (flipped) lm_push(m, q2, q1, f.vid[i], 0u, dist_to_plane - border);
            else lm_push(m, q1, q2, 0u, f.vid[i], dist_to_plane - border);
        }
    }
    if (flipped) { m.ln1 = -normal1_2; m.ln2 = normal1; } else { m.ln1 = normal1; m.ln2 = -normal1_2; }
    lm_match_contacts(m, nold);
}
