/*
 * oracle/ro_shapes.h — TEST INFRASTRUCTURE ONLY.
 *
 * Restatement of the pieces of parry3d 0.30.2 (NOT vendored in /root/reference; semver
 * requirement at /root/reference/Cargo.toml:76) that sit on the hot path — call site
 * /root/reference/src/geometry/narrow_phase/pair_update.rs:323-330:
 *   - Cuboid/Ball AABB and mass properties,
 *   - SAT for cuboid-cuboid (3+3 face axes one-way, 9 edge-edge axes two-way),
 *   - support faces + projected 2-D face/face clipping (PolygonalFeature::contacts),
 *   - ContactManifold::try_update_contacts (1 degree / 1e-6 thresholds), match_contacts,
 *   - ball-ball and convex(cuboid)-ball single-point manifolds,
 *   - capsules: segment-segment closest points (capsule-capsule), convex(capsule)-ball, and cuboid-capsule (SAT of the
 *     cuboid against the capsule's segment: 6 face axes one-way + 3 edge axes two-way, then the support face clipped against
 *     the segment and pushed out by the radius).  One deliberate simplification, marked below: the segment is treated as ONE
 *     edge in the edge/edge pass of the clipping (a 2-vertex PolygonalFeature walked generically would meet it twice).
 * The algorithm is restated from parry's published source as recalled; it cannot be
 * diffed against the crate here ("manifold-level parity unpinned", see rapier_oracle.h).
 */
#ifndef RO_SHAPES_H
#define RO_SHAPES_H
#include "ro_math.h"
#include <float.h>
#include <string.h>

#define RO_MAX_MANIFOLD_PTS 8   /* face/face clipping yields at most 8 points */
#define RO_FID_UNKNOWN 0xffffffffu

/* rapier ContactData — /root/reference/src/geometry/contact_pair.rs:54-74 */
typedef struct {
    float impulse;
    float warmstart_impulse;
    v3 warmstart_tangent_world;
    float warmstart_twist_impulse;
    v3 solver_dp1, solver_dp2;
} ContactData;

/* parry TrackedContact */
typedef struct {
    v3 local_p1, local_p2;
    float dist;
    uint32_t fid1, fid2;
    ContactData data;
} TrackedContact;

typedef struct {
    TrackedContact points[RO_MAX_MANIFOLD_PTS];
    int npoints;
    v3 local_n1, local_n2;
} Manifold;

typedef struct {
    v3 vertices[4];
    uint32_t vids[4], eids[4], fid;
} PolyFace;

static inline v3 cuboid_support_point(v3 he, v3 dir) {
    return V3(copysignf(he.x, dir.x), copysignf(he.y, dir.y), copysignf(he.z, dir.z));
}

/* Feature ids only need to be consistent between frames (they key match_contacts):
 * vertex id = sign bits (0..7); edge id = 8 + 8*min(va,vb) + max(va,vb); face = 100 + 2*axis + neg. */
static inline uint32_t cuboid_vid(v3 v) {
    return (uint32_t)((v.x < 0.0f) | ((v.y < 0.0f) << 1) | ((v.z < 0.0f) << 2));
}
static inline PolyFace cuboid_support_face(v3 he, v3 dir) {
    PolyFace f;
    float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
    int iamax = 0; float best = ax;
    if (ay > best) { best = ay; iamax = 1; }
    if (az > best) { best = az; iamax = 2; }
    float sign = copysignf(1.0f, vget(dir, iamax));
    if (iamax == 0) {
        f.vertices[0] = V3(he.x * sign, he.y, he.z);   f.vertices[1] = V3(he.x * sign, -he.y, he.z);
        f.vertices[2] = V3(he.x * sign, -he.y, -he.z); f.vertices[3] = V3(he.x * sign, he.y, -he.z);
    } else if (iamax == 1) {
        f.vertices[0] = V3(he.x, he.y * sign, he.z);   f.vertices[1] = V3(-he.x, he.y * sign, he.z);
        f.vertices[2] = V3(-he.x, he.y * sign, -he.z); f.vertices[3] = V3(he.x, he.y * sign, -he.z);
    } else {
        f.vertices[0] = V3(he.x, he.y, he.z * sign);   f.vertices[1] = V3(he.x, -he.y, he.z * sign);
        f.vertices[2] = V3(-he.x, -he.y, he.z * sign); f.vertices[3] = V3(-he.x, he.y, he.z * sign);
    }
    for (int i = 0; i < 4; ++i) f.vids[i] = cuboid_vid(f.vertices[i]);
    for (int i = 0; i < 4; ++i) {
        uint32_t a = f.vids[i], b = f.vids[(i + 1) & 3];
        uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
        f.eids[i] = 8u + 8u * lo + hi;
    }
    f.fid = 100u + 2u * (uint32_t)iamax + (sign < 0.0f ? 1u : 0u);
    return f;
}

/* sat::cuboid_cuboid_find_local_separating_normal_oneway */
static inline float sat_normal_oneway(v3 he1, v3 he2, pose pos12, v3 *out_dir) {
    float best = -FLT_MAX; v3 best_dir = V3(0, 0, 0);
    for (int i = 0; i < 3; ++i) {
        float sign = copysignf(1.0f, vget(pos12.t, i));
        v3 axis1 = V3(0, 0, 0); vset(&axis1, i, sign);
        v3 axis2 = qrot_inv(pos12.r, vneg(axis1));
        v3 local_pt2 = cuboid_support_point(he2, axis2);
        v3 pt2 = pose_tp(pos12, local_pt2);
        float sep = vget(pt2, i) * sign - vget(he1, i);
        if (sep > best) { best = sep; best_dir = axis1; }
    }
    *out_dir = best_dir;
    return best;
}
/* sat::cuboid_cuboid_compute_separation_wrt_local_line */
static inline float sat_sep_wrt_line(v3 he1, v3 he2, pose pos12, v3 axis1, v3 *out_axis) {
    float signum = copysignf(1.0f, vdot(pos12.t, axis1));
    axis1 = vmul(axis1, signum);
    v3 axis2 = qrot_inv(pos12.r, vneg(axis1));
    v3 local_pt1 = cuboid_support_point(he1, axis1);
    v3 local_pt2 = cuboid_support_point(he2, axis2);
    v3 pt2 = pose_tp(pos12, local_pt2);
    *out_axis = axis1;
    return vdot(vsub(pt2, local_pt1), axis1);
}
/* sat::cuboid_cuboid_find_local_separating_edge_twoway */
static inline float sat_edge_twoway(v3 he1, v3 he2, pose pos12, v3 *out_dir) {
    float best = -FLT_MAX; v3 best_dir = V3(0, 0, 0);
    v3 x2 = qrot(pos12.r, V3(1, 0, 0)), y2 = qrot(pos12.r, V3(0, 1, 0)), z2 = qrot(pos12.r, V3(0, 0, 1));
    v3 axes[9] = {
        V3(0, -x2.z, x2.y), V3(x2.z, 0, -x2.x), V3(-x2.y, x2.x, 0),
        V3(0, -y2.z, y2.y), V3(y2.z, 0, -y2.x), V3(-y2.y, y2.x, 0),
        V3(0, -z2.z, z2.y), V3(z2.z, 0, -z2.x), V3(-z2.y, z2.x, 0),
    };
    for (int k = 0; k < 9; ++k) {
        float n = vlen(axes[k]);
        if (n > FLT_EPSILON) {
            v3 ax;
            float sep = sat_sep_wrt_line(he1, he2, pos12, vmul(axes[k], 1.0f / n), &ax);
            if (sep > best) { best = sep; best_dir = ax; }
        }
    }
    *out_dir = best_dir;
    return best;
}

static inline int ro_relative_eq0(float x) {
    /* approx::relative_eq!(x, 0.0): |x| <= max(eps_abs, eps_rel*max(|x|,0)) with both = FLT_EPSILON */
    return fabsf(x) <= FLT_EPSILON;
}
static inline int ro_ulps_eq(float a, float b) {
    if (fabsf(a - b) <= FLT_EPSILON) return 1;
    if ((a < 0) != (b < 0)) return 0;
    int32_t ia, ib; memcpy(&ia, &a, 4); memcpy(&ib, &b, 4);
    int32_t d = ia > ib ? ia - ib : ib - ia;
    return d <= 4;
}
/* polygonal_feature3d.rs closest_points_line2d */
static inline int closest_points_line2d(const float e1[2][2], const float e2[2][2], float *s_out, float *t_out) {
    float d1x = e1[1][0] - e1[0][0], d1y = e1[1][1] - e1[0][1];
    float d2x = e2[1][0] - e2[0][0], d2y = e2[1][1] - e2[0][1];
    float rx = e1[0][0] - e2[0][0], ry = e1[0][1] - e2[0][1];
    float a = d1x * d1x + d1y * d1y;
    float e = d2x * d2x + d2y * d2y;
    float f = d2x * rx + d2y * ry;
    const float eps = FLT_EPSILON;
    if (a <= eps && e <= eps) { *s_out = 0; *t_out = 0; return 1; }
    if (a <= eps) { *s_out = 0; *t_out = f / e; return 1; }
    float c = d1x * rx + d1y * ry;
    if (e <= eps) { *s_out = -c / a; *t_out = 0; return 1; }
    float b = d1x * d2x + d1y * d2y;
    float ae = a * e, bb = b * b;
    float denom = ae - bb;
    int parallel = denom <= eps || ro_ulps_eq(ae, bb);
    if (parallel) return 0;
    float s = (b * f - c * e) / denom;
    *s_out = s; *t_out = (b * s + f) / e;
    return 1;
}

static inline void manifold_push(Manifold *m, v3 p1, v3 p2, uint32_t f1, uint32_t f2, float dist) {
    if (m->npoints >= RO_MAX_MANIFOLD_PTS) return;
    TrackedContact *c = &m->points[m->npoints++];
    memset(c, 0, sizeof(*c));
    c->local_p1 = p1; c->local_p2 = p2; c->fid1 = f1; c->fid2 = f2; c->dist = dist;
}

/* PolygonalFeature::contacts_face_face; face2 already expressed in shape-1 space. */
static inline void contacts_face_face(pose pos12, const PolyFace *face1, v3 sep_axis1, const PolyFace *face2, Manifold *m) {
    v3 basis[2]; orthonormal_basis(sep_axis1, basis);
    float pf1[4][2], pf2[4][2];
    for (int i = 0; i < 4; ++i) {
        pf1[i][0] = vdot(face1->vertices[i], basis[0]); pf1[i][1] = vdot(face1->vertices[i], basis[1]);
        pf2[i][0] = vdot(face2->vertices[i], basis[0]); pf2[i][1] = vdot(face2->vertices[i], basis[1]);
    }
#define PERP(ax, ay, bx, by) ((ax) * (by) - (ay) * (bx))
    {   /* vertices of face1 inside face2 */
        v3 normal2_1 = vcross(vsub(face2->vertices[2], face2->vertices[1]), vsub(face2->vertices[0], face2->vertices[1]));
        float denom = vdot(normal2_1, sep_axis1);
        if (!ro_relative_eq0(denom)) {
            for (int i = 0; i < 4; ++i) {
                float px = pf1[i][0], py = pf1[i][1];
                float sign = PERP(pf2[0][0] - pf2[3][0], pf2[0][1] - pf2[3][1], px - pf2[3][0], py - pf2[3][1]);
                int outside = 0;
                for (int j = 0; j < 3; ++j) {
                    float ns = PERP(pf2[j + 1][0] - pf2[j][0], pf2[j + 1][1] - pf2[j][1], px - pf2[j][0], py - pf2[j][1]);
                    if (sign == 0.0f) sign = ns;
                    else if (sign * ns < 0.0f) { outside = 1; break; }
                }
                if (outside) continue;
                float dist = vdot(vsub(face2->vertices[0], face1->vertices[i]), normal2_1) / denom;
                v3 local_p1 = face1->vertices[i];
                v3 local_p2_1 = vadd(face1->vertices[i], vmul(sep_axis1, dist));
                manifold_push(m, local_p1, pose_itp(pos12, local_p2_1), face1->vids[i], face2->fid, dist);
            }
        }
    }
    {   /* vertices of face2 inside face1 */
        v3 normal1 = vcross(vsub(face1->vertices[2], face1->vertices[1]), vsub(face1->vertices[0], face1->vertices[1]));
        float denom = -vdot(normal1, sep_axis1);
        if (!ro_relative_eq0(denom)) {
            for (int i = 0; i < 4; ++i) {
                float px = pf2[i][0], py = pf2[i][1];
                float sign = PERP(pf1[0][0] - pf1[3][0], pf1[0][1] - pf1[3][1], px - pf1[3][0], py - pf1[3][1]);
                int outside = 0;
                for (int j = 0; j < 3; ++j) {
                    float ns = PERP(pf1[j + 1][0] - pf1[j][0], pf1[j + 1][1] - pf1[j][1], px - pf1[j][0], py - pf1[j][1]);
                    if (sign == 0.0f) sign = ns;
                    else if (sign * ns < 0.0f) { outside = 1; break; }
                }
                if (outside) continue;
                float dist = vdot(vsub(face1->vertices[0], face2->vertices[i]), normal1) / denom;
                v3 local_p2_1 = face2->vertices[i];
                v3 local_p1 = vsub(face2->vertices[i], vmul(sep_axis1, dist));
                manifold_push(m, local_p1, pose_itp(pos12, local_p2_1), face1->fid, face2->vids[i], dist);
            }
        }
    }
#undef PERP
    /* edge/edge intersections */
    for (int j = 0; j < 4; ++j) {
        float e2[2][2] = {{pf2[j][0], pf2[j][1]}, {pf2[(j + 1) & 3][0], pf2[(j + 1) & 3][1]}};
        for (int i = 0; i < 4; ++i) {
            float e1[2][2] = {{pf1[i][0], pf1[i][1]}, {pf1[(i + 1) & 3][0], pf1[(i + 1) & 3][1]}};
            float s, t;
            if (closest_points_line2d(e1, e2, &s, &t)) {
                if (s > 0.0f && s < 1.0f && t > 0.0f && t < 1.0f) {
                    v3 a0 = face1->vertices[i], a1 = face1->vertices[(i + 1) & 3];
                    v3 b0 = face2->vertices[j], b1 = face2->vertices[(j + 1) & 3];
                    v3 local_p1 = vadd(vmul(a0, 1.0f - s), vmul(a1, s));
                    v3 local_p2_1 = vadd(vmul(b0, 1.0f - t), vmul(b1, t));
                    float dist = vdot(vsub(local_p2_1, local_p1), sep_axis1);
                    manifold_push(m, local_p1, pose_itp(pos12, local_p2_1), face1->eids[i], face2->eids[j], dist);
                }
            }
        }
    }
}

/* ContactManifold::try_update_contacts (COS_1_DEGREES, dist^2 threshold 1e-6) */
static inline int manifold_try_update_contacts(Manifold *m, pose pos12) {
    const float DOT_THRESHOLD = 0.99984769515f; /* cos(1 deg) */
    const float DIST_SQ_THRESHOLD = 1.0e-6f;
    if (m->npoints == 0) return 0;
    v3 local_n2 = qrot(pos12.r, m->local_n2);
    if (-vdot(m->local_n1, local_n2) < DOT_THRESHOLD) return 0;
    /* parry mutates points as it goes and bails out mid-way; the full recompute that
     * follows overwrites them, so working on a copy is equivalent. */
    TrackedContact tmp[RO_MAX_MANIFOLD_PTS];
    memcpy(tmp, m->points, sizeof(tmp));
    for (int i = 0; i < m->npoints; ++i) {
        TrackedContact *pt = &tmp[i];
        v3 local_p2 = pose_tp(pos12, pt->local_p2);
        v3 dpt = vsub(local_p2, pt->local_p1);
        float dist = vdot(dpt, m->local_n1);
        if (dist * pt->dist < 0.0f) return 0;
        v3 new_local_p1 = vsub(local_p2, vmul(m->local_n1, dist));
        if (vlen2(vsub(pt->local_p1, new_local_p1)) > DIST_SQ_THRESHOLD) return 0;
        pt->dist = dist;
        pt->local_p1 = new_local_p1;
    }
    memcpy(m->points, tmp, sizeof(tmp));
    return 1;
}

/* contact_manifold_cuboid_cuboid */
static inline void manifold_cuboid_cuboid(pose pos12, v3 he1, v3 he2, float prediction, Manifold *m) {
    if (manifold_try_update_contacts(m, pos12)) return;
    pose pos21 = pose_inv(pos12);
    v3 d1, d2, d3;
    float s1 = sat_normal_oneway(he1, he2, pos12, &d1);
    if (s1 > prediction) { m->npoints = 0; return; }
    float s2 = sat_normal_oneway(he2, he1, pos21, &d2);
    if (s2 > prediction) { m->npoints = 0; return; }
    float s3 = sat_edge_twoway(he1, he2, pos12, &d3);
    if (s3 > prediction) { m->npoints = 0; return; }
    v3 best_dir = d1;
    if (s2 > s1 && s2 > s3) best_dir = qrot(pos12.r, vneg(d2));
    else if (s3 > s1) best_dir = d3;

    v3 local_n2 = qrot(pos21.r, vneg(best_dir));
    PolyFace f1 = cuboid_support_face(he1, best_dir);
    PolyFace f2 = cuboid_support_face(he2, local_n2);
    for (int i = 0; i < 4; ++i) f2.vertices[i] = pose_tp(pos12, f2.vertices[i]);

    TrackedContact old[RO_MAX_MANIFOLD_PTS]; int nold = m->npoints;
    memcpy(old, m->points, sizeof(old));
    m->npoints = 0;
    contacts_face_face(pos12, &f1, best_dir, &f2, m);
    m->local_n1 = best_dir;
    m->local_n2 = local_n2;
    /* match_contacts: transfer tracked data by feature-id pair */
    for (int i = 0; i < m->npoints; ++i)
        for (int j = 0; j < nold; ++j)
            if (m->points[i].fid1 == old[j].fid1 && m->points[i].fid2 == old[j].fid2)
                m->points[i].data = old[j].data;
}

/* contact_manifold_ball_ball */
static inline void manifold_ball_ball(pose pos12, float r1, float r2, float prediction, Manifold *m) {
    v3 c = pos12.t;
    float len = vlen(c);
    float dist = len - r1 - r2;
    if (dist < prediction) {
        v3 n1 = len > 0.0f ? vmul(c, 1.0f / len) : V3(0, 1, 0);
        v3 n2 = qrot_inv(pos12.r, vneg(n1));
        v3 p1 = vmul(n1, r1), p2 = vmul(n2, r2);
        if (m->npoints != 0) {
            m->points[0].local_p1 = p1; m->points[0].local_p2 = p2; m->points[0].dist = dist;
            m->points[0].fid1 = 0; m->points[0].fid2 = 0; m->npoints = 1;
        } else {
            manifold_push(m, p1, p2, 0, 0, dist);
        }
        m->local_n1 = n1; m->local_n2 = n2;
    } else {
        m->npoints = 0;
    }
}

/* contact_manifold_convex_ball with shape1 = cuboid; `flipped` = the ball is collider 1.  The projection is the NON-solid one
 * (project_local_point(.., false)): a centre inside the cuboid projects onto the nearest face and the normal / distance are negated
 * (`proj.is_inside`), so a deeply penetrating ball is pushed back out through that face. */
static inline void manifold_cuboid_ball(pose pos12, v3 he1, float r2, float prediction, Manifold *m, int flipped) {
    v3 pt = pos12.t;
    /* Aabb::project_local_point */
    v3 mins_pt = vsub(vneg(he1), pt), pt_maxs = vsub(pt, he1);
    v3 shift = V3(ro_maxf(mins_pt.x, 0) - ro_maxf(pt_maxs.x, 0), ro_maxf(mins_pt.y, 0) - ro_maxf(pt_maxs.y, 0),
                  ro_maxf(mins_pt.z, 0) - ro_maxf(pt_maxs.z, 0));
    int inside = (shift.x == 0.0f && shift.y == 0.0f && shift.z == 0.0f);
    if (inside) { /* nearest face: the largest (closest to zero) of the six negative slacks */
        float best = -FLT_MAX; int best_id = 0, is_mins = 0;
        for (int i = 0; i < 3; ++i) {
            float mp = vget(mins_pt, i), pm = vget(pt_maxs, i);
            if (mp < pm) { if (pm > best) { best_id = i; is_mins = 0; best = pm; } }
            else if (mp > best) { best_id = i; is_mins = 1; best = mp; }
        }
        shift = V3(0, 0, 0); vset(&shift, best_id, is_mins ? best : -best);
    }
    v3 proj = vadd(pt, shift);
    v3 dpos = vsub(pt, proj);
    float dist = vlen(dpos);
    if (!(dist > 0.0f)) return; /* Unit::try_new_and_get(dpos, 0.0) fails: manifold left untouched */
    v3 n1 = vmul(dpos, 1.0f / dist);
    if (inside) { n1 = vneg(n1); dist = -dist; }
    if (dist <= r2 + prediction) {
        v3 n2 = qrot_inv(pos12.r, vneg(n1));
        v3 p2 = vmul(n2, r2);
        float d = dist - r2;
        v3 a = flipped ? p2 : proj, b = flipped ? proj : p2;
        if (m->npoints != 1) {
            m->npoints = 0;
            manifold_push(m, a, b, RO_FID_UNKNOWN, RO_FID_UNKNOWN, d);
        } else {
            m->points[0].local_p1 = a; m->points[0].local_p2 = b; m->points[0].dist = d;
        }
        if (flipped) { m->local_n1 = n2; m->local_n2 = n1; }
        else { m->local_n1 = n1; m->local_n2 = n2; }
    } else {
        m->npoints = 0;
    }
}
/* ---- capsules (parry shape::Capsule = segment [a, b] + radius; ColliderBuilder::capsule_x/y/z: a = -hh e_axis, b = +hh e_axis) ---- */
static inline v3 capsule_axis_dir(int axis) { return axis == 0 ? V3(1, 0, 0) : axis == 2 ? V3(0, 0, 1) : V3(0, 1, 0); }
/* Segment::project_local_point */
static inline v3 segment_project_point(v3 a, v3 b, v3 pt) {
    v3 ab = vsub(b, a), ap = vsub(pt, a);
    float ab_ap = vdot(ab, ap), sqnab = vdot(ab, ab);
    if (ab_ap <= 0.0f) return a;
    if (ab_ap >= sqnab) return b;
    float u = ab_ap / sqnab;
    return vadd(a, vmul(ab, u));
}
/* query::details::closest_points_segment_segment_with_locations_nD (Ericson, Real-Time Collision Detection 5.1.9) */
static inline void closest_points_segment_segment(v3 a1, v3 b1, v3 a2, v3 b2, float *s_out, float *t_out) {
    v3 d1 = vsub(b1, a1), d2 = vsub(b2, a2), r = vsub(a1, a2);
    float a = vdot(d1, d1), e = vdot(d2, d2), f = vdot(d2, r);
    const float eps = FLT_EPSILON;
    float s, t;
    if (a <= eps && e <= eps) { s = 0.0f; t = 0.0f; }
    else if (a <= eps) { s = 0.0f; t = ro_clampf(f / e, 0.0f, 1.0f); }
    else {
        float c = vdot(d1, r);
        if (e <= eps) { t = 0.0f; s = ro_clampf(-c / a, 0.0f, 1.0f); }
        else {
            float b = vdot(d1, d2);
            float ae = a * e, bb = b * b, denom = ae - bb;
            if (denom > eps && !ro_ulps_eq(ae, bb)) s = ro_clampf((b * f - c * e) / denom, 0.0f, 1.0f); else s = 0.0f;
            t = (b * s + f) / e;
            if (t < 0.0f) { t = 0.0f; s = ro_clampf(-c / a, 0.0f, 1.0f); }
            else if (t > 1.0f) { t = 1.0f; s = ro_clampf((b - c) / a, 0.0f, 1.0f); }
        }
    }
    *s_out = s; *t_out = t;
}
/* contact_manifold_capsule_capsule (3-D): one contact between the closest points of the two segments */
static inline void manifold_capsule_capsule(pose pos12, float hh1, float r1, int axis1, float hh2, float r2, int axis2, float prediction, Manifold *m) {
    v3 e1 = capsule_axis_dir(axis1), e2 = capsule_axis_dir(axis2);
    v3 a1 = vmul(e1, -hh1), b1 = vmul(e1, hh1);
    v3 a2 = pose_tp(pos12, vmul(e2, -hh2)), b2 = pose_tp(pos12, vmul(e2, hh2));
    float s, t; closest_points_segment_segment(a1, b1, a2, b2, &s, &t);
    v3 p1 = vadd(vmul(a1, 1.0f - s), vmul(b1, s)), p2_1 = vadd(vmul(a2, 1.0f - t), vmul(b2, t));
    v3 d = vsub(p2_1, p1);
    float l = vlen(d);
    v3 n1 = l > FLT_EPSILON ? vmul(d, 1.0f / l) : V3(0, 1, 0);
    float dist = vdot(d, n1) - r1 - r2;
    if (dist <= prediction) {
        v3 n2 = qrot_inv(pos12.r, vneg(n1));
        v3 lp1 = vadd(p1, vmul(n1, r1)), lp2 = vadd(pose_itp(pos12, p2_1), vmul(n2, r2));
        if (m->npoints != 0) {
            m->points[0].local_p1 = lp1; m->points[0].local_p2 = lp2; m->points[0].dist = dist;
            m->points[0].fid1 = 0; m->points[0].fid2 = 0; m->npoints = 1;
        } else manifold_push(m, lp1, lp2, 0, 0, dist);
        m->local_n1 = n1; m->local_n2 = n2;
    } else m->npoints = 0;
}
/* contact_manifold_convex_ball with shape1 = capsule (Capsule::project_local_point, non-solid: a centre inside the capsule projects
 * onto its surface along the direction from the segment, normal and distance negated); `flipped` = the ball is collider 1 */
static inline void manifold_capsule_ball(pose pos12, float hh1, float r1, int axis1, float r2, float prediction, Manifold *m, int flipped) {
    v3 e1 = capsule_axis_dir(axis1);
    v3 pt = pos12.t;
    v3 sp = segment_project_point(vmul(e1, -hh1), vmul(e1, hh1), pt);
    v3 dproj = vsub(pt, sp);
    float dseg = vlen(dproj);
    int inside; v3 proj;
    if (dseg > FLT_EPSILON) { inside = dseg <= r1; proj = vadd(sp, vmul(vmul(dproj, 1.0f / dseg), r1)); }
    else { inside = 1; proj = vadd(sp, V3(0, r1, 0)); } /* centre on the segment: pushed along +Y */
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
/* sat::cuboid_support_map_find_local_separating_normal_oneway with shape2 = segment [a2, b2] (already in the cuboid's frame) */
static inline float sat_cuboid_segment_normal_oneway(v3 he1, v3 a2, v3 b2, v3 *out_dir) {
    float best = -FLT_MAX; v3 best_dir = V3(0, 0, 0);
    for (int i = 0; i < 3; ++i)
        for (int sg = 0; sg < 2; ++sg) {
            float sign = sg == 0 ? -1.0f : 1.0f;
            v3 axis1 = V3(0, 0, 0); vset(&axis1, i, sign);
            /* support point of the segment toward -axis1 */
            v3 dir = vneg(axis1);
            v3 pt2 = vdot(a2, dir) > vdot(b2, dir) ? a2 : b2;
            float sep = vget(pt2, i) * sign - vget(he1, i);
            if (sep > best) { best = sep; best_dir = axis1; }
        }
    *out_dir = best_dir;
    return best;
}
/* sat::cuboid_support_map_compute_separation_wrt_local_line (two-way) + cuboid_segment_find_local_separating_edge_twoway */
static inline float sat_cuboid_segment_edge_twoway(v3 he1, v3 a2, v3 b2, v3 *out_dir) {
    float best = -FLT_MAX; v3 best_dir = V3(0, 0, 0);
    v3 x2 = vsub(b2, a2);
    v3 axes[3] = {V3(0, -x2.z, x2.y), V3(x2.z, 0, -x2.x), V3(-x2.y, x2.x, 0)};
    for (int k = 0; k < 3; ++k) {
        float n = vlen(axes[k]);
        if (!(n > FLT_EPSILON)) continue;
        v3 axis1 = vmul(axes[k], 1.0f / n);
        v3 lp1 = cuboid_support_point(he1, axis1);
        v3 q = vdot(a2, vneg(axis1)) > vdot(b2, vneg(axis1)) ? a2 : b2;
        float sep1 = vdot(vsub(q, lp1), axis1);
        v3 naxis = vneg(axis1);
        v3 lp1b = cuboid_support_point(he1, naxis);
        v3 qb = vdot(a2, axis1) > vdot(b2, axis1) ? a2 : b2;
        float sep2 = vdot(vsub(qb, lp1b), naxis);
        float sep = sep1 > sep2 ? sep1 : sep2;
        v3 ax = sep1 > sep2 ? axis1 : naxis;
        if (sep > best) { best = sep; best_dir = ax; }
    }
    *out_dir = best_dir;
    return best;
}
/* contact_manifold_cuboid_capsule: `pos12` = pose of the capsule in the cuboid's frame, `upd` = pose of collider 2 in collider
 * 1's frame (== pos12 unless flipped), `flipped` = the capsule is collider 1 (points, feature ids and normals swap on output) */
static inline void manifold_cuboid_capsule(pose pos12, pose upd, v3 he1, float hh2, float r2, int axis2, float prediction, Manifold *m, int flipped) {
    if (manifold_try_update_contacts(m, upd)) return;
    pose pos21 = pose_inv(pos12);
    v3 e2 = capsule_axis_dir(axis2);
    v3 a2 = pose_tp(pos12, vmul(e2, -hh2)), b2 = pose_tp(pos12, vmul(e2, hh2));
    v3 d1, d3;
    float s1 = sat_cuboid_segment_normal_oneway(he1, a2, b2, &d1);
    if (s1 > r2 + prediction) { m->npoints = 0; return; }
    float s3 = sat_cuboid_segment_edge_twoway(he1, a2, b2, &d3);
    if (s3 > r2 + prediction) { m->npoints = 0; return; }
    v3 best_dir = s3 > s1 ? d3 : d1;
    v3 n2 = qrot(pos21.r, vneg(best_dir));
    PolyFace f1 = cuboid_support_face(he1, best_dir);
    v3 sv[2] = {a2, b2};
    const uint32_t seg_vid[2] = {0u, 2u}, seg_eid = 1u;

    Manifold t; t.npoints = 0;
    v3 basis[2]; orthonormal_basis(best_dir, basis);
    float pf1[4][2], ps[2][2];
    for (int i = 0; i < 4; ++i) { pf1[i][0] = vdot(f1.vertices[i], basis[0]); pf1[i][1] = vdot(f1.vertices[i], basis[1]); }
    for (int i = 0; i < 2; ++i) { ps[i][0] = vdot(sv[i], basis[0]); ps[i][1] = vdot(sv[i], basis[1]); }
#define PERP(ax, ay, bx, by) ((ax) * (by) - (ay) * (bx))
    {   /* segment vertices inside the face */
        v3 normal1 = vcross(vsub(f1.vertices[2], f1.vertices[1]), vsub(f1.vertices[0], f1.vertices[1]));
        float denom = -vdot(normal1, best_dir);
        if (!ro_relative_eq0(denom)) {
            for (int i = 0; i < 2; ++i) {
                float px = ps[i][0], py = ps[i][1];
                float sign = PERP(pf1[0][0] - pf1[3][0], pf1[0][1] - pf1[3][1], px - pf1[3][0], py - pf1[3][1]);
                int outside = 0;
                for (int j = 0; j < 3; ++j) {
                    float ns = PERP(pf1[j + 1][0] - pf1[j][0], pf1[j + 1][1] - pf1[j][1], px - pf1[j][0], py - pf1[j][1]);
                    if (sign == 0.0f) sign = ns;
                    else if (sign * ns < 0.0f) { outside = 1; break; }
                }
                if (outside) continue;
                float dist = vdot(vsub(f1.vertices[0], sv[i]), normal1) / denom;
                v3 local_p1 = vsub(sv[i], vmul(best_dir, dist));
                manifold_push(&t, local_p1, pose_itp(pos12, sv[i]), f1.fid, seg_vid[i], dist);
            }
        }
    }
#undef PERP
    {   /* the segment against the face's edges (ONE pass over the segment, see the header) */
        float e2p[2][2] = {{ps[0][0], ps[0][1]}, {ps[1][0], ps[1][1]}};
        for (int i = 0; i < 4; ++i) {
            float e1p[2][2] = {{pf1[i][0], pf1[i][1]}, {pf1[(i + 1) & 3][0], pf1[(i + 1) & 3][1]}};
            float s, tt;
            if (closest_points_line2d(e1p, e2p, &s, &tt) && s > 0.0f && s < 1.0f && tt > 0.0f && tt < 1.0f) {
                v3 local_p1 = vadd(vmul(f1.vertices[i], 1.0f - s), vmul(f1.vertices[(i + 1) & 3], s));
                v3 local_p2_1 = vadd(vmul(sv[0], 1.0f - tt), vmul(sv[1], tt));
                float dist = vdot(vsub(local_p2_1, local_p1), best_dir);
                manifold_push(&t, local_p1, pose_itp(pos12, local_p2_1), f1.eids[i], seg_eid, dist);
            }
        }
    }
    TrackedContact old[RO_MAX_MANIFOLD_PTS]; int nold = m->npoints;
    memcpy(old, m->points, sizeof(old));
    m->npoints = 0;
    for (int i = 0; i < t.npoints; ++i) {
        TrackedContact c = t.points[i];
        c.local_p2 = vadd(c.local_p2, vmul(n2, r2)); /* push the segment point out to the capsule's surface */
        c.dist = c.dist - r2;
        if (flipped) { v3 tp = c.local_p1; c.local_p1 = c.local_p2; c.local_p2 = tp; uint32_t tf = c.fid1; c.fid1 = c.fid2; c.fid2 = tf; }
        m->points[m->npoints++] = c;
    }
    if (flipped) { m->local_n1 = n2; m->local_n2 = best_dir; } else { m->local_n1 = best_dir; m->local_n2 = n2; }
    for (int i = 0; i < m->npoints; ++i)
        for (int j = 0; j < nold; ++j)
            if (m->points[i].fid1 == old[j].fid1 && m->points[i].fid2 == old[j].fid2)
                m->points[i].data = old[j].data;
}
/* ---- half-spaces (parry shape::HalfSpace{normal}: the solid region dot(normal, p) <= 0 of the collider's frame;
 * ColliderBuilder::halfspace(outward_normal), collider.rs) ---- */
/* contact_manifold_convex_ball with shape1 = half-space: HalfSpace::project_local_point(pt, false) = pt - normal * dot(normal, pt),
 * is_inside = dot <= 0; the contact normal is always the plane's, the distance is signed.  `flipped` = the ball is collider 1 */
static inline void manifold_halfspace_ball(pose pos12, v3 normal1, float r2, float prediction, Manifold *m, int flipped) {
    v3 pt = pos12.t;
    float dd = vdot(normal1, pt);
    v3 proj = vadd(pt, vmul(vneg(normal1), dd));
    v3 dpos = vsub(pt, proj);
    float dist = vlen(dpos);
    if (!(dist > 0.0f)) return;
    v3 n1 = vmul(dpos, 1.0f / dist);
    if (dd <= 0.0f) { n1 = vneg(n1); dist = -dist; }
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
/* contact_manifold_halfspace_pfm: shape2 = a cuboid (support face toward the plane, border radius 0) or a capsule (its segment, both
 * end points, border radius = the capsule's); every feature vertex within `prediction` of the plane is a contact.  No
 * try_update_contacts in this generator: the points are recomputed on every full update and matched by feature id.
 * `pos12` = pose of shape 2 in the half-space's frame; `flipped` = the half-space is collider 2 */
static inline void manifold_halfspace_pfm(pose pos12, v3 normal1, int shape2_is_capsule, v3 he2, float hh2, float r2, int axis2, float prediction, Manifold *m, int flipped) {
    v3 normal1_2 = qrot_inv(pos12.r, normal1);
    v3 vtx[4]; uint32_t vid[4]; int nv; float border = 0.0f;
    if (shape2_is_capsule) {
        v3 e2 = capsule_axis_dir(axis2);
        vtx[0] = vmul(e2, -hh2); vtx[1] = vmul(e2, hh2); vid[0] = 0u; vid[1] = 2u; nv = 2; border = r2;
    } else {
        PolyFace f = cuboid_support_face(he2, vneg(normal1_2));
        for (int i = 0; i < 4; ++i) { vtx[i] = f.vertices[i]; vid[i] = f.vids[i]; }
        nv = 4;
    }
    TrackedContact old[RO_MAX_MANIFOLD_PTS]; int nold = m->npoints;
    memcpy(old, m->points, sizeof(old));
    m->npoints = 0;
    for (int i = 0; i < nv; ++i) {
        v3 vtx2_1 = pose_tp(pos12, vtx[i]);
        float dist_to_plane = vdot(vtx2_1, normal1);
        if (dist_to_plane - border <= prediction) {
            v3 p1 = vsub(vtx2_1, vmul(normal1, dist_to_plane));
            v3 p2 = vsub(vtx[i], vmul(normal1_2, border));
            if (flipped) manifold_push(m, p2, p1, vid[i], 0u, dist_to_plane - border);
            else manifold_push(m, p1, p2, 0u, vid[i], dist_to_plane - border);
        }
    }
    if (flipped) { m->local_n1 = vneg(normal1_2); m->local_n2 = normal1; }
    else { m->local_n1 = normal1; m->local_n2 = vneg(normal1_2); }
    for (int i = 0; i < m->npoints; ++i)
        for (int j = 0; j < nold; ++j)
            if (m->points[i].fid1 == old[j].fid1 && m->points[i].fid2 == old[j].fid2)
                m->points[i].data = old[j].data;
}
#endif
