// rp_narrowphase.hip — contact manifold generation and pair bookkeeping on device.
//
// One thread per live contact pair (NarrowPhase::compute_contacts, contacts.rs:22-251 dispatches
// pair_update::process_pair per edge; here the "rayon broadcast + atomic cursor" becomes a grid).
// Per pair (pair_update.rs:67-680): recycle test -> [full update: parry contact_manifolds ->
// combine friction/restitution -> 4-point reduction -> plane sort -> solver contacts -> anchor
// localisation / lever-arm freezing -> recycle state] -> begin/end-touch transition.
// The parry3d pieces (cuboid SAT + face clipping, ball cases, try_update_contacts, match_contacts)
// are not in /root/reference; they are implemented from the crate's published algorithm.
//
// Then (contacts.rs:300-385, narrow_phase/mod.rs:90-172) begin-touch pairs get a persistent colour by
// first-fit over per-body 128-bit masks in (min body, max body) order.  The serial greedy loop is
// reproduced exactly as a wavefront over the pairs' dependency DAG inside ONE workgroup (k_color_pairs):
// a pair is coloured once its predecessor (by key) at either of its dynamic bodies is, so it sees
// precisely the masks the serial order would have produced.  Sensor pairs (intersections.rs) are
// intersection-tested here too.
#include "rp_pairs.h"
#include <float.h>

#define PT(plane, k, s) plane[(size_t)(k) * w.pool_cap + (s)]

// The working manifold of one lane.  Its point arrays are indexed at run time (points are pushed, matched, reduced), which the
// compiler can only serve from scratch memory: 3.6 KB per lane in round 2, every access a memory round trip — one full update took
// ~70 us however few pairs a step queued, and the launch wrote ~130 MB of scratch back to HBM.  The arrays now live in LDS:
// NP_THREADS lanes per workgroup, element k of lane t at [k * NP_THREADS + t] (conflict-free for 4-byte items, 12-byte V3 items
// step 3 banks per lane).  NP_LDS_DWORDS dwords per lane: region A = the manifold (72), region B = the old ContactData while it is
// permuted (64), aliased before that by the scratch of try_update_contacts / match_contacts (40).
#define NP_THREADS 128
#define NP_LDS_DWORDS 136
template <typename T> struct LdsCol { T *p; RP_DEV T &operator[](int k) const { return p[k * NP_THREADS]; } };
struct LocalManifold {
    LdsCol<V3> lp1, lp2;
    LdsCol<float> dist;
    LdsCol<unsigned> fid;     // fid1 | fid2 << 16
    LdsCol<int> src;          // old point index whose ContactData this point inherits (-1 = fresh)
    // region B: 64 floats per lane, always accessed AS floats (one element type: the two uses below alias in time, and accesses of
    // different types to one address may be reordered by type-based alias analysis).  Generators: entries 3i..3i+2 = a point, 24 + i = a
    // distance, 32 + i = an old feature-id word; permutation of the old ContactData: 4k..4k+3 = impulses, 32 + 4k.. = warm-start vector.
    LdsCol<float> b;
    RP_DEV void b_set3(int i, V3 v) const { b[3 * i] = v.x; b[3 * i + 1] = v.y; b[3 * i + 2] = v.z; }
    RP_DEV V3 b_get3(int i) const { return v3(b[3 * i], b[3 * i + 1], b[3 * i + 2]); }
    RP_DEV void b_set4(int e, float4 v) const { b[e] = v.x; b[e + 1] = v.y; b[e + 2] = v.z; b[e + 3] = v.w; }
    RP_DEV float4 b_get4(int e) const { return make_float4(b[e], b[e + 1], b[e + 2], b[e + 3]); }
    RP_DEV void old_fid_set(int i, unsigned f) const { b[32 + i] = __uint_as_float(f); }
    RP_DEV unsigned old_fid(int i) const { return __float_as_uint(b[32 + i]); }
    int n;
    V3 ln1, ln2;
    RP_DEV void bind(float *lds) { // lds = this workgroup's NP_THREADS * NP_LDS_DWORDS floats
        float *a = lds, *b = lds + NP_THREADS * 72; // (local `b` = region B's base)
        lp1.p = (V3 *)a + threadIdx.x; lp2.p = (V3 *)(a + NP_THREADS * 24) + threadIdx.x;
        dist.p = a + NP_THREADS * 48 + threadIdx.x; fid.p = (unsigned *)(a + NP_THREADS * 56) + threadIdx.x; src.p = (int *)(a + NP_THREADS * 64) + threadIdx.x;
        this->b.p = b + threadIdx.x;
    }
};

struct Face { V3 v[4]; unsigned vid[4], eid[4], fid; };

RP_DEV V3 cuboid_support_point(V3 he, V3 d) { return v3(copysignf(he.x, d.x), copysignf(he.y, d.y), copysignf(he.z, d.z)); }
RP_DEV unsigned cuboid_vid(V3 v) { return (unsigned)((v.x < 0.0f) | ((v.y < 0.0f) << 1) | ((v.z < 0.0f) << 2)); }

__device__ Face cuboid_support_face(V3 he, V3 dir) {
    Face f;
    float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
    int iamax = 0; float best = ax;
    if (ay > best) { best = ay; iamax = 1; }
    if (az > best) { best = az; iamax = 2; }
    float sign = copysignf(1.0f, comp(dir, iamax));
    if (iamax == 0) {
        f.v[0] = v3(he.x * sign, he.y, he.z); f.v[1] = v3(he.x * sign, -he.y, he.z);
        f.v[2] = v3(he.x * sign, -he.y, -he.z); f.v[3] = v3(he.x * sign, he.y, -he.z);
    } else if (iamax == 1) {
        f.v[0] = v3(he.x, he.y * sign, he.z); f.v[1] = v3(-he.x, he.y * sign, he.z);
        f.v[2] = v3(-he.x, he.y * sign, -he.z); f.v[3] = v3(he.x, he.y * sign, -he.z);
    } else {
        f.v[0] = v3(he.x, he.y, he.z * sign); f.v[1] = v3(he.x, -he.y, he.z * sign);
        f.v[2] = v3(-he.x, -he.y, he.z * sign); f.v[3] = v3(-he.x, he.y, he.z * sign);
    }
    for (int i = 0; i < 4; ++i) f.vid[i] = cuboid_vid(f.v[i]);
    for (int i = 0; i < 4; ++i) {
        unsigned a = f.vid[i], b = f.vid[(i + 1) & 3];
        unsigned lo = a < b ? a : b, hi = a < b ? b : a;
        f.eid[i] = 8u + 8u * lo + hi;
    }
    f.fid = 100u + 2u * (unsigned)iamax + (sign < 0.0f ? 1u : 0u);
    return f;
}

__device__ float sat_normal_oneway(V3 he1, V3 he2, Pose pos12, V3 &out_dir) {
    float best = -FLT_MAX; V3 best_dir = v3(0, 0, 0);
    for (int i = 0; i < 3; ++i) {
        float sign = copysignf(1.0f, comp(pos12.t, i));
        V3 axis1 = v3(i == 0 ? sign : 0.0f, i == 1 ? sign : 0.0f, i == 2 ? sign : 0.0f);
        V3 axis2 = qrot_inv(pos12.r, -axis1);
        V3 pt2 = pose_tp(pos12, cuboid_support_point(he2, axis2));
        float sep = comp(pt2, i) * sign - comp(he1, i);
        if (sep > best) { best = sep; best_dir = axis1; }
    }
    out_dir = best_dir;
    return best;
}
__device__ float sat_edge_twoway(V3 he1, V3 he2, Pose pos12, V3 &out_dir) {
    float best = -FLT_MAX; V3 best_dir = v3(0, 0, 0);
    V3 c[3] = {qrot(pos12.r, v3(1, 0, 0)), qrot(pos12.r, v3(0, 1, 0)), qrot(pos12.r, v3(0, 0, 1))};
    for (int k = 0; k < 9; ++k) {
        V3 e = c[k / 3];
        int a = k % 3;
        V3 axis = a == 0 ? v3(0, -e.z, e.y) : (a == 1 ? v3(e.z, 0, -e.x) : v3(-e.y, e.x, 0));
        float n = len(axis);
        if (n > FLT_EPSILON) {
            V3 axis1 = axis * (1.0f / n);
            float signum = copysignf(1.0f, dot(pos12.t, axis1));
            axis1 = axis1 * signum;
            V3 axis2 = qrot_inv(pos12.r, -axis1);
            V3 lp1 = cuboid_support_point(he1, axis1);
            V3 pt2 = pose_tp(pos12, cuboid_support_point(he2, axis2));
            float sep = dot(pt2 - lp1, axis1);
            if (sep > best) { best = sep; best_dir = axis1; }
        }
    }
    out_dir = best_dir;
    return best;
}

RP_DEV bool ulps_eq(float a, float b) {
    if (fabsf(a - b) <= FLT_EPSILON) return true;
    if ((a < 0) != (b < 0)) return false;
    int ia = __float_as_int(a), ib = __float_as_int(b);
    int d = ia > ib ? ia - ib : ib - ia;
    return d <= 4;
}
__device__ bool closest_points_line2d(float a0x, float a0y, float a1x, float a1y, float b0x, float b0y, float b1x, float b1y,
                                      float &s_out, float &t_out) {
    float d1x = a1x - a0x, d1y = a1y - a0y, d2x = b1x - b0x, d2y = b1y - b0y;
    float rx = a0x - b0x, ry = a0y - b0y;
    float a = d1x * d1x + d1y * d1y, e = d2x * d2x + d2y * d2y, f = d2x * rx + d2y * ry;
    const float eps = FLT_EPSILON;
    if (a <= eps && e <= eps) { s_out = 0; t_out = 0; return true; }
    if (a <= eps) { s_out = 0; t_out = f / e; return true; }
    float c = d1x * rx + d1y * ry;
    if (e <= eps) { s_out = -c / a; t_out = 0; return true; }
    float b = d1x * d2x + d1y * d2y;
    float ae = a * e, bb = b * b, denom = ae - bb;
    bool parallel = denom <= eps || ulps_eq(ae, bb);
    if (parallel) return false;
    float s = (b * f - c * e) / denom;
    s_out = s; t_out = (b * s + f) / e;
    return true;
}

RP_DEV void lm_push(LocalManifold &m, V3 p1, V3 p2, unsigned f1, unsigned f2, float dist) {
    if (m.n >= RP_MAX_PTS) return;
    int i = m.n++;
    m.lp1[i] = p1; m.lp2[i] = p2; m.dist[i] = dist; m.fid[i] = f1 | (f2 << 16); m.src[i] = -1;
}

#define PERP(ax, ay, bx, by) ((ax) * (by) - (ay) * (bx))
__device__ void contacts_face_face(Pose pos12, const Face &f1, V3 sep, const Face &f2, LocalManifold &m) {
    V3 b0, b1; orthonormal_basis(sep, b0, b1);
    float p1x[4], p1y[4], p2x[4], p2y[4];
    for (int i = 0; i < 4; ++i) { p1x[i] = dot(f1.v[i], b0); p1y[i] = dot(f1.v[i], b1); p2x[i] = dot(f2.v[i], b0); p2y[i] = dot(f2.v[i], b1); }
    {
        V3 normal2_1 = cross(f2.v[2] - f2.v[1], f2.v[0] - f2.v[1]);
        float denom = dot(normal2_1, sep);
        if (!(fabsf(denom) <= FLT_EPSILON)) {
            for (int i = 0; i < 4; ++i) {
                float px = p1x[i], py = p1y[i];
                float sign = PERP(p2x[0] - p2x[3], p2y[0] - p2y[3], px - p2x[3], py - p2y[3]);
                bool outside = false;
                for (int j = 0; j < 3; ++j) {
                    float ns = PERP(p2x[j + 1] - p2x[j], p2y[j + 1] - p2y[j], px - p2x[j], py - p2y[j]);
                    if (sign == 0.0f) sign = ns; else if (sign * ns < 0.0f) { outside = true; break; }
                }
                if (outside) continue;
                float dist = dot(f2.v[0] - f1.v[i], normal2_1) / denom;
                V3 lp2_1 = f1.v[i] + sep * dist;
                lm_push(m, f1.v[i], pose_itp(pos12, lp2_1), f1.vid[i], f2.fid, dist);
            }
        }
    }
    {
        V3 normal1 = cross(f1.v[2] - f1.v[1], f1.v[0] - f1.v[1]);
        float denom = -dot(normal1, sep);
        if (!(fabsf(denom) <= FLT_EPSILON)) {
            for (int i = 0; i < 4; ++i) {
                float px = p2x[i], py = p2y[i];
                float sign = PERP(p1x[0] - p1x[3], p1y[0] - p1y[3], px - p1x[3], py - p1y[3]);
                bool outside = false;
                for (int j = 0; j < 3; ++j) {
                    float ns = PERP(p1x[j + 1] - p1x[j], p1y[j + 1] - p1y[j], px - p1x[j], py - p1y[j]);
                    if (sign == 0.0f) sign = ns; else if (sign * ns < 0.0f) { outside = true; break; }
                }
                if (outside) continue;
                float dist = dot(f1.v[0] - f2.v[i], normal1) / denom;
                V3 lp1 = f2.v[i] - sep * dist;
                lm_push(m, lp1, pose_itp(pos12, f2.v[i]), f1.fid, f2.vid[i], dist);
            }
        }
    }
    for (int j = 0; j < 4; ++j) {
        int j1 = (j + 1) & 3;
        for (int i = 0; i < 4; ++i) {
            int i1 = (i + 1) & 3;
            float s, t;
            if (closest_points_line2d(p1x[i], p1y[i], p1x[i1], p1y[i1], p2x[j], p2y[j], p2x[j1], p2y[j1], s, t)) {
                if (s > 0.0f && s < 1.0f && t > 0.0f && t < 1.0f) {
                    V3 lp1 = f1.v[i] * (1.0f - s) + f1.v[i1] * s;
                    V3 lp2_1 = f2.v[j] * (1.0f - t) + f2.v[j1] * t;
                    float dist = dot(lp2_1 - lp1, sep);
                    lm_push(m, lp1, pose_itp(pos12, lp2_1), f1.eid[i], f2.eid[j], dist);
                }
            }
        }
    }
}
#undef PERP

// ContactManifold::try_update_contacts (cos 1 degree, 1e-6 squared distance)
__device__ bool try_update_contacts(LocalManifold &m, Pose pos12) {
    if (m.n == 0) return false;
    V3 ln2 = qrot(pos12.r, m.ln2);
    if (-dot(m.ln1, ln2) < 0.99984769515f) return false;
    for (int i = 0; i < m.n; ++i) {
        V3 lp2 = pose_tp(pos12, m.lp2[i]);
        float dist = dot(lp2 - m.lp1[i], m.ln1);
        if (dist * m.dist[i] < 0.0f) return false;
        V3 np1 = lp2 - m.ln1 * dist;
        if (len2(m.lp1[i] - np1) > 1.0e-6f) return false;
        m.b_set3(i, np1); m.b[24 + i] = dist;
    }
    for (int i = 0; i < m.n; ++i) { m.lp1[i] = m.b_get3(i); m.dist[i] = m.b[24 + i]; }
    return true;
}

__device__ void manifold_cuboid_cuboid(Pose pos12, V3 he1, V3 he2, float prediction, LocalManifold &m) {
    if (try_update_contacts(m, pos12)) return;
    Pose pos21 = pose_inv(pos12);
    V3 d1, d2, d3;
    float s1 = sat_normal_oneway(he1, he2, pos12, d1);
    if (s1 > prediction) { m.n = 0; return; }
    float s2 = sat_normal_oneway(he2, he1, pos21, d2);
    if (s2 > prediction) { m.n = 0; return; }
    float s3 = sat_edge_twoway(he1, he2, pos12, d3);
    if (s3 > prediction) { m.n = 0; return; }
    V3 best = d1;
    if (s2 > s1 && s2 > s3) best = qrot(pos12.r, -d2);
    else if (s3 > s1) best = d3;
    V3 ln2 = qrot(pos21.r, -best);
    Face f1 = cuboid_support_face(he1, best);
    Face f2 = cuboid_support_face(he2, ln2);
    for (int i = 0; i < 4; ++i) f2.v[i] = pose_tp(pos12, f2.v[i]);
    int nold = m.n;
    for (int i = 0; i < nold; ++i) m.old_fid_set(i, m.fid[i]);
    m.n = 0;
    contacts_face_face(pos12, f1, best, f2, m);
    m.ln1 = best; m.ln2 = ln2;
    // match_contacts: inherit tracked data by (fid1, fid2); the LAST matching old point wins
    for (int i = 0; i < m.n; ++i)
        for (int j = 0; j < nold; ++j)
            if (m.fid[i] == m.old_fid(j)) m.src[i] = j;
}

__device__ void manifold_ball_ball(Pose pos12, float r1, float r2, float prediction, LocalManifold &m) {
    float l = len(pos12.t);
    float dist = l - r1 - r2;
    if (dist < prediction) {
        V3 n1 = l > 0.0f ? pos12.t * (1.0f / l) : v3(0, 1, 0);
        V3 n2 = qrot_inv(pos12.r, -n1);
        int keep = m.n != 0 ? 0 : -1;
        m.n = 1; m.lp1[0] = n1 * r1; m.lp2[0] = n2 * r2; m.dist[0] = dist; m.fid[0] = 0; m.src[0] = keep;
        m.ln1 = n1; m.ln2 = n2;
    } else m.n = 0;
}
// the tail every convex-vs-ball generator shares (contact_manifold_convex_ball): proj = the ball centre projected on shape 1
// (non-solid), inside = the centre lies in shape 1 (normal and distance negated)
RP_DEV void convex_ball_finish(Pose pos12, V3 pt, V3 proj, bool inside, float r2, float prediction, LocalManifold &m, bool flipped) {
    V3 dpos = pt - proj;
    float dist = len(dpos);
    if (!(dist > 0.0f)) return; // Unit::try_new_and_get(dpos, 0.0) fails: manifold left untouched
    V3 n1 = dpos * (1.0f / dist);
    if (inside) { n1 = -n1; dist = -dist; }
    if (dist <= r2 + prediction) {
        V3 n2 = qrot_inv(pos12.r, -n1);
        V3 p2 = n2 * r2;
        int keep = m.n == 1 ? 0 : -1;
        m.n = 1;
        m.lp1[0] = flipped ? p2 : proj; m.lp2[0] = flipped ? proj : p2; m.dist[0] = dist - r2;
        if (keep < 0) m.fid[0] = RP_FID_UNKNOWN | (RP_FID_UNKNOWN << 16);
        m.src[0] = keep;
        if (flipped) { m.ln1 = n2; m.ln2 = n1; } else { m.ln1 = n1; m.ln2 = n2; }
    } else m.n = 0;
}
// contact_manifold_convex_ball with shape1 = cuboid (Aabb::project_local_point, non-solid: a centre inside the cuboid projects onto the
// nearest face); flipped = ball is collider 1
__device__ void manifold_cuboid_ball(Pose pos12, V3 he1, float r2, float prediction, LocalManifold &m, bool flipped) {
    V3 pt = pos12.t;
    V3 mins_pt = -he1 - pt, pt_maxs = pt - he1;
    V3 shift = v3(rp_max(mins_pt.x, 0.0f) - rp_max(pt_maxs.x, 0.0f), rp_max(mins_pt.y, 0.0f) - rp_max(pt_maxs.y, 0.0f),
                  rp_max(mins_pt.z, 0.0f) - rp_max(pt_maxs.z, 0.0f));
    bool inside = shift.x == 0.0f && shift.y == 0.0f && shift.z == 0.0f;
    if (inside) { // nearest face: the largest (closest to zero) of the six negative slacks
        float best = -FLT_MAX; int best_id = 0; bool is_mins = false;
        for (int i = 0; i < 3; ++i) {
            float mp = comp(mins_pt, i), pm = comp(pt_maxs, i);
            if (mp < pm) { if (pm > best) { best_id = i; is_mins = false; best = pm; } }
            else if (mp > best) { best_id = i; is_mins = true; best = mp; }
        }
        const float sv = is_mins ? best : -best;
        shift = v3(best_id == 0 ? sv : 0.0f, best_id == 1 ? sv : 0.0f, best_id == 2 ? sv : 0.0f);
    }
    convex_ball_finish(pos12, pt, pt + shift, inside, r2, prediction, m, flipped);
}
// ---- capsules (parry shape::Capsule = segment [a, b] + radius; c_he = (half_height, radius, axis)) — restated like oracle/ro_shapes.h ----
RP_DEV V3 segment_project_point(V3 a, V3 b, V3 pt) { // Segment::project_local_point
    V3 ab = b - a, ap = pt - a;
    float ab_ap = dot(ab, ap), sqnab = dot(ab, ab);
    if (ab_ap <= 0.0f) return a;
    if (ab_ap >= sqnab) return b;
    float u = ab_ap / sqnab;
    return a + ab * u;
}
// closest_points_segment_segment_with_locations_nD (Ericson 5.1.9)
__device__ void closest_points_segment_segment(V3 a1, V3 b1, V3 a2, V3 b2, float &s_out, float &t_out) {
    V3 d1 = b1 - a1, d2 = b2 - a2, r = a1 - a2;
    float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    const float eps = FLT_EPSILON;
    float s, t;
    if (a <= eps && e <= eps) { s = 0.0f; t = 0.0f; }
    else if (a <= eps) { s = 0.0f; t = rp_clamp(f / e, 0.0f, 1.0f); }
    else {
        float c = dot(d1, r);
        if (e <= eps) { t = 0.0f; s = rp_clamp(-c / a, 0.0f, 1.0f); }
        else {
            float b = dot(d1, d2);
            float ae = a * e, bb = b * b, denom = ae - bb;
            if (denom > eps && !ulps_eq(ae, bb)) s = rp_clamp((b * f - c * e) / denom, 0.0f, 1.0f); else s = 0.0f;
            t = (b * s + f) / e;
            if (t < 0.0f) { t = 0.0f; s = rp_clamp(-c / a, 0.0f, 1.0f); }
            else if (t > 1.0f) { t = 1.0f; s = rp_clamp((b - c) / a, 0.0f, 1.0f); }
        }
    }
    s_out = s; t_out = t;
}
// contact_manifold_capsule_capsule (3-D): one contact between the closest points of the two segments
__device__ void manifold_capsule_capsule(Pose pos12, float4 c1, float4 c2, float prediction, LocalManifold &m) {
    V3 e1 = capsule_axis_dir((int)c1.z), e2 = capsule_axis_dir((int)c2.z);
    V3 a1 = e1 * -c1.x, b1 = e1 * c1.x;
    V3 a2 = pose_tp(pos12, e2 * -c2.x), b2 = pose_tp(pos12, e2 * c2.x);
    float s, t; closest_points_segment_segment(a1, b1, a2, b2, s, t);
    V3 p1 = a1 * (1.0f - s) + b1 * s, p2_1 = a2 * (1.0f - t) + b2 * t;
    V3 d = p2_1 - p1;
    float l = len(d);
    V3 n1 = l > FLT_EPSILON ? d * (1.0f / l) : v3(0, 1, 0);
    float dist = dot(d, n1) - c1.y - c2.y;
    if (dist <= prediction) {
        V3 n2 = qrot_inv(pos12.r, -n1);
        int keep = m.n != 0 ? 0 : -1;
        m.n = 1; m.lp1[0] = p1 + n1 * c1.y; m.lp2[0] = pose_itp(pos12, p2_1) + n2 * c2.y; m.dist[0] = dist; m.fid[0] = 0; m.src[0] = keep;
        m.ln1 = n1; m.ln2 = n2;
    } else m.n = 0;
}
// contact_manifold_convex_ball with shape1 = capsule (Capsule::project_local_point, non-solid: a centre inside the capsule projects
// onto its surface along the direction from the segment); flipped = the ball is collider 1
__device__ void manifold_capsule_ball(Pose pos12, float4 c1, float r2, float prediction, LocalManifold &m, bool flipped) {
    V3 e1 = capsule_axis_dir((int)c1.z);
    V3 pt = pos12.t;
    V3 sp = segment_project_point(e1 * -c1.x, e1 * c1.x, pt);
    V3 dproj = pt - sp;
    float dseg = len(dproj);
    bool inside; V3 proj;
    if (dseg > FLT_EPSILON) { inside = dseg <= c1.y; proj = sp + (dproj * (1.0f / dseg)) * c1.y; }
    else { inside = true; proj = sp + v3(0, c1.y, 0); } // centre on the segment: pushed along +Y
    convex_ball_finish(pos12, pt, proj, inside, r2, prediction, m, flipped);
}
// ---- half-spaces (parry shape::HalfSpace{normal}; c_he = the unit outward normal) — restated like oracle/ro_shapes.h ----
// contact_manifold_convex_ball with shape1 = half-space (HalfSpace::project_local_point); flipped = the ball is collider 1
__device__ void manifold_halfspace_ball(Pose pos12, V3 normal1, float r2, float prediction, LocalManifold &m, bool flipped) {
    V3 pt = pos12.t;
    float dd = dot(normal1, pt);
    V3 proj = pt + (-normal1) * dd;
    convex_ball_finish(pos12, pt, proj, dd <= 0.0f, r2, prediction, m, flipped);
}
// contact_manifold_halfspace_pfm: shape 2 = a cuboid (its support face toward the plane) or a capsule (its segment, border radius =
// the capsule's); every feature vertex within `prediction` of the plane is a contact.  pos12 = pose of shape 2 in the half-space's
// frame; flipped = the half-space is collider 2
__device__ void manifold_halfspace_pfm(Pose pos12, V3 normal1, int sh2, float4 c2, float prediction, LocalManifold &m, bool flipped) {
    V3 normal1_2 = qrot_inv(pos12.r, normal1);
    V3 vtx[4]; unsigned vid[4]; int nv; float border = 0.0f;
    if (sh2 == RP_SHAPE_CAPSULE) {
        V3 e2 = capsule_axis_dir((int)c2.z);
        vtx[0] = e2 * -c2.x; vtx[1] = e2 * c2.x; vid[0] = 0u; vid[1] = 2u; vtx[2] = vtx[1]; vtx[3] = vtx[1]; vid[2] = vid[3] = 2u; nv = 2; border = c2.y;
    } else {
        Face f = cuboid_support_face(v3(c2), -normal1_2);
        for (int i = 0; i < 4; ++i) { vtx[i] = f.v[i]; vid[i] = f.vid[i]; }
        nv = 4;
    }
    int nold = m.n;
    for (int i = 0; i < nold; ++i) m.old_fid_set(i, m.fid[i]);
    m.n = 0;
    for (int i = 0; i < nv; ++i) {
        V3 vtx2_1 = pose_tp(pos12, vtx[i]);
        float dist_to_plane = dot(vtx2_1, normal1);
        if (dist_to_plane - border <= prediction) {
            V3 p1 = vtx2_1 - normal1 * dist_to_plane;
            V3 p2 = vtx[i] - normal1_2 * border;
            if (flipped) lm_push(m, p2, p1, vid[i], 0u, dist_to_plane - border);
            else lm_push(m, p1, p2, 0u, vid[i], dist_to_plane - border);
        }
    }
    if (flipped) { m.ln1 = -normal1_2; m.ln2 = normal1; } else { m.ln1 = normal1; m.ln2 = -normal1_2; }
    for (int i = 0; i < m.n; ++i)
        for (int j = 0; j < nold; ++j)
            if (m.fid[i] == m.old_fid(j)) m.src[i] = j;
}
// sat::cuboid_support_map_find_local_separating_normal_oneway with shape2 = the segment [a2, b2] (cuboid frame)
__device__ float sat_cuboid_segment_normal_oneway(V3 he1, V3 a2, V3 b2, V3 &out_dir) {
    float best = -FLT_MAX; V3 best_dir = v3(0, 0, 0);
    for (int i = 0; i < 3; ++i)
        for (int sg = 0; sg < 2; ++sg) {
            float sign = sg == 0 ? -1.0f : 1.0f;
            V3 axis1 = v3(i == 0 ? sign : 0.0f, i == 1 ? sign : 0.0f, i == 2 ? sign : 0.0f);
            V3 dir = -axis1;
            V3 pt2 = dot(a2, dir) > dot(b2, dir) ? a2 : b2;
            float sep = comp(pt2, i) * sign - comp(he1, i);
            if (sep > best) { best = sep; best_dir = axis1; }
        }
    out_dir = best_dir;
    return best;
}
// cuboid_segment_find_local_separating_edge_twoway (cuboid_support_map_compute_separation_wrt_local_line, both directions)
__device__ float sat_cuboid_segment_edge_twoway(V3 he1, V3 a2, V3 b2, V3 &out_dir) {
    float best = -FLT_MAX; V3 best_dir = v3(0, 0, 0);
    V3 x2 = b2 - a2;
    V3 axes[3] = {v3(0, -x2.z, x2.y), v3(x2.z, 0, -x2.x), v3(-x2.y, x2.x, 0)};
    for (int k = 0; k < 3; ++k) {
        float n = len(axes[k]);
        if (!(n > FLT_EPSILON)) continue;
        V3 axis1 = axes[k] * (1.0f / n);
        V3 lp1 = cuboid_support_point(he1, axis1);
        V3 q = dot(a2, -axis1) > dot(b2, -axis1) ? a2 : b2;
        float sep1 = dot(q - lp1, axis1);
        V3 naxis = -axis1;
        V3 lp1b = cuboid_support_point(he1, naxis);
        V3 qb = dot(a2, axis1) > dot(b2, axis1) ? a2 : b2;
        float sep2 = dot(qb - lp1b, naxis);
        float sep = sep1 > sep2 ? sep1 : sep2;
        V3 ax = sep1 > sep2 ? axis1 : naxis;
        if (sep > best) { best = sep; best_dir = ax; }
    }
    out_dir = best_dir;
    return best;
}
// contact_manifold_cuboid_capsule: pos12 = pose of the capsule in the cuboid's frame, upd = pose of collider 2 in collider 1's
// frame (== pos12 unless flipped), flipped = the capsule is collider 1 (points, feature ids and normals swap on output)
#define PERP(ax, ay, bx, by) ((ax) * (by) - (ay) * (bx))
__device__ void manifold_cuboid_capsule(Pose pos12, Pose upd, V3 he1, float4 c2, float prediction, LocalManifold &m, bool flipped) {
    if (try_update_contacts(m, upd)) return;
    Pose pos21 = pose_inv(pos12);
    const float hh2 = c2.x, r2 = c2.y;
    V3 e2 = capsule_axis_dir((int)c2.z);
    V3 a2 = pose_tp(pos12, e2 * -hh2), b2 = pose_tp(pos12, e2 * hh2);
    V3 d1, d3;
    float s1 = sat_cuboid_segment_normal_oneway(he1, a2, b2, d1);
    if (s1 > r2 + prediction) { m.n = 0; return; }
    float s3 = sat_cuboid_segment_edge_twoway(he1, a2, b2, d3);
    if (s3 > r2 + prediction) { m.n = 0; return; }
    V3 best = s3 > s1 ? d3 : d1;
    V3 n2 = qrot(pos21.r, -best);
    Face f1 = cuboid_support_face(he1, best);
    int nold = m.n;
    for (int i = 0; i < nold; ++i) m.old_fid_set(i, m.fid[i]);
    m.n = 0;
    V3 bx, by; orthonormal_basis(best, bx, by);
    float p1x[4], p1y[4];
    for (int i = 0; i < 4; ++i) { p1x[i] = dot(f1.v[i], bx); p1y[i] = dot(f1.v[i], by); }
    const float s0x = dot(a2, bx), s0y = dot(a2, by), s1x = dot(b2, bx), s1y = dot(b2, by);
    { // segment vertices inside the face
        V3 normal1 = cross(f1.v[2] - f1.v[1], f1.v[0] - f1.v[1]);
        float denom = -dot(normal1, best);
        if (!(fabsf(denom) <= FLT_EPSILON)) {
            for (int i = 0; i < 2; ++i) {
                float px = i == 0 ? s0x : s1x, py = i == 0 ? s0y : s1y;
                V3 sv = i == 0 ? a2 : b2;
                float sign = PERP(p1x[0] - p1x[3], p1y[0] - p1y[3], px - p1x[3], py - p1y[3]);
                bool outside = false;
                for (int j = 0; j < 3; ++j) {
                    float ns = PERP(p1x[j + 1] - p1x[j], p1y[j + 1] - p1y[j], px - p1x[j], py - p1y[j]);
                    if (sign == 0.0f) sign = ns; else if (sign * ns < 0.0f) { outside = true; break; }
                }
                if (outside) continue;
                float dist = dot(f1.v[0] - sv, normal1) / denom;
                V3 lp1 = sv - best * dist;
                lm_push(m, lp1, pose_itp(pos12, sv), f1.fid, i == 0 ? 0u : 2u, dist);
            }
        }
    }
    for (int i = 0; i < 4; ++i) { // the segment against the face's edges (one pass over the segment)
        int i1 = (i + 1) & 3;
        float s, t;
        if (closest_points_line2d(p1x[i], p1y[i], p1x[i1], p1y[i1], s0x, s0y, s1x, s1y, s, t) && s > 0.0f && s < 1.0f && t > 0.0f && t < 1.0f) {
            V3 lp1 = f1.v[i] * (1.0f - s) + f1.v[i1] * s;
            V3 lp2_1 = a2 * (1.0f - t) + b2 * t;
            float dist = dot(lp2_1 - lp1, best);
            lm_push(m, lp1, pose_itp(pos12, lp2_1), f1.eid[i], 1u, dist);
        }
    }
    for (int i = 0; i < m.n; ++i) {
        m.lp2[i] = m.lp2[i] + n2 * r2; // push the segment point out to the capsule's surface
        m.dist[i] = m.dist[i] - r2;
        if (flipped) { V3 tp = m.lp1[i]; m.lp1[i] = m.lp2[i]; m.lp2[i] = tp; m.fid[i] = (m.fid[i] >> 16) | (m.fid[i] << 16); }
    }
    if (flipped) { m.ln1 = n2; m.ln2 = best; } else { m.ln1 = best; m.ln2 = n2; }
    for (int i = 0; i < m.n; ++i)
        for (int j = 0; j < nold; ++j)
            if (m.fid[i] == m.old_fid(j)) m.src[i] = j;
}
#undef PERP

// ---- cylinders, cones: GJK / EPA + polygonal feature maps (only instantiated in the CONVEX kernels) ----
#include "rp_convex.h"

// manifold_reduction::reduce_manifold_naive — geometry/manifold_reduction.rs:4-84
__device__ void reduce_manifold(const LocalManifold &m, int sel[4], int &nsel, float prediction) {
    if (m.n <= 4) return;
    sel[0] = sel[1] = sel[2] = sel[3] = -1;
    float deepest = FLT_MAX;
    for (int i = 0; i < m.n; ++i) if (m.dist[i] < deepest) { deepest = m.dist[i]; sel[0] = i; }
    if (sel[0] < 0) { nsel = 0; return; }
    V3 a = m.lp1[sel[0]];
    float furthest = -FLT_MAX;
    for (int i = 0; i < m.n; ++i) {
        float d = len2(m.lp1[i] - a);
        if (i != sel[0] && m.dist[i] <= prediction && d > furthest) { furthest = d; sel[1] = i; }
    }
    if (sel[1] < 0) { nsel = 1; return; }
    V3 b = m.lp1[sel[1]];
    if (a.x == b.x && a.y == b.y && a.z == b.z) { nsel = 1; return; }
    V3 tangent = cross(b - a, m.ln1);
    float mind = FLT_MAX, maxd = -FLT_MAX;
    for (int i = 0; i < m.n; ++i) {
        if (i == sel[0] || i == sel[1] || m.dist[i] > prediction) continue;
        float d = dot(m.lp1[i] - a, tangent);
        if (d < mind) { mind = d; sel[2] = i; }
        if (d > maxd) { maxd = d; sel[3] = i; }
    }
    if (sel[2] < 0) nsel = 2; else if (sel[2] == sel[3]) nsel = 3; else nsel = 4;
}

// max(|mins|, |maxs|) of the shape's local AABB (the recycle extent of pair_update.rs:582-613)
RP_DEV float shape_origin_radius(int sh, float4 he, float border = 0.0f) {
    if (sh >= RP_SHAPE_ROUND_CUBOID) return len(v3(he.x + border, he.y + border, he.z + border)); // the inner local box loosened by the border
    if (sh == RP_SHAPE_CUBOID || sh >= RP_SHAPE_CYLINDER) return len(v3(he)); // (cylinder / cone: he = the local AABB's half extents)
    if (sh == RP_SHAPE_CAPSULE) { int ax = (int)he.z; return len(v3(ax == 0 ? he.x + he.y : he.y, ax == 1 ? he.x + he.y : he.y, ax == 2 ? he.x + he.y : he.y)); }
    if (sh == RP_SHAPE_HALFSPACE) return INFINITY; // |(MAX/2, MAX/2, MAX/2)| overflows: a pair with a half-space never recycles
    return len(v3(he.x, he.x, he.x));
}
RP_DEV float combine_coeff(float a, float b, int ra, int rb) {
    int rule = ra > rb ? ra : rb; // coefficient_combine_rule.rs:58-86
    switch (rule) {
    case RP_RULE_AVERAGE: return (a + b) / 2.0f;
    case RP_RULE_MIN: return fabsf(a < b ? a : b);
    case RP_RULE_MULTIPLY: return a * b;
    case RP_RULE_MAX: return a > b ? a : b;
    case RP_RULE_CLAMPED_SUM: return rp_clamp(a + b, 0.0f, 1.0f);
    default: return sqrtf(rp_max(a, 0.0f) * rp_max(b, 0.0f));
    }
}
RP_DEV int effective_dominance(const DevWorld &w, int body) {
    if (body >= 0) { int f = w.b_flags[body]; if ((f & RP_BF_TYPE_MASK) != RP_BODY_FIXED) return (int)(signed char)((f >> RP_BF_DOM_SHIFT) & 0xff); }
    return 128;
}
// a side that takes part in the colouring: any non-fixed body (assign_pair_solver_color, mod.rs:104-105)
RP_DEV bool body_dynamic(const DevWorld &w, int body) { return body >= 0 && (w.b_flags[body] & RP_BF_TYPE_MASK) != RP_BODY_FIXED; }

// parry intersection_test for the three shapes (sensor pairs, narrow_phase/intersections.rs:120-160): ball-ball (centre distance),
// cuboid-cuboid (no separating axis among the 3 + 3 face normals and the 9 edge cross products), a ball against a convex shape (solid
// point projection), capsule-capsule (segment distance); cuboid-capsule (GJK in parry) minimises the distance from the capsule's
// segment to the box over the segment parameter (a convex function: ternary search, 48 fixed iterations).  Same arithmetic as
// the oracle's shapes_intersect.
RP_DEV float point_box_dist2(V3 p, V3 he) {
    float dx = rp_max(fabsf(p.x) - he.x, 0.0f), dy = rp_max(fabsf(p.y) - he.y, 0.0f), dz = rp_max(fabsf(p.z) - he.z, 0.0f);
    return dx * dx + dy * dy + dz * dz;
}
template <bool CONVEX> __device__ __noinline__ bool shapes_intersect(const DevWorld &w, int s1, float4 h1, int s2, float4 h2, Pose pos12, float border1, float border2) {
    if constexpr (CONVEX) {
        if (s1 >= RP_SHAPE_CYLINDER || s2 >= RP_SHAPE_CYLINDER) { // cylinders, cones: GJK (intersection_test_support_map_support_map)
            const SmShape a = sm_shape_of(w, s1, h1, border1), b = sm_shape_of(w, s2, h2, border2);
            if (s1 == RP_SHAPE_HALFSPACE) return dot(v3(h1), pose_tp(pos12, sm_support(b, qrot_inv(pos12.r, -v3(h1))))) - b.border <= 0.0f;
            if (s2 == RP_SHAPE_HALFSPACE) { Pose pos21 = pose_inv(pos12); return dot(v3(h2), pose_tp(pos21, sm_support(a, qrot_inv(pos21.r, -v3(h2))))) - a.border <= 0.0f; }
            return sm_intersects(a, b, pos12);
        }
    }
    if (s1 > s2) { int ts = s1; s1 = s2; s2 = ts; float4 th = h1; h1 = h2; h2 = th; pos12 = pose_inv(pos12); } // ball < cuboid < capsule < half-space
    if (s2 == RP_SHAPE_HALFSPACE) { // intersection_test_support_map_halfspace: the support point toward -normal lies in the solid side
        if (s1 == RP_SHAPE_HALFSPACE) return false;
        Pose pos21 = pose_inv(pos12);
        V3 n = v3(h2), dir = qrot_inv(pos21.r, -n), sp;
        if (s1 == RP_SHAPE_BALL) sp = dir * h1.x;
        else if (s1 == RP_SHAPE_CUBOID) sp = cuboid_support_point(v3(h1), dir);
        else {
            V3 e = capsule_axis_dir((int)h1.z), a = e * -h1.x, b = e * h1.x;
            sp = (dot(dir, a) > dot(dir, b) ? a : b) + dir * h1.y;
        }
        return dot(n, pose_tp(pos21, sp)) <= 0.0f;
    }
    if (s1 == RP_SHAPE_BALL && s2 == RP_SHAPE_BALL) { float r = h1.x + h2.x; return dot(pos12.t, pos12.t) <= r * r; }
    if (s1 == RP_SHAPE_BALL && s2 == RP_SHAPE_CUBOID) { V3 c = pose_itp(pos12, v3(0, 0, 0)); return point_box_dist2(c, v3(h2)) <= h1.x * h1.x; }
    if (s1 == RP_SHAPE_BALL && s2 == RP_SHAPE_CAPSULE) {
        V3 c = pose_itp(pos12, v3(0, 0, 0)), e = capsule_axis_dir((int)h2.z);
        V3 q = segment_project_point(e * -h2.x, e * h2.x, c);
        float r = h1.x + h2.y; V3 d = c - q;
        return dot(d, d) <= r * r;
    }
    if (s1 == RP_SHAPE_CUBOID && s2 == RP_SHAPE_CUBOID) {
        V3 d;
        if (sat_normal_oneway(v3(h1), v3(h2), pos12, d) > 0.0f) return false;
        if (sat_normal_oneway(v3(h2), v3(h1), pose_inv(pos12), d) > 0.0f) return false;
        if (sat_edge_twoway(v3(h1), v3(h2), pos12, d) > 0.0f) return false;
        return true;
    }
    if (s1 == RP_SHAPE_CUBOID && s2 == RP_SHAPE_CAPSULE) {
        V3 e = capsule_axis_dir((int)h2.z);
        V3 a = pose_tp(pos12, e * -h2.x), b = pose_tp(pos12, e * h2.x);
        float lo = 0.0f, hi = 1.0f;
        for (int it = 0; it < 48; ++it) {
            float m1 = lo + (hi - lo) / 3.0f, m2 = hi - (hi - lo) / 3.0f;
            float d1 = point_box_dist2(a * (1.0f - m1) + b * m1, v3(h1)), d2 = point_box_dist2(a * (1.0f - m2) + b * m2, v3(h1));
            if (d1 <= d2) hi = m2; else lo = m1;
        }
        float t = 0.5f * (lo + hi);
        return point_box_dist2(a * (1.0f - t) + b * t, v3(h1)) <= h2.y * h2.y;
    }
    V3 e1 = capsule_axis_dir((int)h1.z), e2 = capsule_axis_dir((int)h2.z);
    V3 a1 = e1 * -h1.x, b1 = e1 * h1.x, a2 = pose_tp(pos12, e2 * -h2.x), b2 = pose_tp(pos12, e2 * h2.x);
    float sp, tp; closest_points_segment_segment(a1, b1, a2, b2, sp, tp);
    V3 d = (a2 * (1.0f - tp) + b2 * tp) - (a1 * (1.0f - sp) + b1 * sp);
    float r = h1.y + h2.y;
    return dot(d, d) <= r * r;
}

// A sensor pair lives in the intersection graph (narrow_phase/intersections.rs:17-175): no manifold, no solver contact, no colour, no
// wake-up; it is re-tested while one of its bodies may have moved and raises Started / Stopped | SENSOR on a change.
template <bool CONVEX> __device__ __forceinline__ void sensor_pair_update(DevWorld &w, int s, int c1, int c2, Pose pos12) {
    const int rb1 = w.c_parent[c1], rb2 = w.c_parent[c2];
    int pf = w.p_pflags[s] & ~RP_PF_RECYCLE;
    const bool had_i = (pf & RP_PF_INTERSECTING) != 0;
    const bool now_i = (rb1 == rb2 && rb1 >= 0) ? false : shapes_intersect<CONVEX>(w, w.c_shape[c1], w.c_he[c1], w.c_shape[c2], w.c_he[c2], pos12, CONVEX ? w.c_mat[c1].w : 0.0f, CONVEX ? w.c_mat[c2].w : 0.0f);
    w.p_npts[s] = 0; w.p_nsc[s] = 0;
    if (now_i != had_i) {
        pf ^= RP_PF_INTERSECTING;
        if (pair_wants_collision_events(w, c1, c2)) push_collision_event(w, c1, c2, now_i ? 1 : 0, RP_COLLISION_EVENT_SENSOR, cur_step(w));
    }
    w.p_pflags[s] = pf;
}

// The full narrow-phase update of one pair (pair_update.rs:173-613).
// parry DefaultQueryDispatcher::contact_manifold_convex_convex (pair_update.rs:323-330 reaches it through contact_manifolds) for two
// primitive SHAPES in the c_he layout (a collider, a part of a compound, a mesh triangle: tri1 / tri2), shape 2 at pos12 in shape 1's frame
template <bool CONVEX> __device__ __forceinline__ void dispatch_manifold(const DevWorld &w, int sh1, float4 he1, float bd1, const V3 *tri1, int sh2, float4 he2, float bd2, const V3 *tri2,
                                                                         Pose pos12, float prediction, LocalManifold &m) {
    bool convex_pair = false;
    if constexpr (CONVEX) { // cylinders, cones (rp_convex.h): the dispatcher's order — ball arms, half-space arms, pfm_pfm
        convex_pair = sh1 >= RP_SHAPE_CYLINDER || sh2 >= RP_SHAPE_CYLINDER;
        if (convex_pair) {
            SmShape a = sm_shape_of(w, sh1, he1, bd1), b = sm_shape_of(w, sh2, he2, bd2);
            if (tri1) { a.tri[0] = tri1[0]; a.tri[1] = tri1[1]; a.tri[2] = tri1[2]; }
            if (tri2) { b.tri[0] = tri2[0]; b.tri[1] = tri2[1]; b.tri[2] = tri2[2]; }
            if (sh2 == RP_SHAPE_BALL) manifold_sm_ball(pos12, a, he2.x, prediction, m, false);
            else if (sh1 == RP_SHAPE_BALL) manifold_sm_ball(pose_inv(pos12), b, he1.x, prediction, m, true);
            else if (sh1 == RP_SHAPE_HALFSPACE) manifold_halfspace_sm(pos12, v3(he1), b, prediction, m, false);
            else if (sh2 == RP_SHAPE_HALFSPACE) manifold_halfspace_sm(pose_inv(pos12), v3(he2), a, prediction, m, true);
            else manifold_pfm_pfm(pos12, a, b, prediction, m);
        }
    }
    if (convex_pair) {}
    else if (sh1 == RP_SHAPE_HALFSPACE || sh2 == RP_SHAPE_HALFSPACE) { // the ball arms come before (HalfSpace, pfm) | (pfm, HalfSpace) in the dispatcher
        if (sh1 == sh2) m.n = 0; // unsupported pair
        else if (sh1 == RP_SHAPE_HALFSPACE && sh2 == RP_SHAPE_BALL) manifold_halfspace_ball(pos12, v3(he1), he2.x, prediction, m, false);
        else if (sh2 == RP_SHAPE_HALFSPACE && sh1 == RP_SHAPE_BALL) manifold_halfspace_ball(pose_inv(pos12), v3(he2), he1.x, prediction, m, true);
        else if (sh1 == RP_SHAPE_HALFSPACE) manifold_halfspace_pfm(pos12, v3(he1), sh2, he2, prediction, m, false);
        else manifold_halfspace_pfm(pose_inv(pos12), v3(he2), sh1, he1, prediction, m, true);
    }
    else if (sh1 == RP_SHAPE_CUBOID && sh2 == RP_SHAPE_CUBOID) manifold_cuboid_cuboid(pos12, v3(he1), v3(he2), prediction, m);
    else if (sh1 == RP_SHAPE_BALL && sh2 == RP_SHAPE_BALL) manifold_ball_ball(pos12, he1.x, he2.x, prediction, m);
    else if (sh1 == RP_SHAPE_CAPSULE && sh2 == RP_SHAPE_CAPSULE) manifold_capsule_capsule(pos12, he1, he2, prediction, m);
    else if (sh1 == RP_SHAPE_CUBOID && sh2 == RP_SHAPE_CAPSULE) manifold_cuboid_capsule(pos12, pos12, v3(he1), he2, prediction, m, false);
    else if (sh1 == RP_SHAPE_CAPSULE && sh2 == RP_SHAPE_CUBOID) manifold_cuboid_capsule(pose_inv(pos12), pos12, v3(he2), he1, prediction, m, true);
    else if (sh1 == RP_SHAPE_CAPSULE && sh2 == RP_SHAPE_BALL) manifold_capsule_ball(pos12, he1, he2.x, prediction, m, false);
    else if (sh1 == RP_SHAPE_BALL && sh2 == RP_SHAPE_CAPSULE) manifold_capsule_ball(pose_inv(pos12), he2, he1.x, prediction, m, true);
    else if (sh1 == RP_SHAPE_CUBOID) manifold_cuboid_ball(pos12, v3(he1), he2.x, prediction, m, false);
    else manifold_cuboid_ball(pose_inv(pos12), v3(he2), he1.x, prediction, m, true);
}

// The pair-level tail of a full update (pair_update.rs:582-650, contacts.rs:300-364): solver-contact count of the pair's manifold 0,
// recycle state, hint, begin / end-touch transition (events, colour, journal, the colouring queue).  Shared by pair_full_update and
// the composite pairs' cluster path (rp_composite.h).  sh / he / border: the COLLIDERS' shapes.
__device__ __forceinline__ void pair_update_finish(DevWorld &w, int s, int c1, int c2, int rb1, int rb2, int csh1, float4 che1, int csh2, float4 che2, float cbd1, float cbd2,
                                                    Pose pc1, Pose pc2, Pose cpos12, int nsc, int had, bool no_contact, float restitution) {
    const float prediction = w.prm.prediction;
    w.p_nsc[s] = nsc;
    // recycle state — pair_update.rs:582-613
    float recycle = w.prm.recycle_distance;
    if (no_contact) { // ContactPair::clear: no manifold, no recycle state; skipped from now on (k_np_test) until the joint set changes
        w.p_pflags[s] = (w.p_pflags[s] & ~RP_PF_RECYCLE) | RP_PF_NO_CONTACT;
        w.p_misc[s] = make_float4(restitution, 0.0f, 0.0f, 0.0f);
    } else if (recycle > 0.0f) {
        float max_extent;
        if (w.p_pflags[s] & RP_PF_RECYCLE) max_extent = w.p_misc[s].y;
        else {
            float e1 = shape_origin_radius(csh1, che1, cbd1), e2 = shape_origin_radius(csh2, che2, cbd2);
            max_extent = rp_max(e1, e2);
        }
        float max_drift = nsc > 0 ? recycle : rp_min(recycle, prediction);
        w.p_misc[s] = make_float4(restitution, max_extent, max_drift, 0.0f);
        w.r_t[s] = f4(cpos12.t, 0.0f); w.r_r[s] = f4(cpos12.r); w.r_rot1[s] = f4(pc1.r); w.r_rot2[s] = f4(pc2.r);
        w.p_pflags[s] |= RP_PF_RECYCLE;
    } else {
        w.p_misc[s] = make_float4(restitution, 0.0f, 0.0f, 0.0f);
    }
    atomicAdd(&w.flags[FL_FULL_UPDATES], 1);
    // begin/end-touch transition — pair_update.rs:622-629, contacts.rs:300-364
    int has = nsc > 0;
    if (w.sleep_enabled) {
        // :636-650 hint refreshed from the final state; a pair that re-enters the selection dirties the layout
        if (has && pair_hint_cleared(w, s, make_int2(rb1, rb2))) w.flags[FL_LAYOUT_DIRTY] = 1;
        w.p_hint_seq[s] = cur_step(w);
        // wake rule (contacts.rs:333-351): a begin-touch wakes the sleeping side (its whole island, rp_sleep.hip)
        if (has && !had) {
            if (body_sleeping(w, rb1)) atomicMax(&w.b_wake_req[rb1], 1);
            if (body_sleeping(w, rb2)) atomicMax(&w.b_wake_req[rb2], 1);
        }
    }
    if (has != had) {
        w.flags[FL_LAYOUT_DIRTY] = 1;
        if (pair_wants_collision_events(w, c1, c2)) push_collision_event(w, c1, c2, has, 0, cur_step(w)); // contacts.rs:316-323
        if (!has) { // end touch: free the colour now (clear_pair_solver_color, mod.rs:157-172); the contact link is unlinked (contacts.rs:359)
            if (w.sleep_enabled) pi_journal(w, rb1, rb2, 2, c1, c2);
            int color = w.p_color[s];
            if (color < RP_COLOR_OVERFLOW) {
                int2 cb = w.p_colorb[s];
                unsigned bit = 1u << (color & 31);
                if (cb.x >= 0) atomicAnd(&w.b_cmask[4 * cb.x + (color >> 5)], ~bit);
                if (cb.y >= 0) atomicAnd(&w.b_cmask[4 * cb.y + (color >> 5)], ~bit);
            }
            w.p_color[s] = RP_COLOR_UNCOLORED; w.p_colorb[s] = make_int2(-1, -1);
        } else { // begin touch: queue for the sorted greedy colouring
            int t = atomicAdd(&w.flags[FL_TODO_COUNT], 1);
            // canonical greedy order: (min body, max body) as in contacts.rs:369-385; ties (several collider pairs between the
            // same two bodies) by the colliders' attachment ordinals — the reference uses its edge creation order there
            unsigned a = rb1 >= 0 ? (unsigned)rb1 : 0xfffffu, b = rb2 >= 0 ? (unsigned)rb2 : 0xfffffu;
            unsigned lo = a < b ? a : b, hi = a < b ? b : a;
            // (the tie: ordinal on the lower body id — a real body: < 4,096 colliders — then the ordinal on the higher one, which may be
            // "no body": parentless colliders count up to 2^20, b3d_large_world has a million.  Kept beside the key, in todo_tmp — scratch
            // of k_layout_rebuild, which runs after the colouring)
            const unsigned o1 = (unsigned)w.c_ord[c1], o2 = (unsigned)w.c_ord[c2];
            const unsigned tie = a < b ? (o1 << 20) | o2 : (o2 << 20) | o1;
            w.todo_slot[t] = s;
            w.todo_key[t] = ((unsigned long long)lo << 44) | ((unsigned long long)hi << 24);
            w.todo_tmp[t] = (int)tie;
        }
    }
}

// A composite pair on its PLAIN path (rp_composite.h): the ONE candidate sub-shape pair the pair's manifold belongs to this step.
struct SubSel {
    int sh1, sh2; float4 he1, he2; float bd1, bd2; V3 tri1[3], tri2[3];
    Pose wp1, wp2;            // world poses the manifold's points are local to (collider pose x part pose)
    Pose rel;                 // sub-shape 2 in sub-shape 1's frame
    bool has_pos1; Pose pos1; // side 1's part pose (carry_warmstart_data's subshape_pos1)
    bool fresh;               // the stored manifold belongs to another sub-shape pair (or the pair held clusters): start from an empty one
    bool none;                // no candidate at all: no manifold
    int prev_ncl;             // solver clusters of the previous step: the warm-start source when clustering stops applying
};
__device__ void composite_carry_to_plain(DevWorld &w, int s, const LocalManifold &m, const SubSel &sub, float4 *cimp, float4 *cwst); // rp_composite.h
template <bool CONVEX> __device__ __forceinline__ void pair_full_update(DevWorld &w, int s, int c1, int c2, Pose pc1, Pose pc2, Pose pos12, float *np_lds, const SubSel *sub = nullptr) {
    const float prediction = w.prm.prediction;
    int rb1 = w.c_parent[c1], rb2 = w.c_parent[c2];
    int sh1 = w.c_shape[c1], sh2 = w.c_shape[c2];
    float4 he1 = w.c_he[c1], he2 = w.c_he[c2];
    if (w.has_sensors && pair_is_sensor(w, c1, c2)) { sensor_pair_update<CONVEX>(w, s, c1, c2, pos12); return; }
    int had = w.p_nsc[s] > 0;
    const bool no_contact = joints_disable_contacts(w, rb1, rb2); // pair_update.rs:191-201: clear_filtered_pair

    LocalManifold m;
    m.bind(np_lds);
    m.n = w.p_npts[s];
    m.ln1 = v3(w.p_ln1[s]); m.ln2 = v3(w.p_ln2[s]);
    for (int k = 0; k < m.n; ++k) {
        float4 a = PT(w.pt_lp1d, k, s), b = PT(w.pt_lp2f, k, s);
        m.lp1[k] = v3(a); m.dist[k] = a.w; m.lp2[k] = v3(b); m.fid[k] = __float_as_uint(b.w); m.src[k] = k;
    }
    if (sub && sub->fresh) { m.n = 0; m.ln1 = v3(0, 0, 0); m.ln2 = v3(0, 0, 0); }
    int nold = m.n;
    // (a composite pair: the shapes that meet are the candidate sub-shapes, in the sub-shapes' relative pose; the collider-level values
    // above keep serving the recycle state)
    const int csh1 = sh1, csh2 = sh2; const float4 che1 = he1, che2 = he2; const Pose cpos12 = pos12;
    float bd1 = w.c_mat[c1].w, bd2 = w.c_mat[c2].w;
    const Pose wp1 = sub ? sub->wp1 : pc1, wp2 = sub ? sub->wp2 : pc2;
    if (sub) { sh1 = sub->sh1; sh2 = sub->sh2; he1 = sub->he1; he2 = sub->he2; bd1 = sub->bd1; bd2 = sub->bd2; pos12 = sub->rel; }
    // pair_update.rs:323-330 -> parry DefaultQueryDispatcher::contact_manifolds
    if (sub && sub->none) m.n = 0;
    else dispatch_manifold<CONVEX>(w, sh1, he1, bd1, sub ? sub->tri1 : nullptr, sh2, he2, bd2, sub ? sub->tri2 : nullptr, pos12, prediction, m);

    // carry ContactData (impulse, warm starts) to the new point order
    float4 cimp[RP_MAX_PTS], cwst[RP_MAX_PTS];
    const bool from_clusters = sub && sub->prev_ncl > 0; // clustering stopped applying: the warm-start data comes back from the clusters, by position (pair_update.rs:385-396)
    if (from_clusters) composite_carry_to_plain(w, s, m, *sub, cimp, cwst);
    for (int k = 0; k < nold; ++k) { m.b_set4(4 * k, PT(w.pt_imp, k, s)); m.b_set4(32 + 4 * k, PT(w.pt_wst, k, s)); }
    for (int k = 0; k < m.n; ++k) {
        int j = m.src[k];
        float4 im = j >= 0 ? m.b_get4(4 * j) : make_float4(0, 0, 0, 0);
        float4 ws = j >= 0 ? m.b_get4(32 + 4 * j) : make_float4(0, 0, 0, 0);
        if (from_clusters) { im = cimp[k]; ws = cwst[k]; }
        PT(w.pt_imp, k, s) = im; PT(w.pt_wst, k, s) = ws;
        PT(w.pt_lp1d, k, s) = f4(m.lp1[k], m.dist[k]);
        PT(w.pt_lp2f, k, s) = f4(m.lp2[k], __uint_as_float(m.fid[k]));
    }
    w.p_npts[s] = m.n;
    if (!no_contact) { w.p_ln1[s] = f4(m.ln1, 0.0f); w.p_ln2[s] = f4(m.ln2, 0.0f); } // (a filtered pair never reaches the generator in the reference: its cached normal stays)

    float4 mat1 = w.c_mat[c1], mat2 = w.c_mat[c2];
    int2 ru1 = w.c_rules[c1], ru2 = w.c_rules[c2];
    float friction = combine_coeff(mat1.x, mat2.x, ru1.x, ru2.x);
    float restitution = combine_coeff(mat1.y, mat2.y, ru1.y, ru2.y);
    int rel_dom = effective_dominance(w, rb1) - effective_dominance(w, rb2);
    V3 normal = qrot(wp1.r, m.ln1);
    w.p_normal[s] = f4(normal, friction);
    w.p_reldom[s] = rel_dom;

    int nsc = 0;
    if (no_contact) { m.n = 0; w.p_npts[s] = 0; }
    if (m.n > 0) {
        int sel[4] = {0, 1, 2, 3};
        int nsel = m.n < 4 ? m.n : 4;
        reduce_manifold(m, sel, nsel, prediction);
        if (nsel > 1) { // pair_update.rs:430-457
            V3 b0, b1; orthonormal_basis(m.ln1, b0, b1);
            float k0[4], k1[4]; int ks[4];
            for (int i = 0; i < nsel; ++i) { V3 lp = m.lp1[sel[i]]; k0[i] = dot(lp, b0); k1[i] = dot(lp, b1); ks[i] = sel[i]; }
            for (int i = 1; i < nsel; ++i) {
                float a0 = k0[i], a1 = k1[i]; int as = ks[i]; int j = i;
                while (j > 0 && (k0[j - 1] > a0 || (k0[j - 1] == a0 && k1[j - 1] > a1))) { k0[j] = k0[j - 1]; k1[j] = k1[j - 1]; ks[j] = ks[j - 1]; j--; }
                k0[j] = a0; k1[j] = a1; ks[j] = as;
            }
            for (int i = 0; i < nsel; ++i) sel[i] = ks[i];
        }
        bool has1 = rb1 >= 0 && rel_dom <= 0, has2 = rb2 >= 0 && rel_dom >= 0;
        Pose com1, com2;
        com1.r = q4(0, 0, 0, 1); com1.t = v3(0, 0, 0); com2 = com1;
        V3 lv1 = v3(0, 0, 0), av1 = lv1, wc1 = lv1, lv2 = lv1, av2 = lv1, wc2 = lv1;
        if (rb1 >= 0) { lv1 = v3(w.b_linvel[rb1]); av1 = v3(w.b_angvel[rb1]); wc1 = v3(w.b_wcom[rb1]); }
        if (rb2 >= 0) { lv2 = v3(w.b_linvel[rb2]); av2 = v3(w.b_angvel[rb2]); wc2 = v3(w.b_wcom[rb2]); }
        if (has1) { Pose bp; bp.r = q4(w.b_rot[rb1]); bp.t = v3(w.b_pos[rb1]); com1.r = bp.r; com1.t = pose_tp(bp, v3(w.b_lcom_invm[rb1])); }
        if (has2) { Pose bp; bp.r = q4(w.b_rot[rb2]); bp.t = v3(w.b_pos[rb2]); com2.r = bp.r; com2.t = pose_tp(bp, v3(w.b_lcom_invm[rb2])); }
        for (int q = 0; q < nsel; ++q) { // pair_update.rs:459-498 + :536-577
            int cid = sel[q];
            float eff_dist = m.dist[cid];
            V3 wpt1 = pose_tp(wp1, m.lp1[cid]);
            V3 wpt2 = pose_tp(wp2, m.lp2[cid]);
            bool keep = eff_dist < prediction;
            if (!keep) {
                V3 vel1 = rb1 >= 0 ? lv1 + cross(av1, wpt1 - wc1) : v3(0, 0, 0);
                V3 vel2 = rb2 >= 0 ? lv2 + cross(av2, wpt2 - wc2) : v3(0, 0, 0);
                keep = eff_dist + dot(vel2 - vel1, normal) * w.prm.p.dt < prediction;
            }
            if (!keep) continue;
            float shift = dot(wpt2 - wpt1, normal) - eff_dist;
            V3 p1 = wpt1 + normal * shift;
            V3 point = (p1 + wpt2) * 0.5f;
            PT(w.pt_dp1, cid, s) = f4(has1 ? point - com1.t : point, 0.0f);
            PT(w.pt_dp2, cid, s) = f4(has2 ? point - com2.t : point, 0.0f);
            V3 a1 = has1 ? pose_itp(com1, p1) : p1;
            V3 a2 = has2 ? pose_itp(com2, wpt2) : wpt2;
            PT(w.sc_a1, nsc, s) = f4(a1, eff_dist);
            PT(w.sc_a2, nsc, s) = f4(a2, __int_as_float(cid));
            nsc++;
        }
    }
    pair_update_finish(w, s, c1, c2, rb1, rb2, csh1, che1, csh2, che2, mat1.w, mat2.w, pc1, pc2, cpos12, nsc, had, no_contact, restitution);
}

#include "rp_composite.h"

// The narrow phase runs as two kernels: k_np_test (one thread per pair slot, a few dozen flops: who is awake, the
// contact-recycling test of pair_update.rs:111-171) queues the pairs that need contact determination, k_np_update runs
// the full update (parry manifolds, reduction, solver contacts: ~2.7 KB of scratch per lane) over that queue only.  On a
// settled scene the queue is empty and the heavy kernel exits at once instead of taxing every pair with its launch
// footprint (29 us -> a few us per step on b3d_many_pyramids).
__global__ void k_np_test(DevWorld w) {
    if (blockIdx.x == 0) bp_close_incremental(w); // (the broad-phase pass in front of this kernel, if it was an incremental one)
    if (collision_done(w)) return; // (rp_world.h "lean step graphs")
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < top; s += stride) {
        int c1 = w.p_c1[s];
        if (c1 < 0) continue;
        if (w.has_composite && pair_is_aux(w, s)) continue; // (a cluster of a composite pair: its parent slot is the pair)
        int c2 = w.p_c2[s];
        if (w.n_nc && (w.p_pflags[s] & RP_PF_NO_CONTACT)) continue; // filtered by a contact-disabling joint (cleared below)
        Pose pc1, pc2;
        pc1.r = q4(w.c_rot[c1]); pc1.t = v3(w.c_pos[c1]);
        pc2.r = q4(w.c_rot[c2]); pc2.t = v3(w.c_pos[c2]);
        Pose pos12 = pose_inv_mul(pc1, pc2);
        if (w.sleep_enabled) {
            int2 rb = w.p_rb[s];
            // pair_update.rs:98-106: neither body awake (fixed or asleep) -> skipped
            if (!body_active(w, rb.x) && !body_active(w, rb.y)) continue;
            if (pair_recycle_ok(w, s, pc1, pc2, pos12)) {
                // :141-161 a count-cleared hint (the pair slept) is recomputed: the pair re-enters the selection
                if (pair_hint_cleared(w, s, rb)) { w.p_hint_seq[s] = cur_step(w); if (w.p_nsc[s] != 0) w.flags[FL_LAYOUT_DIRTY] = 1; }
                continue;
            }
        } else if (pair_recycle_ok(w, s, pc1, pc2, pos12)) continue;
        w.np_list[atomicAdd(&w.flags[FL_NP_COUNT], 1)] = s;
    }
}
template <bool CONVEX> __global__ void __launch_bounds__(NP_THREADS) k_np_update(DevWorld w) {
    __shared__ __align__(16) float np_lds[NP_THREADS * NP_LDS_DWORDS];
    if (collision_done(w)) return; // (rp_world.h "lean step graphs")
    int count = w.flags[FL_NP_COUNT];
    if (count > w.pool_cap) count = w.pool_cap;
    int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        int s = w.np_list[i];
        int c1 = w.p_c1[s], c2 = w.p_c2[s];
        if (w.has_composite && (shape_is_composite(w.c_shape[c1]) || shape_is_composite(w.c_shape[c2]))) continue; // k_np_composite takes it
        Pose pc1, pc2;
        pc1.r = q4(w.c_rot[c1]); pc1.t = v3(w.c_pos[c1]);
        pc2.r = q4(w.c_rot[c2]); pc2.t = v3(w.c_pos[c2]);
        Pose pos12 = pose_inv_mul(pc1, pc2);
        pair_full_update<CONVEX>(w, s, c1, c2, pc1, pc2, pos12, np_lds);
    }
}

// Fast graph, worlds with sensors: k_fast_front leaves the sensor pairs alone (they hold no recycle state), this pass re-tests them
// from the collider poses k_fast_front just refreshed — what the narrow phase of a full step would have done for them.  Runs only
// once no later kernel of the graph can still abort the step.
template <bool CONVEX> __global__ void k_sensor_pass(DevWorld w) {
    if (w.flags[FL_FAST_ABORT]) return;
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < top; s += stride) {
        int c1 = w.p_c1[s];
        if (c1 < 0) continue;
        int c2 = w.p_c2[s];
        if (!pair_is_sensor(w, c1, c2)) continue;
        if (w.sleep_enabled) { int2 rb = w.p_rb[s]; if (!body_active(w, rb.x) && !body_active(w, rb.y)) continue; } // pair_update.rs:98-106
        Pose pc1, pc2;
        pc1.r = q4(w.c_rot[c1]); pc1.t = v3(w.c_pos[c1]);
        pc2.r = q4(w.c_rot[c2]); pc2.t = v3(w.c_pos[c2]);
        sensor_pair_update<CONVEX>(w, s, c1, c2, pose_inv_mul(pc1, pc2));
    }
}
// The fused fast step (one kernel that validates and solves, rp_islands.hip) cannot raise a sensor's events itself: this launch in front
// of it tests every sensor pair READ-ONLY from the body poses and sends the step to the full graph (FL_FAST_ABORT: k_island_solve then
// only retires the launch) when an intersection would start or stop — the replay raises the event.  While nothing changes, a world
// with sensors keeps the fused step.
template <bool CONVEX> __global__ void k_sensor_check(DevWorld w) {
    if (w.flags[FL_FAST_ABORT]) return;
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    int stride = gridDim.x * blockDim.x;
    bool changed = false;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < top; s += stride) {
        int c1 = w.p_c1[s];
        if (c1 < 0) continue;
        int c2 = w.p_c2[s];
        if (!pair_is_sensor(w, c1, c2)) continue;
        const int rb1 = w.c_parent[c1], rb2 = w.c_parent[c2];
        if (w.sleep_enabled && !body_active(w, rb1) && !body_active(w, rb2)) continue; // pair_update.rs:98-106
        const Pose pc1 = collider_world_pose_of(w, c1, rb1), pc2 = collider_world_pose_of(w, c2, rb2);
        const bool had_i = (w.p_pflags[s] & RP_PF_INTERSECTING) != 0;
        const bool now_i = (rb1 == rb2 && rb1 >= 0) ? false : shapes_intersect<CONVEX>(w, w.c_shape[c1], w.c_he[c1], w.c_shape[c2], w.c_he[c2], pose_inv_mul(pc1, pc2), CONVEX ? w.c_mat[c1].w : 0.0f, CONVEX ? w.c_mat[c2].w : 0.0f);
        if (now_i != had_i) changed = true;
    }
    if (changed) w.flags[FL_FAST_ABORT] = 1;
}
void rp_launch_sensor_check(const DevWorld &w, hipStream_t st) {
    if (!w.has_sensors || w.n_colliders == 0) return;
    int blocks = (w.pool_cap + 255) / 256; if (blocks > 1024) blocks = 1024;
    if (w.has_convex) hipLaunchKernelGGL(k_sensor_check<true>, dim3(blocks), dim3(256), 0, st, w);
    else hipLaunchKernelGGL(k_sensor_check<false>, dim3(blocks), dim3(256), 0, st, w);
}
void rp_launch_sensor_fast(const DevWorld &w, hipStream_t st) {
    if (!w.has_sensors || w.n_colliders == 0) return;
    int blocks = (w.pool_cap + 255) / 256; if (blocks > 1024) blocks = 1024;
    if (w.has_convex) hipLaunchKernelGGL(k_sensor_pass<true>, dim3(blocks), dim3(256), 0, st, w);
    else hipLaunchKernelGGL(k_sensor_pass<false>, dim3(blocks), dim3(256), 0, st, w);
}

// the set of contact-disabling joints changed: every filtered pair is evaluated again by the next narrow phase
__global__ void k_clear_no_contact(DevWorld w) {
    int top = w.flags[FL_POOL_TOP];
    if (top > w.pool_cap) top = w.pool_cap;
    int stride = gridDim.x * blockDim.x;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < top; s += stride) if (w.p_c1[s] >= 0) w.p_pflags[s] &= ~RP_PF_NO_CONTACT;
}
void rp_launch_clear_no_contact(const DevWorld &w, hipStream_t st) {
    int blocks = (w.pool_cap + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_clear_no_contact, dim3(blocks), dim3(256), 0, st, w);
}

// ------------------------------------------------------------------------------------------
// Deferred greedy colouring (apply_deferred_solver_coloring, contacts.rs:369-385 +
// assign_pair_solver_color, narrow_phase/mod.rs:90-154).
//
// The reference colours this step's begin-touch pairs one after the other in key order; a pair's colour depends only on the
// masks of its (at most two) dynamic bodies at that moment, i.e. on the pairs with SMALLER keys at those bodies.  That is a
// dependency DAG with at most two predecessors per pair (the previous pair, by key, at each body), and any schedule that
// respects it yields the serial result.  ONE workgroup runs it as a wavefront over that DAG:
//   passes 1-4  per-body lists of the queued pairs (count -> reserve -> fill -> rank by key: the same count/rank scheme as the
//               toucher lists of rp_flow.hip), each pair learns its successor at either body and its number of predecessors;
//   rounds      the frontier (pairs without an uncoloured predecessor — never two at one body) is coloured in parallel; each
//               coloured pair releases its successors, which form the next frontier.
// Work is O(pairs + sum of degree^2), a round costs a few dependent L2 round trips, and the number of rounds is the depth of the
// DAG — the first step of b3d_large_pyramid (59,900 pairs) takes a few ms where the bidding scheme of round 1, which rescanned
// every pending pair in every round, took 525 ms.
RP_DEV unsigned ld_u32(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define RP_COLOR_CHAIN_HOPS 8
RP_DEV int ld_i32a(int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(1024) k_color_pairs(DevWorld w) {
    // first launch behind the collision stage, full graphs only: a dead lean step's resume (rp_world.h "lean step graphs") is past the
    // kernels that skip on the marker
    if (threadIdx.x == 0 && w.flags[FL_FAST_ABORT] == 2) w.flags[FL_FAST_ABORT] = 0;
    const int T = w.flags[FL_TODO_COUNT];
    if (T == 0) return;
    __shared__ int cursor, n_cur, n_next;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) { cursor = 0; n_cur = 0; n_next = 0; }
#ifdef RP_COLOR_PROFILE // tools/color_profile.py: time stamps (10 ns ticks) after every pass, rounds and frontier sizes of the largest launch
#define COL_STAMP(k) do { if (tid == 0 && T > 1000) w.dbg[40 + (k)] = (long long)wall_clock64(); } while (0)
    int prof_rounds = 0; long long prof_items = 0;
#else
#define COL_STAMP(k) do { } while (0)
#endif
    COL_STAMP(0);
    __syncthreads();
    // pass 1: dynamic sides, per-body counts; the first pair to touch a body reserves its list in pass 2
    for (int t = tid; t < T; t += nt) {
        int s = w.todo_slot[t];
        int b1 = w.c_parent[w.p_c1[s]], b2 = w.c_parent[w.p_c2[s]];
        if (!body_dynamic(w, b1)) b1 = -1;
        if (!body_dynamic(w, b2)) b2 = -1;
        int first = 0;
        if (b1 >= 0 && atomicAdd(&w.col_cnt[b1], 1) == 0) first |= 1;
        if (b2 >= 0 && atomicAdd(&w.col_cnt[b2], 1) == 0) first |= 2;
        w.col_rec[t] = make_int4(b1, b2, first, s);
    }
    __threadfence(); __syncthreads();
    COL_STAMP(1);
    // pass 2: reserve
    for (int t = tid; t < T; t += nt) {
        int4 r = w.col_rec[t];
        if (r.z & 1) w.col_begin[r.x] = atomicAdd(&cursor, ld_i32a(&w.col_cnt[r.x]));
        if (r.z & 2) w.col_begin[r.y] = atomicAdd(&cursor, ld_i32a(&w.col_cnt[r.y]));
    }
    __threadfence(); __syncthreads();
    COL_STAMP(2);
    // pass 3: fill (any order)
    for (int t = tid; t < T; t += nt) {
        int4 r = w.col_rec[t];
        if (r.x >= 0) w.col_list[ld_i32a(&w.col_begin[r.x]) + atomicAdd(&w.col_fill[r.x], 1)] = t;
        if (r.y >= 0) w.col_list[ld_i32a(&w.col_begin[r.y]) + atomicAdd(&w.col_fill[r.y], 1)] = t;
    }
    __threadfence(); __syncthreads();
    COL_STAMP(3);
    // pass 4: rank by key inside each body's list -> sorted lists
    for (int t = tid; t < T; t += nt) {
        int4 r = w.col_rec[t];
        const unsigned long long key = w.todo_key[t];
        const unsigned tie = (unsigned)w.todo_tmp[t];
        int2 rk = make_int2(-1, -1);
        for (int side = 0; side < 2; ++side) {
            int b = side ? r.y : r.x;
            if (b < 0) continue;
            int beg = ld_i32a(&w.col_begin[b]), n = ld_i32a(&w.col_cnt[b]), q = 0;
            for (int k = 0; k < n; ++k) { const int u = ld_i32a(&w.col_list[beg + k]); const unsigned long long ku = w.todo_key[u]; q += ku < key || (ku == key && (unsigned)w.todo_tmp[u] < tie); }
            w.col_sorted[beg + q] = t;
            if (side) rk.y = q; else rk.x = q;
        }
        w.col_rank[t] = rk;
        w.col_deps[t] = (rk.x > 0) + (rk.y > 0);
    }
    __threadfence(); __syncthreads();
    COL_STAMP(4);
    // pass 5: successors, first frontier; the per-body counters go back to rest
    for (int t = tid; t < T; t += nt) {
        int4 r = w.col_rec[t];
        int2 rk = w.col_rank[t];
        int s1 = -1, s2 = -1;
        if (r.x >= 0 && rk.x + 1 < ld_i32a(&w.col_cnt[r.x])) s1 = ld_i32a(&w.col_sorted[ld_i32a(&w.col_begin[r.x]) + rk.x + 1]);
        if (r.y >= 0 && rk.y + 1 < ld_i32a(&w.col_cnt[r.y])) s2 = ld_i32a(&w.col_sorted[ld_i32a(&w.col_begin[r.y]) + rk.y + 1]);
        w.col_succ[t] = make_int2(s1, s2);
        if (rk.x <= 0 && rk.y <= 0) w.col_q[atomicAdd(&n_cur, 1)] = t;
    }
    __threadfence(); __syncthreads();
    for (int t = tid; t < T; t += nt) {
        int4 r = w.col_rec[t];
        if (r.x >= 0) { w.col_cnt[r.x] = 0; w.col_fill[r.x] = 0; }
        if (r.y >= 0) { w.col_cnt[r.y] = 0; w.col_fill[r.y] = 0; }
    }
    // rounds
    COL_STAMP(5);
    int *qc = w.col_q, *qn = w.col_q + w.pool_cap;
    for (;;) {
        const int n = n_cur;
        __syncthreads();
        if (n == 0) break;
#ifdef RP_COLOR_PROFILE
        prof_rounds++; prof_items += n;
#endif
        for (int f = tid; f < n; f += nt) {
            // a thread follows its pair's chain: the first successor it releases is coloured by the same thread at once (most of the
            // DAG is chains — body k's pairs one after the other), only further released successors wait in the queue for the next round
            int t = ld_i32a(&qc[f]);
            int4 r = w.col_rec[t];
            int2 su = w.col_succ[t];
            for (int hops = 0; t >= 0; ++hops) {
                // (bounded: a thread that walked a long chain to its end would hold the round open while every other ready pair waits)
                if (hops == RP_COLOR_CHAIN_HOPS) { qn[atomicAdd(&n_next, 1)] = t; break; }
                // the records of both successors are fetched now, behind the mask / atomic round trips of this pair: a hop is three
                // dependent L2 round trips instead of four
                int4 rx = r, ry = r; int2 sx = su, sy = su;
                if (su.x >= 0) { rx = w.col_rec[su.x]; sx = w.col_succ[su.x]; }
                if (su.y >= 0) { ry = w.col_rec[su.y]; sy = w.col_succ[su.y]; }
                const int b1 = r.x, b2 = r.y, s = r.w;
                const bool d1 = b1 >= 0, d2 = b2 >= 0;
                int color = 128;
                unsigned m[4] = {0, 0, 0, 0};
                if (d1) for (int q = 0; q < 4; ++q) m[q] |= ld_u32(&w.b_cmask[4 * b1 + q]);
                if (d2) for (int q = 0; q < 4; ++q) m[q] |= ld_u32(&w.b_cmask[4 * b2 + q]);
                if (d1 && d2) {
                    for (int c = 0; c < RP_DYNAMIC_COLOR_COUNT; ++c) if (!((m[c >> 5] >> (c & 31)) & 1u)) { color = c; break; }
                } else if (d1 || d2) {
                    for (int c = 127; c >= 0; --c) if (!((m[c >> 5] >> (c & 31)) & 1u)) { color = c; break; }
                }
                if (color >= 128) { w.p_color[s] = RP_COLOR_OVERFLOW; w.p_colorb[s] = make_int2(-1, -1); }
                else {
                    unsigned bit = 1u << (color & 31);
                    if (d1) atomicOr(&w.b_cmask[4 * b1 + (color >> 5)], bit);
                    if (d2) atomicOr(&w.b_cmask[4 * b2 + (color >> 5)], bit);
                    w.p_color[s] = color;
                    w.p_colorb[s] = (d1 && d2) ? make_int2(b1, b2) : make_int2(d1 ? b1 : b2, -1);
                }
                // the mask bits have reached L2 before a successor can be released: this kernel is ONE workgroup (one XCD's L2) and the
                // masks are only ever touched by L2 atomics / L1-bypassing loads, so draining this wave's memory operations orders them —
                // no agent-scope release (L2 write-back) per hop
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // (when the same pair follows at both bodies it holds two predecessors' worth of this pair: the second decrement releases it)
                int next = -1; bool from_y = false;
                if (su.x >= 0 && atomicSub(&w.col_deps[su.x], 1) == 1) next = su.x;
                if (su.y >= 0 && atomicSub(&w.col_deps[su.y], 1) == 1) { if (next < 0) { next = su.y; from_y = true; } else qn[atomicAdd(&n_next, 1)] = su.y; }
                t = next;
                if (from_y) { r = ry; su = sy; } else { r = rx; su = sx; }
                t = next;
            }
        }
        __threadfence(); __syncthreads();
        if (tid == 0) { n_cur = n_next; n_next = 0; }
        int *tmp = qc; qc = qn; qn = tmp;
        __syncthreads();
    }
    COL_STAMP(6);
#ifdef RP_COLOR_PROFILE
    if (tid == 0 && T > 1000) { w.dbg[48] = prof_rounds; w.dbg[49] = prof_items; w.dbg[50] = T; }
#endif
}

// ------------------------------------------------------------------------------------------
// Solver contact graph buckets (maintain_solver_contact_graph, solver_graph.rs:129-361) and the
// stage order of init.rs:163-254: colours with >= 32 four-lane chunks ascending, then the smaller
// colours ascending; the overflow colour is always swept last (serially).  The order is decided
// from ALL active manifolds per colour (the reference's bucket sizes); positions in the global
// constraint planes are only handed to manifolds that are not solved by the island kernel.
void rp_launch_islands_build(const DevWorld &w, hipStream_t st);
void rp_launch_joint_coloring(const DevWorld &w, hipStream_t st);
void rp_launch_wake(const DevWorld &w, hipStream_t st, int phase);
void rp_launch_sleep(const DevWorld &w, hipStream_t st);

// `part`: 0 = contact determination (NarrowPhase::compute_contacts: test, update, deferred colouring, begin-touch wake-ups),
// 1 = island construction in the reference's stage accounting (sleep decision, joint colouring, solver contact graph
// buckets, contact islands), -1 = both (the step graphs), 2 = test + update alone (the lean step graph: rp_world.h).
// composite pairs of this step's queue (worlds with a compound / mesh collider only): one thread per pair, cm_ws_threads threads
static void rp_launch_np_composite(const DevWorld &w, hipStream_t st) {
    if (!w.has_composite || w.cm_ws_threads <= 0) return;
    hipLaunchKernelGGL(k_np_composite<true>, dim3(w.cm_ws_threads / NP_THREADS), dim3(NP_THREADS), 0, st, w);
    hipLaunchKernelGGL(k_cm_finish, dim3(1), dim3(256), 0, st, w);
}
void rp_launch_narrowphase_part(const DevWorld &w, hipStream_t st, int part) {
    int blocks = (w.pool_cap + 255) / 256; if (blocks > 2048) blocks = 2048;
    if (part == 2) {
        if (w.n_colliders == 0) return;
        hipLaunchKernelGGL(k_np_test, dim3(blocks), dim3(256), 0, st, w);
        int nb = (w.pool_cap + NP_THREADS - 1) / NP_THREADS;
        if (w.has_convex) hipLaunchKernelGGL(k_np_update<true>, dim3(nb < 512 ? nb : 512), dim3(NP_THREADS), 0, st, w);
        else hipLaunchKernelGGL(k_np_update<false>, dim3(nb < 512 ? nb : 512), dim3(NP_THREADS), 0, st, w);
        rp_launch_np_composite(w, st);
        return;
    }
    if (part != 1) {
        if (w.n_colliders > 0) {
            hipLaunchKernelGGL(k_np_test, dim3(blocks), dim3(256), 0, st, w);
            // at most one workgroup per CU: with 128 VGPRs and 2.7 KB of scratch per lane a second round of workgroups costs ~10 us
            // even when the queue is empty (measured: 341 -> 256 workgroups = 140 -> 130 us per full step on b3d_many_pyramids)
            // (sizing this grid from the recent queue lengths was measured: 5-10 us per full step, paid for with a ~10 ms re-capture of
            // the step graphs whenever the size changed — not kept)
            { int nb = (w.pool_cap + NP_THREADS - 1) / NP_THREADS;
              if (w.has_convex) hipLaunchKernelGGL(k_np_update<true>, dim3(nb < 512 ? nb : 512), dim3(NP_THREADS), 0, st, w); // (worlds with a cylinder / cone: GJK / EPA compiled in, a polytope per lane in scratch)
              else hipLaunchKernelGGL(k_np_update<false>, dim3(nb < 512 ? nb : 512), dim3(NP_THREADS), 0, st, w); } // 2 workgroups per CU (70 KB of LDS each)
            rp_launch_np_composite(w, st); // composite pairs (rp_composite.h)
            hipLaunchKernelGGL(k_color_pairs, dim3(1), dim3(1024), 0, st, w);
        }
        rp_launch_wake(w, st, 1); // begin-touch wake-ups (contacts.rs:333-351)
    }
    if (part != 0) {
        rp_launch_sleep(w, st);   // sleep timers + the whole-island sleep decision (solve.rs:196-300, manager.rs:335-388); collider-less bodies too
        // (a world without a single collider still has a solver layout: its bodies — under forces, on joints — sit on the global path;
        // returning here left such worlds unsolved: test_pipeline_unit.py, the reference's collider-less pipeline tests)
        rp_launch_joint_coloring(w, st); // joints avoid this step's contact colours (init_joints, joints.rs:25-329)
        rp_launch_islands_build(w, st); // colour buckets, contact islands, stage layout, constraint positions: one launch (rp_islands.hip)
    }
}
void rp_launch_narrowphase(const DevWorld &w, hipStream_t st) { rp_launch_narrowphase_part(w, st, -1); }

// ---- continuous collision detection (dynamics/ccd): kernel and launcher ----
#include "rp_ccd.h"
