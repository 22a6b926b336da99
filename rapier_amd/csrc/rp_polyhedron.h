// rp_polyhedron.h — convex polyhedra on the host side of the library (included by rp_api.hip only).
//
// ColliderBuilder::convex_mesh(points, indices) / convex_hull(points) (/root/reference/src/geometry/collider.rs:1039, :1070) build
// parry3d's ConvexPolyhedron (from_convex_mesh: triangles with equal normals merge into polygonal faces with vertex loops and edges),
// MassProperties::from_convex_polyhedron (signed tetrahedra from the centre of mass, Tonon's closed-form tensor) and
// point_cloud_bounding_sphere.  parry3d is not under /root/reference; this is a canonical form of our own, the same polyhedron
// whatever triangulation of its faces comes in:
//   vertices     the points the triangles use, in index order;
//   faces        maximal sets of edge-adjacent triangles whose unit normals agree (dot > 1 - 1e-5), as vertex loops, counter-clockwise
//                seen from outside, starting at the loop's smallest vertex; faces sorted by their loops; normals by Newell's sums;
//   edges        the faces' boundary edges sorted by (smaller vertex, larger vertex);
//   feature ids  vertex v -> v, edge e -> 0x4000 | e, face f -> 0x8000 | f;
//   mass         parry's formulas over the fan triangulation of the canonical loops.
// A collider stores the polyhedron RECENTRED on the centre of its local AABB and carries that offset in its pose (pos_wrt_parent *
// translation(centre)), so the broad phase, the recycle extents and the CCD pre-filter see a shape whose local box is symmetric about
// the collider origin, like every other one; the AABB is that box transformed (a superset of ConvexPolyhedron::aabb's point-cloud
// box: more near-miss pairs, the same contacts).  convex_hull = an incremental hull in double precision (first tetrahedron from
// extreme points, then every point in index order: faces that see it die, the horizon is re-faced).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#define RP_POLY_MAX_VERTS 256

struct HostPolyhedron {
    std::vector<float> pts;                  // recentred, xyz per vertex
    std::vector<float> fnormal;              // xyz per face
    std::vector<int> ffirst, fcount, loop_v, loop_e;
    int ne = 0;
    float centre[3] = {0, 0, 0}, half[3] = {0, 0, 0}, origin_radius = 0.0f;
    float sphere_centre[3] = {0, 0, 0}, sphere_radius = 0.0f; // ORIGINAL frame
    float volume = 0.0f, com[3] = {0, 0, 0}, inertia[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; // unit density, ORIGINAL frame, about the com
    int nv() const { return (int)pts.size() / 3; }
    int nf() const { return (int)ffirst.size(); }
};

namespace rp_poly {
struct P3 { float x, y, z; };
static inline P3 mk(float x, float y, float z) { P3 r = {x, y, z}; return r; }
static inline P3 sub(P3 a, P3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline P3 add(P3 a, P3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline P3 mul(P3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline float dot(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline P3 cross(P3 a, P3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float len(P3 a) { return std::sqrt(dot(a, a)); }
static inline float tet_signed_volume(P3 p1, P3 p2, P3 p3, P3 p4) { return dot(sub(p2, p1), cross(sub(p3, p1), sub(p4, p1))) / 6.0f; }
// tetrahedron_unit_inertia_tensor_wrt_point (Tonon 2004)
static inline void tet_unit_inertia(P3 pt, P3 p1, P3 p2, P3 p3, P3 p4, float out[3][3]) {
    P3 q1 = sub(p1, pt), q2 = sub(p2, pt), q3 = sub(p3, pt), q4 = sub(p4, pt);
    float x1 = q1.x, y1 = q1.y, z1 = q1.z, x2 = q2.x, y2 = q2.y, z2 = q2.z, x3 = q3.x, y3 = q3.y, z3 = q3.z, x4 = q4.x, y4 = q4.y, z4 = q4.z;
    float dx = x1 * x1 + x1 * x2 + x2 * x2 + x1 * x3 + x2 * x3 + x3 * x3 + x1 * x4 + x2 * x4 + x3 * x4 + x4 * x4;
    float dy = y1 * y1 + y1 * y2 + y2 * y2 + y1 * y3 + y2 * y3 + y3 * y3 + y1 * y4 + y2 * y4 + y3 * y4 + y4 * y4;
    float dz = z1 * z1 + z1 * z2 + z2 * z2 + z1 * z3 + z2 * z3 + z3 * z3 + z1 * z4 + z2 * z4 + z3 * z4 + z4 * z4;
    float a0 = (dy + dz) * 0.1f, b0 = (dz + dx) * 0.1f, c0 = (dx + dy) * 0.1f;
    float a1 = (y1 * z1 * 2.0f + y2 * z1 + y3 * z1 + y4 * z1 + y1 * z2 + y2 * z2 * 2.0f + y3 * z2 + y4 * z2 + y1 * z3 + y2 * z3 + y3 * z3 * 2.0f + y4 * z3 + y1 * z4 + y2 * z4 + y3 * z4 + y4 * z4 * 2.0f) * 0.05f;
    float b1 = (x1 * z1 * 2.0f + x2 * z1 + x3 * z1 + x4 * z1 + x1 * z2 + x2 * z2 * 2.0f + x3 * z2 + x4 * z2 + x1 * z3 + x2 * z3 + x3 * z3 * 2.0f + x4 * z3 + x1 * z4 + x2 * z4 + x3 * z4 + x4 * z4 * 2.0f) * 0.05f;
    float c1 = (x1 * y1 * 2.0f + x2 * y1 + x3 * y1 + x4 * y1 + x1 * y2 + x2 * y2 * 2.0f + x3 * y2 + x4 * y2 + x1 * y3 + x2 * y3 + x3 * y3 * 2.0f + x4 * y3 + x1 * y4 + x2 * y4 + x3 * y4 + x4 * y4 * 2.0f) * 0.05f;
    out[0][0] = a0; out[0][1] = -c1; out[0][2] = -b1;
    out[1][0] = -c1; out[1][1] = b0; out[1][2] = -a1;
    out[2][0] = -b1; out[2][1] = -a1; out[2][2] = c0;
}
static inline int uf_find(std::vector<int> &uf, int x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; }

// the canonical polyhedron of a closed, outward-wound convex triangle mesh; false = not one this construction can take
static bool build(HostPolyhedron &P, int n_points, const float *xyz, int n_tris, const uint32_t *tris) {
    if (n_points < 4 || n_tris < 4) return false;
    std::vector<int> remap(n_points, -1);
    for (int t = 0; t < 3 * n_tris; ++t) { if (tris[t] >= (uint32_t)n_points) return false; remap[tris[t]] = 0; }
    int nv = 0;
    for (int i = 0; i < n_points; ++i) if (remap[i] == 0) remap[i] = nv++;
    if (nv < 4 || nv > RP_POLY_MAX_VERTS) return false;
    std::vector<P3> pts(nv);
    for (int i = 0; i < n_points; ++i) if (remap[i] >= 0) pts[remap[i]] = mk(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    std::vector<int> tv(3 * n_tris);
    for (int t = 0; t < 3 * n_tris; ++t) tv[t] = remap[tris[t]];
    std::vector<P3> tn(n_tris);
    std::vector<int> owner((size_t)nv * nv, -1);
    for (int t = 0; t < n_tris; ++t) {
        int a = tv[3 * t], b = tv[3 * t + 1], c = tv[3 * t + 2];
        P3 n = cross(sub(pts[b], pts[a]), sub(pts[c], pts[a]));
        float l = len(n);
        if (a == b || b == c || a == c || !(l > 0.0f)) return false;
        tn[t] = mul(n, 1.0f / l);
        const int e[3][2] = {{a, b}, {b, c}, {c, a}};
        for (int k = 0; k < 3; ++k) { if (owner[(size_t)e[k][0] * nv + e[k][1]] >= 0) return false; owner[(size_t)e[k][0] * nv + e[k][1]] = t; }
    }
    for (int t = 0; t < n_tris; ++t)
        for (int k = 0; k < 3; ++k) if (owner[(size_t)tv[3 * t + (k + 1) % 3] * nv + tv[3 * t + k]] < 0) return false; // closed: every edge has its twin
    std::vector<int> uf(n_tris);
    for (int t = 0; t < n_tris; ++t) uf[t] = t;
    for (int t = 0; t < n_tris; ++t)
        for (int k = 0; k < 3; ++k) {
            int o = owner[(size_t)tv[3 * t + (k + 1) % 3] * nv + tv[3 * t + k]];
            if (o > t && dot(tn[t], tn[o]) > 1.0f - 1.0e-5f) { int ra = uf_find(uf, t), rb = uf_find(uf, o); if (ra != rb) uf[ra > rb ? ra : rb] = ra > rb ? rb : ra; }
        }
    int nfaces = 0;
    std::vector<int> face_of(n_tris, -1);
    for (int t = 0; t < n_tris; ++t) { int r = uf_find(uf, t); if (face_of[r] < 0) face_of[r] = nfaces++; }
    for (int t = 0; t < n_tris; ++t) face_of[t] = face_of[uf_find(uf, t)];
    std::vector<int> lv, lfirst(nfaces), lcount(nfaces), next(nv);
    for (int f = 0; f < nfaces; ++f) {
        std::fill(next.begin(), next.end(), -1);
        int nb = 0, start = nv;
        for (int t = 0; t < n_tris; ++t) {
            if (face_of[t] != f) continue;
            for (int k = 0; k < 3; ++k) {
                int a = tv[3 * t + k], b = tv[3 * t + (k + 1) % 3];
                if (face_of[owner[(size_t)b * nv + a]] == f) continue; // an inner edge of the face
                if (next[a] >= 0) return false;                       // the boundary passes a vertex twice
                next[a] = b; ++nb;
                if (a < start) start = a;
            }
        }
        if (nb < 3) return false;
        lfirst[f] = (int)lv.size(); lcount[f] = nb;
        int cur = start;
        for (int k = 0; k < nb; ++k) { lv.push_back(cur); cur = next[cur]; if (cur < 0) return false; }
        if (cur != start) return false;
    }
    std::vector<int> order(nfaces);
    for (int f = 0; f < nfaces; ++f) order[f] = f;
    auto cmp_loops = [&](int fa, int fb) {
        int na = lcount[fa], nb = lcount[fb], n = na < nb ? na : nb;
        for (int i = 0; i < n; ++i) if (lv[lfirst[fa] + i] != lv[lfirst[fb] + i]) return lv[lfirst[fa] + i] < lv[lfirst[fb] + i] ? -1 : 1;
        return na == nb ? 0 : (na < nb ? -1 : 1);
    };
    for (int i = 1; i < nfaces; ++i) { int o = order[i], j = i; while (j > 0 && cmp_loops(order[j - 1], o) > 0) { order[j] = order[j - 1]; --j; } order[j] = o; }
    P.ffirst.assign(nfaces, 0); P.fcount.assign(nfaces, 0); P.loop_v.clear();
    for (int i = 0; i < nfaces; ++i) {
        int f = order[i];
        P.ffirst[i] = (int)P.loop_v.size(); P.fcount[i] = lcount[f];
        for (int k = 0; k < lcount[f]; ++k) P.loop_v.push_back(lv[lfirst[f] + k]);
    }
    const int nl = (int)P.loop_v.size();
    std::vector<int> ekey;
    for (int f = 0; f < nfaces; ++f)
        for (int k = 0; k < P.fcount[f]; ++k) {
            int a = P.loop_v[P.ffirst[f] + k], b = P.loop_v[P.ffirst[f] + (k + 1) % P.fcount[f]];
            if (a < b) ekey.push_back(a * nv + b);
        }
    std::sort(ekey.begin(), ekey.end());
    P.ne = (int)ekey.size();
    if (P.ne * 2 != nl) return false;
    P.loop_e.assign(nl, -1);
    for (int f = 0; f < nfaces; ++f)
        for (int k = 0; k < P.fcount[f]; ++k) {
            int a = P.loop_v[P.ffirst[f] + k], b = P.loop_v[P.ffirst[f] + (k + 1) % P.fcount[f]];
            int key = a < b ? a * nv + b : b * nv + a;
            auto it = std::lower_bound(ekey.begin(), ekey.end(), key);
            if (it == ekey.end() || *it != key) return false;
            P.loop_e[P.ffirst[f] + k] = (int)(it - ekey.begin());
        }
    // mass properties in the given frame
    P3 gc = mk(0, 0, 0);
    for (int i = 0; i < nv; ++i) gc = add(gc, pts[i]);
    gc = mul(gc, 1.0f / (float)nv);
    P3 res = mk(0, 0, 0); float vol = 0.0f;
    for (int f = 0; f < nfaces; ++f)
        for (int k = 1; k + 1 < P.fcount[f]; ++k) {
            P3 p2 = pts[P.loop_v[P.ffirst[f]]], p3 = pts[P.loop_v[P.ffirst[f] + k]], p4 = pts[P.loop_v[P.ffirst[f] + k + 1]];
            float tvol = tet_signed_volume(gc, p2, p3, p4);
            P3 c = mul(add(add(add(gc, p2), p3), p4), 0.25f);
            res = add(res, mul(c, tvol)); vol += tvol;
        }
    if (!(vol > 0.0f)) return false;
    P3 com = mul(res, 1.0f / vol);
    float itot[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int f = 0; f < nfaces; ++f)
        for (int k = 1; k + 1 < P.fcount[f]; ++k) {
            P3 p2 = pts[P.loop_v[P.ffirst[f]]], p3 = pts[P.loop_v[P.ffirst[f] + k]], p4 = pts[P.loop_v[P.ffirst[f] + k + 1]];
            float tvol = tet_signed_volume(com, p2, p3, p4);
            float ip[3][3]; tet_unit_inertia(com, com, p2, p3, p4, ip);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) itot[i][j] = itot[i][j] + ip[i][j] * tvol;
        }
    P.volume = vol; P.com[0] = com.x; P.com[1] = com.y; P.com[2] = com.z; memcpy(P.inertia, itot, sizeof(itot));
    P.sphere_centre[0] = gc.x; P.sphere_centre[1] = gc.y; P.sphere_centre[2] = gc.z; P.sphere_radius = 0.0f;
    P3 mn = pts[0], mx = pts[0];
    for (int i = 0; i < nv; ++i) {
        float d = len(sub(pts[i], gc)); if (d > P.sphere_radius) P.sphere_radius = d;
        mn = mk(std::min(mn.x, pts[i].x), std::min(mn.y, pts[i].y), std::min(mn.z, pts[i].z));
        mx = mk(std::max(mx.x, pts[i].x), std::max(mx.y, pts[i].y), std::max(mx.z, pts[i].z));
    }
    P3 ctr = mul(add(mn, mx), 0.5f);
    P.centre[0] = ctr.x; P.centre[1] = ctr.y; P.centre[2] = ctr.z;
    P3 half = mk(0, 0, 0); P.origin_radius = 0.0f;
    P.pts.resize(3 * (size_t)nv);
    for (int i = 0; i < nv; ++i) {
        pts[i] = sub(pts[i], ctr);
        half = mk(std::max(half.x, std::fabs(pts[i].x)), std::max(half.y, std::fabs(pts[i].y)), std::max(half.z, std::fabs(pts[i].z)));
        float d = len(pts[i]); if (d > P.origin_radius) P.origin_radius = d;
        P.pts[3 * i] = pts[i].x; P.pts[3 * i + 1] = pts[i].y; P.pts[3 * i + 2] = pts[i].z;
    }
    P.half[0] = half.x; P.half[1] = half.y; P.half[2] = half.z;
    P.fnormal.resize(3 * (size_t)nfaces);
    for (int f = 0; f < nfaces; ++f) {
        P3 n = mk(0, 0, 0);
        for (int k = 0; k < P.fcount[f]; ++k) {
            P3 a = pts[P.loop_v[P.ffirst[f] + k]], b = pts[P.loop_v[P.ffirst[f] + (k + 1) % P.fcount[f]]];
            n = add(n, mk((a.y - b.y) * (a.z + b.z), (a.z - b.z) * (a.x + b.x), (a.x - b.x) * (a.y + b.y)));
        }
        float l = len(n);
        if (!(l > 0.0f)) return false;
        n = mul(n, 1.0f / l);
        P.fnormal[3 * f] = n.x; P.fnormal[3 * f + 1] = n.y; P.fnormal[3 * f + 2] = n.z;
    }
    return true;
}

// parry's transformation::convex_hull stands behind ColliderBuilder::convex_hull; in its place an incremental hull (double precision):
// outward-wound triangles over the input points' indices; false = the points are (numerically) coplanar
static bool convex_hull(int n, const float *xyz, std::vector<uint32_t> &out) {
    struct D3 { double x, y, z; };
    auto P = [&](int i) { D3 r = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}; return r; };
    auto dsub = [](D3 a, D3 b) { D3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; };
    auto ddot = [](D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
    auto dcross = [](D3 a, D3 b) { D3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; return r; };
    if (n < 4) return false;
    double scale = 0.0;
    for (int i = 0; i < 3 * n; ++i) { if (!std::isfinite(xyz[i])) return false; scale = std::max(scale, (double)std::fabs(xyz[i])); }
    if (!(scale > 0.0)) return false;
    const double eps = 1.0e-7 * scale;
    // first tetrahedron: the two points farthest apart along x, the point farthest from their line, the point farthest from that plane
    int i0 = 0, i1 = 0;
    for (int i = 1; i < n; ++i) { if (xyz[3 * i] < xyz[3 * i0]) i0 = i; if (xyz[3 * i] > xyz[3 * i1]) i1 = i; }
    if (i0 == i1) { for (int i = 1; i < n; ++i) { if (xyz[3 * i + 1] < xyz[3 * i0 + 1]) i0 = i; if (xyz[3 * i + 1] > xyz[3 * i1 + 1]) i1 = i; } }
    if (i0 == i1) { for (int i = 1; i < n; ++i) { if (xyz[3 * i + 2] < xyz[3 * i0 + 2]) i0 = i; if (xyz[3 * i + 2] > xyz[3 * i1 + 2]) i1 = i; } }
    if (i0 == i1) return false;
    D3 d01 = dsub(P(i1), P(i0));
    int i2 = -1; double best = eps * eps * ddot(d01, d01) / (scale * scale);
    for (int i = 0; i < n; ++i) { D3 c = dcross(d01, dsub(P(i), P(i0))); double v = ddot(c, c); if (v > best) { best = v; i2 = i; } }
    if (i2 < 0) return false;
    D3 nrm = dcross(d01, dsub(P(i2), P(i0)));
    const double nl = std::sqrt(ddot(nrm, nrm));
    int i3 = -1; double bestd = eps;
    for (int i = 0; i < n; ++i) { double v = std::fabs(ddot(nrm, dsub(P(i), P(i0)))) / nl; if (v > bestd) { bestd = v; i3 = i; } }
    if (i3 < 0) return false;
    struct F { int a, b, c; D3 n; double d; bool alive; };
    std::vector<F> faces;
    auto add_face = [&](int a, int b, int c) {
        D3 nn = dcross(dsub(P(b), P(a)), dsub(P(c), P(a)));
        double l = std::sqrt(ddot(nn, nn));
        F f; f.a = a; f.b = b; f.c = c; f.alive = true;
        if (l > 0.0) { f.n.x = nn.x / l; f.n.y = nn.y / l; f.n.z = nn.z / l; } else { f.n.x = f.n.y = f.n.z = 0.0; }
        f.d = ddot(f.n, P(a));
        faces.push_back(f);
    };
    if (ddot(nrm, dsub(P(i3), P(i0))) > 0.0) std::swap(i1, i2); // i3 on the inner side of (i0, i1, i2)
    add_face(i0, i1, i2); add_face(i0, i2, i3); add_face(i0, i3, i1); add_face(i1, i3, i2);
    std::vector<std::pair<int, int>> horizon;
    for (int i = 0; i < n; ++i) {
        if (i == i0 || i == i1 || i == i2 || i == i3) continue;
        horizon.clear();
        bool any = false;
        for (size_t f = 0; f < faces.size(); ++f) {
            F &g = faces[f];
            if (!g.alive || !(ddot(g.n, P(i)) - g.d > eps)) continue;
            g.alive = false; any = true;
            const int e[3][2] = {{g.a, g.b}, {g.b, g.c}, {g.c, g.a}};
            for (int k = 0; k < 3; ++k) {
                auto it = std::find(horizon.begin(), horizon.end(), std::make_pair(e[k][1], e[k][0]));
                if (it != horizon.end()) horizon.erase(it); else horizon.push_back(std::make_pair(e[k][0], e[k][1]));
            }
        }
        if (!any) continue;
        for (auto &e : horizon) add_face(e.first, e.second, i);
    }
    out.clear();
    for (auto &f : faces) if (f.alive) { out.push_back((uint32_t)f.a); out.push_back((uint32_t)f.b); out.push_back((uint32_t)f.c); }
    return out.size() >= 12;
}
} // namespace rp_poly
