/* ro_polyhedron.h — convex polyhedra of the oracle (test infrastructure only, like the rest of oracle/).
 *
 * ColliderBuilder::convex_mesh(points, indices) / convex_hull(points) (/root/reference/src/geometry/collider.rs:1039, :1070) build
 * parry3d's ConvexPolyhedron (from_convex_mesh: triangles with equal normals are merged into polygonal faces, every face knows its
 * vertex loop and its edges), MassProperties::from_convex_polyhedron (signed tetrahedra from the centre of mass, Tonon's closed-form
 * inertia tensor), point_cloud_bounding_sphere (centre = the mean of the points).  parry3d is not under /root/reference: the
 * construction below is a canonical form of our own — the same polyhedron whatever triangulation of its faces comes in:
 *   vertices   the points the triangles use, in index order;
 *   faces      maximal sets of edge-adjacent triangles whose unit normals agree (dot > 1 - 1e-5), as vertex loops, counter-clockwise
 *              seen from outside, starting at the loop's smallest vertex; faces sorted by their loops; normal by Newell's sums;
 *   edges      the faces' boundary edges, sorted by (smaller vertex, larger vertex);
 *   feature ids  vertex v -> v, edge e -> 0x4000 | e, face f -> 0x8000 | f;
 *   mass       parry's formulas over the fan triangulation of the canonical loops.
 * A collider stores the polyhedron RECENTRED on the centre of its local AABB, with that offset folded into the collider's pose
 * (pos_wrt_parent * translation(centre)): the broad phase, the recycle extents and the CCD pre-filter then treat it like every
 * other shape whose local box is symmetric about the collider origin.  The AABB is the local box transformed (Cylinder / Cone do
 * the same in parry; ConvexPolyhedron::aabb is the exact point-cloud box: ours contains it — a superset of near-miss pairs, the same
 * contacts).  rapier_amd/csrc/rp_api.hip builds the same form on the host of the product; tests compare the two. */
#ifndef RO_POLYHEDRON_H
#define RO_POLYHEDRON_H
#include "ro_math.h"
#include <stdlib.h>
#include <string.h>

#define RO_POLY_MAX_VERTS 256
#define RO_FID_EDGE 0x4000u
#define RO_FID_FACE 0x8000u

typedef struct {
    int nv; v3 *pts;                 /* recentred on the local AABB's centre */
    int nf; v3 *fnormal; int *ffirst, *fcount;
    int nloop; int *loop_v, *loop_e; /* all faces' vertex loops, concatenated; loop_e[k] = the edge (loop_v[k], next) */
    int ne;
    v3 centre;                       /* what was subtracted: the local AABB's centre in the frame the points were given in */
    v3 half;                         /* half extents of the local AABB */
    float origin_radius;             /* max |p| of the recentred points */
    v3 sphere_centre; float sphere_radius; /* point_cloud_bounding_sphere, in the ORIGINAL frame */
    float volume; v3 com; float inertia[3][3]; /* unit density, ORIGINAL frame, tensor about the centre of mass */
} RoPolyhedron;

static inline void ro_poly_free(RoPolyhedron *p) {
    free(p->pts); free(p->fnormal); free(p->ffirst); free(p->fcount); free(p->loop_v); free(p->loop_e);
    memset(p, 0, sizeof(*p));
}
static inline int ro_poly_uf_find(int *uf, int x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; }
static inline int ro_poly_cmp_loops(const int *a, int na, const int *b, int nb) {
    int n = na < nb ? na : nb;
    for (int i = 0; i < n; ++i) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return na == nb ? 0 : (na < nb ? -1 : 1);
}
/* Tetrahedron::signed_volume / tetrahedron_unit_inertia_tensor_wrt_point (Tonon 2004), p1 = the reference point itself */
static inline float ro_tet_signed_volume(v3 p1, v3 p2, v3 p3, v3 p4) {
    v3 a = vsub(p2, p1), b = vsub(p3, p1), c = vsub(p4, p1);
    return vdot(a, vcross(b, c)) / 6.0f;
}
static inline void ro_tet_unit_inertia(v3 pt, v3 p1, v3 p2, v3 p3, v3 p4, float out[3][3]) {
    v3 q1 = vsub(p1, pt), q2 = vsub(p2, pt), q3 = vsub(p3, pt), q4 = vsub(p4, pt);
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

/* 0 = ok; -1 = not a closed convex triangle mesh this construction can take (too few / too many vertices, an edge that is not
 * shared by exactly two triangles with opposite directions, a face without area, no volume) */
static inline int ro_poly_build(RoPolyhedron *P, int n_points, const float *xyz, int n_tris, const uint32_t *tris) {
    memset(P, 0, sizeof(*P));
    if (n_points < 4 || n_tris < 4) return -1;
    /* vertices in use, in index order */
    int *remap = (int *)malloc(sizeof(int) * n_points);
    for (int i = 0; i < n_points; ++i) remap[i] = -1;
    for (int t = 0; t < 3 * n_tris; ++t) { if (tris[t] >= (uint32_t)n_points) { free(remap); return -1; } remap[tris[t]] = 0; }
    int nv = 0;
    for (int i = 0; i < n_points; ++i) if (remap[i] == 0) remap[i] = nv++;
    if (nv < 4 || nv > RO_POLY_MAX_VERTS) { free(remap); return -1; }
    v3 *pts = (v3 *)malloc(sizeof(v3) * nv);
    for (int i = 0; i < n_points; ++i) if (remap[i] >= 0) pts[remap[i]] = V3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    int *tv = (int *)malloc(sizeof(int) * 3 * n_tris);
    for (int t = 0; t < 3 * n_tris; ++t) tv[t] = remap[tris[t]];
    free(remap);
    /* unit normals; directed-edge table: owner[a * nv + b] = the triangle with the edge a -> b */
    v3 *tn = (v3 *)malloc(sizeof(v3) * n_tris);
    int *owner = (int *)malloc(sizeof(int) * nv * nv);
    for (int i = 0; i < nv * nv; ++i) owner[i] = -1;
    int bad = 0;
    for (int t = 0; t < n_tris; ++t) {
        int a = tv[3 * t], b = tv[3 * t + 1], c = tv[3 * t + 2];
        v3 n = vcross(vsub(pts[b], pts[a]), vsub(pts[c], pts[a]));
        float l = vlen(n);
        if (a == b || b == c || a == c || !(l > 0.0f)) { bad = 1; break; }
        tn[t] = vmul(n, 1.0f / l);
        int e[3][2] = {{a, b}, {b, c}, {c, a}};
        for (int k = 0; k < 3; ++k) { if (owner[e[k][0] * nv + e[k][1]] >= 0) bad = 1; owner[e[k][0] * nv + e[k][1]] = t; }
    }
    for (int t = 0; t < n_tris && !bad; ++t)
        for (int k = 0; k < 3; ++k) if (owner[tv[3 * t + (k + 1) % 3] * nv + tv[3 * t + k]] < 0) bad = 1; /* closed: every edge has its twin */
    if (bad) { free(pts); free(tv); free(tn); free(owner); return -1; }
    /* faces: union of edge-adjacent triangles with the same normal */
    int *uf = (int *)malloc(sizeof(int) * n_tris);
    for (int t = 0; t < n_tris; ++t) uf[t] = t;
    for (int t = 0; t < n_tris; ++t)
        for (int k = 0; k < 3; ++k) {
            int o = owner[tv[3 * t + (k + 1) % 3] * nv + tv[3 * t + k]];
            if (o > t && vdot(tn[t], tn[o]) > 1.0f - 1.0e-5f) { int ra = ro_poly_uf_find(uf, t), rb = ro_poly_uf_find(uf, o); if (ra != rb) uf[ra > rb ? ra : rb] = ra > rb ? rb : ra; }
        }
    /* boundary loops: next[a] = b for the directed edges of a face whose twin belongs to another face */
    int nfaces = 0;
    int *face_of = (int *)malloc(sizeof(int) * n_tris);
    for (int t = 0; t < n_tris; ++t) face_of[t] = -1;
    for (int t = 0; t < n_tris; ++t) { int r = ro_poly_uf_find(uf, t); if (face_of[r] < 0) face_of[r] = nfaces++; }
    for (int t = 0; t < n_tris; ++t) face_of[t] = face_of[ro_poly_uf_find(uf, t)];
    int *lv = (int *)malloc(sizeof(int) * 3 * n_tris), *lfirst = (int *)malloc(sizeof(int) * nfaces), *lcount = (int *)malloc(sizeof(int) * nfaces);
    int *next = (int *)malloc(sizeof(int) * nv);
    int nl = 0;
    for (int f = 0; f < nfaces && !bad; ++f) {
        for (int i = 0; i < nv; ++i) next[i] = -1;
        int nb = 0, start = nv;
        for (int t = 0; t < n_tris; ++t) {
            if (face_of[t] != f) continue;
            for (int k = 0; k < 3; ++k) {
                int a = tv[3 * t + k], b = tv[3 * t + (k + 1) % 3];
                if (face_of[owner[b * nv + a]] == f) continue;     /* an inner edge of the face */
                if (next[a] >= 0) bad = 1;                        /* the boundary passes a vertex twice: not a simple polygon */
                next[a] = b; ++nb;
                if (a < start) start = a;
            }
        }
        if (nb < 3) bad = 1;
        lfirst[f] = nl; lcount[f] = nb;
        int cur = start;
        for (int k = 0; k < nb && !bad; ++k) { lv[nl++] = cur; cur = next[cur]; if (cur < 0) bad = 1; }
        if (!bad && cur != start) bad = 1;
    }
    free(next); free(uf); free(face_of); free(owner); free(tn); free(tv);
    if (bad) { free(pts); free(lv); free(lfirst); free(lcount); return -1; }
    /* faces sorted by their loops */
    int *order = (int *)malloc(sizeof(int) * nfaces);
    for (int f = 0; f < nfaces; ++f) order[f] = f;
    for (int i = 1; i < nfaces; ++i) {
        int o = order[i], j = i;
        while (j > 0 && ro_poly_cmp_loops(lv + lfirst[order[j - 1]], lcount[order[j - 1]], lv + lfirst[o], lcount[o]) > 0) { order[j] = order[j - 1]; --j; }
        order[j] = o;
    }
    P->nv = nv; P->nf = nfaces; P->nloop = nl;
    P->fnormal = (v3 *)malloc(sizeof(v3) * nfaces); P->ffirst = (int *)malloc(sizeof(int) * nfaces); P->fcount = (int *)malloc(sizeof(int) * nfaces);
    P->loop_v = (int *)malloc(sizeof(int) * nl); P->loop_e = (int *)malloc(sizeof(int) * nl);
    int w = 0;
    for (int i = 0; i < nfaces; ++i) {
        int f = order[i];
        P->ffirst[i] = w; P->fcount[i] = lcount[f];
        for (int k = 0; k < lcount[f]; ++k) P->loop_v[w++] = lv[lfirst[f] + k];
    }
    free(order); free(lv); free(lfirst); free(lcount);
    /* edges: sorted (min, max) pairs of the loops' edges */
    int *ekey = (int *)malloc(sizeof(int) * nl);
    int ne = 0;
    for (int f = 0; f < nfaces; ++f)
        for (int k = 0; k < P->fcount[f]; ++k) {
            int a = P->loop_v[P->ffirst[f] + k], b = P->loop_v[P->ffirst[f] + (k + 1) % P->fcount[f]];
            if (a < b) ekey[ne++] = a * nv + b; /* each undirected edge once: from the face that runs it upwards */
        }
    for (int i = 1; i < ne; ++i) { int o = ekey[i], j = i; while (j > 0 && ekey[j - 1] > o) { ekey[j] = ekey[j - 1]; --j; } ekey[j] = o; }
    P->ne = ne;
    for (int f = 0; f < nfaces; ++f)
        for (int k = 0; k < P->fcount[f]; ++k) {
            int a = P->loop_v[P->ffirst[f] + k], b = P->loop_v[P->ffirst[f] + (k + 1) % P->fcount[f]];
            int key = a < b ? a * nv + b : b * nv + a, lo = 0, hi = ne - 1, at = -1;
            while (lo <= hi) { int m = (lo + hi) / 2; if (ekey[m] == key) { at = m; break; } if (ekey[m] < key) lo = m + 1; else hi = m - 1; }
            if (at < 0) bad = 1;
            P->loop_e[P->ffirst[f] + k] = at;
        }
    free(ekey);
    if (bad || ne * 2 != nl) { free(pts); ro_poly_free(P); return -1; }
    /* mass properties in the given frame (parry's formulas over the fan triangulation of the canonical loops) */
    v3 gc = V3(0, 0, 0);
    for (int i = 0; i < nv; ++i) gc = vadd(gc, pts[i]);
    gc = vmul(gc, 1.0f / (float)nv);
    v3 res = V3(0, 0, 0); float vol = 0.0f;
    for (int f = 0; f < nfaces; ++f)
        for (int k = 1; k + 1 < P->fcount[f]; ++k) {
            v3 p2 = pts[P->loop_v[P->ffirst[f]]], p3 = pts[P->loop_v[P->ffirst[f] + k]], p4 = pts[P->loop_v[P->ffirst[f] + k + 1]];
            float tvol = ro_tet_signed_volume(gc, p2, p3, p4);
            v3 c = vmul(vadd(vadd(vadd(gc, p2), p3), p4), 0.25f);
            res = vadd(res, vmul(c, tvol)); vol += tvol;
        }
    if (!(vol > 0.0f)) { free(pts); ro_poly_free(P); return -1; } /* (an inward-wound mesh has a negative volume) */
    v3 com = vmul(res, 1.0f / vol);
    float itot[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int f = 0; f < nfaces; ++f)
        for (int k = 1; k + 1 < P->fcount[f]; ++k) {
            v3 p2 = pts[P->loop_v[P->ffirst[f]]], p3 = pts[P->loop_v[P->ffirst[f] + k]], p4 = pts[P->loop_v[P->ffirst[f] + k + 1]];
            float tvol = ro_tet_signed_volume(com, p2, p3, p4);
            float ip[3][3]; ro_tet_unit_inertia(com, com, p2, p3, p4, ip);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) itot[i][j] = itot[i][j] + ip[i][j] * tvol;
        }
    P->volume = vol; P->com = com; memcpy(P->inertia, itot, sizeof(itot));
    /* bounding sphere (centre = the mean of the points), local AABB, recentring */
    P->sphere_centre = gc; P->sphere_radius = 0.0f;
    v3 mn = pts[0], mx = pts[0];
    for (int i = 0; i < nv; ++i) {
        float d = vlen(vsub(pts[i], gc)); if (d > P->sphere_radius) P->sphere_radius = d;
        mn = V3(ro_minf(mn.x, pts[i].x), ro_minf(mn.y, pts[i].y), ro_minf(mn.z, pts[i].z));
        mx = V3(ro_maxf(mx.x, pts[i].x), ro_maxf(mx.y, pts[i].y), ro_maxf(mx.z, pts[i].z));
    }
    P->centre = vmul(vadd(mn, mx), 0.5f);
    P->half = V3(0, 0, 0); P->origin_radius = 0.0f;
    for (int i = 0; i < nv; ++i) {
        pts[i] = vsub(pts[i], P->centre);
        P->half = V3(ro_maxf(P->half.x, fabsf(pts[i].x)), ro_maxf(P->half.y, fabsf(pts[i].y)), ro_maxf(P->half.z, fabsf(pts[i].z)));
        float d = vlen(pts[i]); if (d > P->origin_radius) P->origin_radius = d;
    }
    P->pts = pts;
    /* face normals (Newell) from the recentred points */
    for (int f = 0; f < nfaces; ++f) {
        v3 n = V3(0, 0, 0);
        for (int k = 0; k < P->fcount[f]; ++k) {
            v3 a = pts[P->loop_v[P->ffirst[f] + k]], b = pts[P->loop_v[P->ffirst[f] + (k + 1) % P->fcount[f]]];
            n = vadd(n, V3((a.y - b.y) * (a.z + b.z), (a.z - b.z) * (a.x + b.x), (a.x - b.x) * (a.y + b.y)));
        }
        float l = vlen(n);
        if (!(l > 0.0f)) { ro_poly_free(P); return -1; }
        P->fnormal[f] = vmul(n, 1.0f / l);
    }
    return 0;
}
#endif
