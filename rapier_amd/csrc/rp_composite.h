// rp_composite.h — composite shapes as ONE collider (compound, triangle mesh, height field) and contact clustering on the device.
//
//   ColliderBuilder::compound / trimesh / heightfield      /root/reference/src/geometry/collider.rs:711, :944, :1089
//   cluster_manifolds_for_solver, carry_warmstart_data     /root/reference/src/geometry/contact_clustering.rs:33, :129
//   use_clusters = contact_clustering && manifolds.len() > 1   /root/reference/src/geometry/narrow_phase/pair_update.rs:350
//   solver manifolds 2+ of a pair -> overflow colour       /root/reference/src/geometry/narrow_phase/solver_graph.rs:534-547
//
// What parry does for these shapes (one manifold per sub-shape pair whose AABBs meet, the part's pose folded into the relative pose) is
// restated from its published algorithm with the simplifications listed in DESIGN.md section 5 (index-ordered candidates, candidate
// set from the tight loosened AABB, sub-manifolds persistent only on the one-candidate path, triangles through the support-map
// path, bounded cluster / point / candidate counts); the oracle (oracle/ro_composite.h) states the same rules and the device agrees
// with it bit for bit.
//
// Layout.  A composite collider is an ordinary collider to the broad phase: c_he = the half extents of its local AABB (the AABB's
// centre rides in the collider's pose, like a polyhedron's), c_he.w = its row in cm_hdr.  Sub-shapes live in flat tables
// (cm_min / cm_max: AABBs in the composite frame; cm_a / cm_b / cm_c: a compound part's c_he, pose translation (+ shape), pose
// rotation — or a triangle's three vertices).  A pair with a composite collider keeps solver manifold 0 in its own pair slot and
// solver manifolds (clusters) 2..4 in AUXILIARY slots of the same pool (RP_PF_AUX, p_aux): to the solver an aux slot is one more
// manifold between the same two bodies, in the overflow colour; the broad and the narrow phase skip it, its parent owns it.
//
// k_np_composite runs behind k_np_update over the same queue (np_list) and takes the composite pairs: ONE THREAD per pair walks the
// candidates, builds the clusters in a per-thread workspace in HBM (cm_ws), carries the warm-start data by position and writes the
// solver manifolds.  Correctness first: a scene with thousands of composite pairs in contact wants a wavefront per pair.
#pragma once

#define RP_MAX_CLUSTERS 4      // solver manifolds per pair (oracle: RO_MAX_CLUSTERS)
#define RP_CLUSTER_PTS 32      // points of one cluster while it is built (oracle: RO_CLUSTER_PTS)
#define RP_MAX_SUBPAIRS 64     // candidate sub-shape pairs of one collider pair per step (oracle: RO_MAX_SUBPAIRS)
#define RP_CM_WS_F4 (RP_MAX_CLUSTERS * RP_CLUSTER_PTS * 4 + RP_MAX_SUBPAIRS / 2) // float4 per thread: 4 per cluster point + the candidate list
#define RP_COS_MERGE_ANGLE 0.996f


// Shape::compute_aabb(pos) of a primitive in the c_he layout (the rule of collider_update_one at an arbitrary pose, not loosened)
RP_DEV void prim_aabb_at(int sh_in, float4 he, float border, const V3 *tri, Pose at, V3 &mn, V3 &mx) {
    const int sh = sm_core_shape(sh_in);
    if (sh == RP_SHAPE_TRIANGLE) {
        V3 a = pose_tp(at, tri[0]), b = pose_tp(at, tri[1]), c = pose_tp(at, tri[2]);
        mn = a; mx = a;
        mn = v3(rp_min(mn.x, b.x), rp_min(mn.y, b.y), rp_min(mn.z, b.z)); mx = v3(rp_max(mx.x, b.x), rp_max(mx.y, b.y), rp_max(mx.z, b.z));
        mn = v3(rp_min(mn.x, c.x), rp_min(mn.y, c.y), rp_min(mn.z, c.z)); mx = v3(rp_max(mx.x, c.x), rp_max(mx.y, c.y), rp_max(mx.z, c.z));
        return;
    }
    if (sh == RP_SHAPE_CUBOID || sh >= RP_SHAPE_CYLINDER) {
        float m[3][3]; quat_to_mat(at.r, m);
        V3 h = v3(fabsf(m[0][0]) * he.x + fabsf(m[0][1]) * he.y + fabsf(m[0][2]) * he.z,
                  fabsf(m[1][0]) * he.x + fabsf(m[1][1]) * he.y + fabsf(m[1][2]) * he.z,
                  fabsf(m[2][0]) * he.x + fabsf(m[2][1]) * he.y + fabsf(m[2][2]) * he.z);
        mn = at.t - h; mx = at.t + h;
    } else if (sh == RP_SHAPE_CAPSULE) {
        V3 e = capsule_axis_dir((int)he.z);
        V3 pa = pose_tp(at, e * -he.x), pb = pose_tp(at, e * he.x);
        V3 r = v3(he.y, he.y, he.y);
        mn = v3(rp_min(pa.x, pb.x), rp_min(pa.y, pb.y), rp_min(pa.z, pb.z)) - r;
        mx = v3(rp_max(pa.x, pb.x), rp_max(pa.y, pb.y), rp_max(pa.z, pb.z)) + r;
    } else {
        V3 h = v3(he.x, he.x, he.x);
        mn = at.t - h; mx = at.t + h;
    }
    if (border > 0.0f) { V3 b = v3(border, border, border); mn = mn - b; mx = mx + b; }
}
RP_DEV bool aabb_meet(V3 amn, V3 amx, V3 bmn, V3 bmx) { // Aabb::intersects
    return amn.x <= bmx.x && bmn.x <= amx.x && amn.y <= bmx.y && bmn.y <= amx.y && amn.z <= bmx.z && bmn.z <= amx.z;
}

// sub-shape i of collider c as the dispatcher wants it
struct SubShape { int sh; float4 he; float bd; V3 tri[3]; Pose pos; bool has_pose; };
RP_DEV SubShape co_sub(const DevWorld &w, int c, int i) {
    SubShape o;
    o.pos.r = q4(0, 0, 0, 1); o.pos.t = v3(0, 0, 0); o.has_pose = false;
    o.tri[0] = v3(0, 0, 0); o.tri[1] = o.tri[0]; o.tri[2] = o.tri[0];
    const int sh = w.c_shape[c];
    if (!shape_is_composite(sh)) { o.sh = sh; o.he = w.c_he[c]; o.bd = w.c_mat[c].w; return o; }
    const int4 h = w.cm_hdr[__float_as_int(w.c_he[c].w)];
    const int row = h.y + i;
    if (sh == RP_SHAPE_COMPOUND) {
        const float4 b = w.cm_b[row];
        o.sh = __float_as_int(b.w); o.he = w.cm_a[row]; o.bd = w.cm_border[row];
        o.pos.t = v3(b); o.pos.r = q4(w.cm_c[row]); o.has_pose = true;
    } else {
        o.sh = RP_SHAPE_TRIANGLE; o.he = make_float4(0, 0, 0, 0); o.bd = 0.0f;
        o.tri[0] = v3(w.cm_a[row]); o.tri[1] = v3(w.cm_b[row]); o.tri[2] = v3(w.cm_c[row]);
    }
    return o;
}
RP_DEV int co_num_subs(const DevWorld &w, int c) { return shape_is_composite(w.c_shape[c]) ? w.cm_hdr[__float_as_int(w.c_he[c].w)].z : 1; }
RP_DEV int co_first_sub(const DevWorld &w, int c) { return w.cm_hdr[__float_as_int(w.c_he[c].w)].y; }

// the per-thread workspace: element e of thread t at cm_ws[e * cm_ws_threads + t]
struct CmWs {
    float4 *base; int stride;
    RP_DEV float4 &at(int e) const { return base[(size_t)e * stride]; }
    RP_DEV float4 &A(int c, int k) const { return at(((c * RP_CLUSTER_PTS + k) << 2) + 0); } // local_p1.xyz, dist
    RP_DEV float4 &B(int c, int k) const { return at(((c * RP_CLUSTER_PTS + k) << 2) + 1); } // local_p2.xyz
    RP_DEV float4 &I(int c, int k) const { return at(((c * RP_CLUSTER_PTS + k) << 2) + 2); } // ContactData: impulse, warmstart_impulse, warmstart_twist
    RP_DEV float4 &W(int c, int k) const { return at(((c * RP_CLUSTER_PTS + k) << 2) + 3); } // ContactData: warmstart_tangent_world
    RP_DEV void cand_set(int q, int i1, int i2) const { float4 &f = at(RP_MAX_CLUSTERS * RP_CLUSTER_PTS * 4 + (q >> 1)); if (q & 1) { f.z = __int_as_float(i1); f.w = __int_as_float(i2); } else { f.x = __int_as_float(i1); f.y = __int_as_float(i2); } }
    RP_DEV int2 cand(int q) const { const float4 f = at(RP_MAX_CLUSTERS * RP_CLUSTER_PTS * 4 + (q >> 1)); return (q & 1) ? make_int2(__float_as_int(f.z), __float_as_int(f.w)) : make_int2(__float_as_int(f.x), __float_as_int(f.y)); }
};

// candidate sub-shape pairs of (c1, c2), ascending (i1, i2) — oracle: comp_candidates
__device__ int comp_candidates(const DevWorld &w, int c1, int c2, Pose pos12, float prediction, const CmWs &ws, bool &overflow) {
    int n = 0; overflow = false;
    const bool k1 = shape_is_composite(w.c_shape[c1]), k2 = shape_is_composite(w.c_shape[c2]);
    const V3 l = v3(prediction, prediction, prediction);
    if (k1 && !k2) {
        const SubShape b = co_sub(w, c2, 0);
        V3 bmn, bmx; prim_aabb_at(b.sh, b.he, b.bd, b.tri, pos12, bmn, bmx); bmn = bmn - l; bmx = bmx + l;
        const int first = co_first_sub(w, c1), cnt = co_num_subs(w, c1);
        for (int i = 0; i < cnt; ++i) if (aabb_meet(v3(w.cm_min[first + i]), v3(w.cm_max[first + i]), bmn, bmx)) { if (n < RP_MAX_SUBPAIRS) ws.cand_set(n++, i, -1); else overflow = true; }
    } else if (!k1 && k2) {
        const SubShape a = co_sub(w, c1, 0);
        V3 amn, amx; prim_aabb_at(a.sh, a.he, a.bd, a.tri, pose_inv(pos12), amn, amx); amn = amn - l; amx = amx + l;
        const int first = co_first_sub(w, c2), cnt = co_num_subs(w, c2);
        for (int j = 0; j < cnt; ++j) if (aabb_meet(amn, amx, v3(w.cm_min[first + j]), v3(w.cm_max[first + j]))) { if (n < RP_MAX_SUBPAIRS) ws.cand_set(n++, -1, j); else overflow = true; }
    } else {
        V3 bmn, bmx; prim_aabb_at(RP_SHAPE_CUBOID, w.c_he[c2], 0.0f, nullptr, pos12, bmn, bmx); bmn = bmn - l; bmx = bmx + l; // the whole of 2 in 1's frame: its local box
        const Pose pos21 = pose_inv(pos12);
        const int f1 = co_first_sub(w, c1), n1 = co_num_subs(w, c1), f2 = co_first_sub(w, c2), n2 = co_num_subs(w, c2);
        for (int i = 0; i < n1; ++i) {
            if (!aabb_meet(v3(w.cm_min[f1 + i]), v3(w.cm_max[f1 + i]), bmn, bmx)) continue;
            const SubShape a = co_sub(w, c1, i);
            V3 amn, amx; prim_aabb_at(a.sh, a.he, a.bd, a.tri, a.has_pose ? pose_mul(pos21, a.pos) : pos21, amn, amx); amn = amn - l; amx = amx + l;
            for (int j = 0; j < n2; ++j) if (aabb_meet(amn, amx, v3(w.cm_min[f2 + j]), v3(w.cm_max[f2 + j]))) { if (n < RP_MAX_SUBPAIRS) ws.cand_set(n++, i, j); else overflow = true; }
        }
    }
    return n;
}

RP_DEV bool data_has_warmstart(float4 imp) { return imp.x != 0.0f || imp.y != 0.0f; }
// carry_warmstart_data (:129-174) with the pair's previous solver manifolds (slot s + its aux slots) as source and the plain manifold
// `m` (points local to sub.pos1) as the one target — clustering stopped applying (pair_update.rs:385-396)
__device__ void composite_carry_to_plain(DevWorld &w, int s, const LocalManifold &m, const SubSel &sub, float4 *cimp, float4 *cwst) {
    for (int k = 0; k < RP_MAX_PTS; ++k) { cimp[k] = make_float4(0, 0, 0, 0); cwst[k] = make_float4(0, 0, 0, 0); }
    const float match_eps_sq = w.prm.prediction * w.prm.prediction;
    for (int a = 0; a < sub.prev_ncl; ++a) {
        const int sa = sm_slot(w, s, a);
        const V3 pn1 = v3(w.p_ln1[sa]);
        const int np = w.p_npts[sa];
        for (int i = 0; i < np; ++i) {
            const float4 pimp = PT(w.pt_imp, i, sa);
            if (!data_has_warmstart(pimp)) continue;
            if (dot(m.ln1, pn1) < RP_COS_MERGE_ANGLE) continue;
            const V3 pp = v3(PT(w.pt_lp1d, i, sa));
            int bp = -1; float best = match_eps_sq;
            for (int k = 0; k < m.n; ++k) {
                if (data_has_warmstart(cimp[k])) continue;
                const V3 p1 = sub.has_pos1 ? pose_tp(sub.pos1, m.lp1[k]) : m.lp1[k];
                const float d = len2(p1 - pp);
                if (d < best) { best = d; bp = k; }
            }
            if (bp >= 0) { cimp[bp] = pimp; cwst[bp] = PT(w.pt_wst, i, sa); }
        }
    }
}


// cluster_manifolds_for_solver (:49-122) for ONE sub-manifold that holds points (n1 / n2: its normals in the collider frames) — oracle: cluster_add_manifold
__device__ void cluster_add_manifold(const CmWs &ws, V3 *cn1, V3 *cn2, int *cnp, int &ncl, const LocalManifold &m, V3 n1, V3 n2, const SubShape &a, const SubShape &b, float dedup_eps_sq) {
    int id = -1;
    for (int c = 0; c < ncl; ++c) if (dot(cn1[c], n1) >= RP_COS_MERGE_ANGLE) { id = c; break; }
    if (id < 0) {
        if (ncl < RP_MAX_CLUSTERS) { id = ncl++; cn1[id] = n1; cn2[id] = n2; cnp[id] = 0; }
        else { float best = -2.0f; for (int c = 0; c < ncl; ++c) { float d = dot(cn1[c], n1); if (d > best) { best = d; id = c; } } } // (bound of this restatement)
    }
    for (int i = 0; i < m.n; ++i) {
        V3 p1 = m.lp1[i], p2 = m.lp2[i];
        const float dist = m.dist[i];
        if (a.has_pose) p1 = pose_tp(a.pos, p1);
        if (b.has_pose) p2 = pose_tp(b.pos, p2);
        int ex = -1;
        const int np = cnp[id];
        for (int k = 0; k < np; ++k) if (len2(v3(ws.A(id, k)) - p1) < dedup_eps_sq) { ex = k; break; }
        int at = -1;
        if (ex >= 0) { if (dist < ws.A(id, ex).w) at = ex; }
        else if (np < RP_CLUSTER_PTS) { at = np; cnp[id] = np + 1; }
        else { int sh = 0; for (int k = 1; k < np; ++k) if (ws.A(id, k).w > ws.A(id, sh).w) sh = k; if (dist < ws.A(id, sh).w) at = sh; }
        if (at >= 0) { ws.A(id, at) = f4(p1, dist); ws.B(id, at) = f4(p2, 0.0f); ws.I(id, at) = make_float4(0, 0, 0, 0); ws.W(id, at) = make_float4(0, 0, 0, 0); }
    }
}
// carry_warmstart_data (:129-174): the pair's previous solver manifolds (slot s + its aux slots, untouched so far) -> the new clusters
__device__ void carry_to_clusters(DevWorld &w, int s, int prev_ncl, const CmWs &ws, const V3 *cn1, const int *cnp, int ncl) {
    const float match_eps_sq = w.prm.prediction * w.prm.prediction;
    for (int a = 0; a < prev_ncl; ++a) {
        const int sa = sm_slot(w, s, a);
        const V3 pn1 = v3(w.p_ln1[sa]);
        const int np = w.p_npts[sa];
        for (int i = 0; i < np; ++i) {
            const float4 pimp = PT(w.pt_imp, i, sa);
            if (!data_has_warmstart(pimp)) continue;
            const V3 pp = v3(PT(w.pt_lp1d, i, sa));
            int bt = -1, bp = -1; float best = match_eps_sq;
            for (int t = 0; t < ncl; ++t) {
                if (dot(cn1[t], pn1) < RP_COS_MERGE_ANGLE) continue;
                for (int k = 0; k < cnp[t]; ++k) {
                    if (data_has_warmstart(ws.I(t, k))) continue;
                    const float d = len2(v3(ws.A(t, k)) - pp);
                    if (d < best) { best = d; bt = t; bp = k; }
                }
            }
            if (bt >= 0) { ws.I(bt, bp) = pimp; ws.W(bt, bp) = PT(w.pt_wst, i, sa); }
        }
    }
}
// manifold_reduction::reduce_manifold_naive over a cluster's points in the workspace (see reduce_manifold)
__device__ void reduce_cluster(const CmWs &ws, int c, int n, V3 ln1, int sel[4], int &nsel, float prediction) {
    if (n <= 4) return;
    sel[0] = sel[1] = sel[2] = sel[3] = -1;
    float deepest = FLT_MAX;
    for (int i = 0; i < n; ++i) if (ws.A(c, i).w < deepest) { deepest = ws.A(c, i).w; sel[0] = i; }
    if (sel[0] < 0) { nsel = 0; return; }
    V3 a = v3(ws.A(c, sel[0]));
    float furthest = -FLT_MAX;
    for (int i = 0; i < n; ++i) {
        float d = len2(v3(ws.A(c, i)) - a);
        if (i != sel[0] && ws.A(c, i).w <= prediction && d > furthest) { furthest = d; sel[1] = i; }
    }
    if (sel[1] < 0) { nsel = 1; return; }
    V3 b = v3(ws.A(c, sel[1]));
    if (a.x == b.x && a.y == b.y && a.z == b.z) { nsel = 1; return; }
    V3 tangent = cross(b - a, ln1);
    float mind = FLT_MAX, maxd = -FLT_MAX;
    for (int i = 0; i < n; ++i) {
        if (i == sel[0] || i == sel[1] || ws.A(c, i).w > prediction) continue;
        float d = dot(v3(ws.A(c, i)) - a, tangent);
        if (d < mind) { mind = d; sel[2] = i; }
        if (d > maxd) { maxd = d; sel[3] = i; }
    }
    if (sel[2] < 0) nsel = 2; else if (sel[2] == sel[3]) nsel = 3; else nsel = 4;
}

// one aux slot for cluster `k` of parent slot s (the narrow phase runs behind the broad phase: nobody else pops the free stack; slots
// freed in this launch wait in free_pending until k_cm_finish)
__device__ int aux_slot_alloc(DevWorld &w, int s, int k) {
    int t = atomicSub(&w.flags[FL_FREE_TOP], 1), slot;
    if (t > 0) slot = w.free_stack[t - 1];
    else { atomicAdd(&w.flags[FL_FREE_TOP], 1); slot = atomicAdd(&w.flags[FL_POOL_TOP], 1); }
    if (slot >= w.pool_cap) { atomicOr(&w.flags[FL_OVERFLOW], RP_OVF_POOL); return -1; }
    w.p_c1[slot] = w.p_c1[s]; w.p_c2[slot] = w.p_c2[s]; w.p_rb[slot] = w.p_rb[s];
    w.p_color[slot] = RP_COLOR_OVERFLOW; w.p_colorb[slot] = make_int2(-1, -1);
    w.p_nsc[slot] = 0; w.p_npts[slot] = 0; w.p_pflags[slot] = RP_PF_AUX; w.p_reldom[slot] = 0; w.p_conspos[slot] = -1; w.p_hint_seq[slot] = 0;
    w.p_stamp[slot] = w.p_stamp[s];
    w.p_aux[slot] = make_int4(s, k, -1, 0); w.p_sub[slot] = make_int2(-1, -1);
    w.flags[FL_LAYOUT_DIRTY] = 1;
    return slot;
}

// The full update of a pair with a composite collider — oracle: process_composite_pair.  `lane_ws`: this thread's workspace.
template <bool CONVEX> __device__ void composite_pair_update(DevWorld &w, int s, int c1, int c2, Pose pc1, Pose pc2, Pose pos12, float *np_lds, const CmWs &ws) {
    const float prediction = w.prm.prediction;
    const int rb1 = w.c_parent[c1], rb2 = w.c_parent[c2];
    if (w.has_sensors && pair_is_sensor(w, c1, c2)) { // intersection_test_composite_shape_shape: any candidate sub-shape pair intersects
        const int was = (w.p_pflags[s] & RP_PF_INTERSECTING) ? 1 : 0;
        int now = 0;
        if (!(rb1 == rb2 && rb1 >= 0)) {
            bool ovf; const int n = comp_candidates(w, c1, c2, pos12, 0.0f, ws, ovf);
            for (int q = 0; q < n && !now; ++q) {
                const int2 cd = ws.cand(q);
                const SubShape a = co_sub(w, c1, cd.x < 0 ? 0 : cd.x), b = co_sub(w, c2, cd.y < 0 ? 0 : cd.y);
                const Pose wa = a.has_pose ? pose_mul(pc1, a.pos) : pc1, wb = b.has_pose ? pose_mul(pc2, b.pos) : pc2;
                float4 ha = a.he, hb = b.he;
                SmShape sa = sm_shape_of(w, a.sh, ha, a.bd), sb = sm_shape_of(w, b.sh, hb, b.bd);
                for (int k = 0; k < 3; ++k) { sa.tri[k] = a.tri[k]; sb.tri[k] = b.tri[k]; }
                now = sm_intersects(sa, sb, pose_inv_mul(wa, wb)) ? 1 : 0; // (every sub-shape pair through the support-map test: see DESIGN.md section 5)
            }
        }
        w.p_npts[s] = 0; w.p_nsc[s] = 0; w.p_pflags[s] = (w.p_pflags[s] & ~(RP_PF_RECYCLE | RP_PF_INTERSECTING)) | (now ? RP_PF_INTERSECTING : 0);
        if (w.p_aux[s].w > 0 || w.p_aux[s].x >= 0) aux_free_all(w, s, true);
        if (was != now && pair_wants_collision_events(w, c1, c2)) push_collision_event(w, c1, c2, now, RP_COLLISION_EVENT_SENSOR, cur_step(w));
        atomicAdd(&w.flags[FL_FULL_UPDATES], 1);
        return;
    }
    bool ovf;
    const int ncand = joints_disable_contacts(w, rb1, rb2) ? 0 : comp_candidates(w, c1, c2, pos12, prediction, ws, ovf);
    const int prev_ncl = w.p_aux[s].w;
    if (ncand <= 1) {
        // ---- one manifold (or none): the plain path, through pair_full_update with the candidate's shapes and poses ----
        SubSel sub;
        const int2 cd = ncand ? ws.cand(0) : make_int2(-1, -1);
        const SubShape a = co_sub(w, c1, cd.x < 0 ? 0 : cd.x), b = co_sub(w, c2, cd.y < 0 ? 0 : cd.y);
        sub.sh1 = a.sh; sub.sh2 = b.sh; sub.he1 = a.he; sub.he2 = b.he; sub.bd1 = a.bd; sub.bd2 = b.bd;
        for (int k = 0; k < 3; ++k) { sub.tri1[k] = a.tri[k]; sub.tri2[k] = b.tri[k]; }
        sub.wp1 = a.has_pose ? pose_mul(pc1, a.pos) : pc1; sub.wp2 = b.has_pose ? pose_mul(pc2, b.pos) : pc2;
        Pose rel = a.has_pose ? pose_inv_mul(a.pos, pos12) : pos12;
        if (b.has_pose) rel = pose_mul(rel, b.pos);
        sub.rel = rel; sub.has_pos1 = a.has_pose; sub.pos1 = a.pos;
        const int2 ps = w.p_sub[s];
        sub.none = ncand == 0;
        sub.fresh = prev_ncl > 0 || ps.x != cd.x || ps.y != cd.y || ncand == 0;
        sub.prev_ncl = ncand == 1 ? prev_ncl : 0;
        pair_full_update<CONVEX>(w, s, c1, c2, pc1, pc2, pos12, np_lds, &sub);
        if (prev_ncl > 0 || w.p_aux[s].x >= 0) aux_free_all(w, s, true);
        w.p_aux[s] = make_int4(-1, -1, -1, 0);
        w.p_sub[s] = cd;
        return;
    }
    // ---- several manifolds: solver clusters ----
    const int had = w.p_nsc[s] > 0;
    V3 cn1[RP_MAX_CLUSTERS], cn2[RP_MAX_CLUSTERS]; int cnp[RP_MAX_CLUSTERS]; int ncl = 0;
    const float dedup_eps = prediction * 0.25f, dedup_eps_sq = dedup_eps * dedup_eps;
    LocalManifold m; m.bind(np_lds);
    for (int q = 0; q < ncand; ++q) {
        const int2 cd = ws.cand(q);
        const SubShape a = co_sub(w, c1, cd.x < 0 ? 0 : cd.x), b = co_sub(w, c2, cd.y < 0 ? 0 : cd.y);
        Pose rel = a.has_pose ? pose_inv_mul(a.pos, pos12) : pos12;
        if (b.has_pose) rel = pose_mul(rel, b.pos);
        m.n = 0; m.ln1 = v3(0, 0, 0); m.ln2 = v3(0, 0, 0);
        dispatch_manifold<CONVEX>(w, a.sh, a.he, a.bd, a.tri, b.sh, b.he, b.bd, b.tri, rel, prediction, m);
        if (m.n == 0) continue;
        const V3 n1 = a.has_pose ? qrot(a.pos.r, m.ln1) : m.ln1, n2 = b.has_pose ? qrot(b.pos.r, m.ln2) : m.ln2;
        cluster_add_manifold(ws, cn1, cn2, cnp, ncl, m, n1, n2, a, b, dedup_eps_sq);
    }
    carry_to_clusters(w, s, prev_ncl, ws, cn1, cnp, ncl);

    const float4 mat1 = w.c_mat[c1], mat2 = w.c_mat[c2];
    const int2 ru1 = w.c_rules[c1], ru2 = w.c_rules[c2];
    const float friction = combine_coeff(mat1.x, mat2.x, ru1.x, ru2.x), restitution = combine_coeff(mat1.y, mat2.y, ru1.y, ru2.y);
    const int rel_dom = effective_dominance(w, rb1) - effective_dominance(w, rb2);
    const bool has1 = rb1 >= 0 && rel_dom <= 0, has2 = rb2 >= 0 && rel_dom >= 0;
    Pose com1, com2; com1.r = q4(0, 0, 0, 1); com1.t = v3(0, 0, 0); com2 = com1;
    V3 lv1 = v3(0, 0, 0), av1 = lv1, wc1 = lv1, lv2 = lv1, av2 = lv1, wc2 = lv1;
    if (rb1 >= 0) { lv1 = v3(w.b_linvel[rb1]); av1 = v3(w.b_angvel[rb1]); wc1 = v3(w.b_wcom[rb1]); }
    if (rb2 >= 0) { lv2 = v3(w.b_linvel[rb2]); av2 = v3(w.b_angvel[rb2]); wc2 = v3(w.b_wcom[rb2]); }
    if (has1) { Pose bp; bp.r = q4(w.b_rot[rb1]); bp.t = v3(w.b_pos[rb1]); com1.r = bp.r; com1.t = pose_tp(bp, v3(w.b_lcom_invm[rb1])); }
    if (has2) { Pose bp; bp.r = q4(w.b_rot[rb2]); bp.t = v3(w.b_pos[rb2]); com2.r = bp.r; com2.t = pose_tp(bp, v3(w.b_lcom_invm[rb2])); }

    // solver contacts of every cluster (pair_update.rs:404-577 with subshape_pos = None): selection kept per cluster
    int csel[RP_MAX_CLUSTERS][4], cnsc[RP_MAX_CLUSTERS];
    for (int c = 0; c < ncl; ++c) {
        const V3 normal = qrot(pc1.r, cn1[c]);
        int sel[4] = {0, 1, 2, 3};
        int nsel = cnp[c] < 4 ? cnp[c] : 4;
        reduce_cluster(ws, c, cnp[c], cn1[c], sel, nsel, prediction);
        if (nsel > 1) { // pair_update.rs:430-457
            V3 b0, b1; orthonormal_basis(cn1[c], b0, b1);
            float k0[4], k1[4]; int ks[4];
            for (int i = 0; i < nsel; ++i) { V3 lp = v3(ws.A(c, sel[i])); k0[i] = dot(lp, b0); k1[i] = dot(lp, b1); ks[i] = sel[i]; }
            for (int i = 1; i < nsel; ++i) {
                float a0 = k0[i], a1 = k1[i]; int as = ks[i]; int j = i;
                while (j > 0 && (k0[j - 1] > a0 || (k0[j - 1] == a0 && k1[j - 1] > a1))) { k0[j] = k0[j - 1]; k1[j] = k1[j - 1]; ks[j] = ks[j - 1]; j--; }
                k0[j] = a0; k1[j] = a1; ks[j] = as;
            }
            for (int i = 0; i < nsel; ++i) sel[i] = ks[i];
        }
        int nsc = 0;
        for (int q = 0; q < nsel; ++q) { // :459-498: the distance / approach test
            const int cid = sel[q];
            const float eff_dist = ws.A(c, cid).w;
            const V3 wpt1 = pose_tp(pc1, v3(ws.A(c, cid))), wpt2 = pose_tp(pc2, v3(ws.B(c, cid)));
            bool keep = eff_dist < prediction;
            if (!keep) {
                V3 vel1 = rb1 >= 0 ? lv1 + cross(av1, wpt1 - wc1) : v3(0, 0, 0);
                V3 vel2 = rb2 >= 0 ? lv2 + cross(av2, wpt2 - wc2) : v3(0, 0, 0);
                keep = eff_dist + dot(vel2 - vel1, normal) * w.prm.p.dt < prediction;
            }
            if (keep) csel[c][nsc++] = cid;
        }
        cnsc[c] = nsc;
    }
    // the clusters with solver contacts first (stable): the pair's own slot holds the first one, which takes the pair's colour
    int order[RP_MAX_CLUSTERS], no = 0;
    for (int c = 0; c < ncl; ++c) if (cnsc[c] > 0) order[no++] = c;
    for (int c = 0; c < ncl; ++c) if (cnsc[c] == 0) order[no++] = c;
    // aux slots follow the cluster count
    int4 aux = w.p_aux[s];
    int auxs[3] = {aux.x, aux.y, aux.z};
    for (int k = 1; k < RP_MAX_CLUSTERS; ++k) {
        if (k < ncl && auxs[k - 1] < 0) auxs[k - 1] = aux_slot_alloc(w, s, k);
        else if (k >= ncl && auxs[k - 1] >= 0) { aux_slot_free(w, auxs[k - 1], true); auxs[k - 1] = -1; }
    }
    for (int k = 0; k < ncl; ++k) {
        const int slot = k == 0 ? s : auxs[k - 1];
        if (slot < 0) continue; // (pool exhausted: RP_OVF_POOL is raised, the step is replayed on a larger pool)
        const int c = order[k];
        const V3 normal = qrot(pc1.r, cn1[c]);
        const int old_nsc = w.p_nsc[slot];
        w.p_ln1[slot] = f4(cn1[c], 0.0f); w.p_ln2[slot] = f4(cn2[c], 0.0f);
        w.p_normal[slot] = f4(normal, friction); w.p_reldom[slot] = rel_dom;
        if (k > 0) w.p_misc[slot] = make_float4(restitution, 0.0f, 0.0f, 0.0f); // (slot 0: the pair's recycle state lives there — pair_update_finish)
        // kept: the points the solver contacts name (contact id = position), then the other points that carry warm-start data
        int np = 0;
        for (int j = 0; j < cnsc[c]; ++j) {
            const int cid = csel[c][j];
            const float4 A = ws.A(c, cid), B = ws.B(c, cid);
            PT(w.pt_lp1d, np, slot) = A; PT(w.pt_lp2f, np, slot) = f4(v3(B), __uint_as_float(RP_FID_UNKNOWN | (RP_FID_UNKNOWN << 16)));
            PT(w.pt_imp, np, slot) = ws.I(c, cid); PT(w.pt_wst, np, slot) = ws.W(c, cid);
            // :536-577 anchors localised, lever arms frozen
            const float eff_dist = A.w;
            const V3 wpt1 = pose_tp(pc1, v3(A)), wpt2 = pose_tp(pc2, v3(B));
            const float shift = dot(wpt2 - wpt1, normal) - eff_dist;
            const V3 p1 = wpt1 + normal * shift;
            const V3 point = (p1 + wpt2) * 0.5f;
            PT(w.pt_dp1, np, slot) = f4(has1 ? point - com1.t : point, 0.0f);
            PT(w.pt_dp2, np, slot) = f4(has2 ? point - com2.t : point, 0.0f);
            PT(w.sc_a1, j, slot) = f4(has1 ? pose_itp(com1, p1) : p1, eff_dist);
            PT(w.sc_a2, j, slot) = f4(has2 ? pose_itp(com2, wpt2) : wpt2, __int_as_float(np));
            ++np;
        }
        for (int i = 0; i < cnp[c] && np < RP_MAX_PTS; ++i) {
            bool named = false;
            for (int j = 0; j < cnsc[c]; ++j) named |= csel[c][j] == i;
            if (named || !data_has_warmstart(ws.I(c, i))) continue;
            PT(w.pt_lp1d, np, slot) = ws.A(c, i); PT(w.pt_lp2f, np, slot) = f4(v3(ws.B(c, i)), __uint_as_float(RP_FID_UNKNOWN | (RP_FID_UNKNOWN << 16)));
            PT(w.pt_imp, np, slot) = ws.I(c, i); PT(w.pt_wst, np, slot) = ws.W(c, i);
            ++np;
        }
        w.p_npts[slot] = np; w.p_nsc[slot] = cnsc[c];
        if (k > 0 && (old_nsc > 0) != (cnsc[c] > 0)) w.flags[FL_LAYOUT_DIRTY] = 1; // an aux manifold entered or left the solver's set
    }
    w.p_aux[s] = make_int4(auxs[0], auxs[1], auxs[2], ncl);
    w.p_sub[s] = make_int2(-2, -2);
    const int nsc0 = ncl > 0 ? cnsc[order[0]] : 0;
    if (ncl == 0) { w.p_npts[s] = 0; w.p_nsc[s] = 0; }
    // recycle state, hint, transition: the pair-level tail of pair_full_update
    pair_update_finish(w, s, c1, c2, rb1, rb2, w.c_shape[c1], w.c_he[c1], w.c_shape[c2], w.c_he[c2], mat1.w, mat2.w, pc1, pc2, pos12, nsc0, had, false, restitution);
}

template <bool CONVEX> __global__ void __launch_bounds__(NP_THREADS) k_np_composite(DevWorld w) {
    __shared__ __align__(16) float np_lds[NP_THREADS * NP_LDS_DWORDS];
    if (collision_done(w)) return;
    int count = w.flags[FL_NP_COUNT];
    if (count > w.pool_cap) count = w.pool_cap;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    CmWs ws; ws.base = w.cm_ws + tid; ws.stride = w.cm_ws_threads;
    for (int i = tid; i < count; i += stride) {
        const int s = w.np_list[i];
        const int c1 = w.p_c1[s], c2 = w.p_c2[s];
        if (!shape_is_composite(w.c_shape[c1]) && !shape_is_composite(w.c_shape[c2])) continue; // k_np_update took it
        Pose pc1, pc2;
        pc1.r = q4(w.c_rot[c1]); pc1.t = v3(w.c_pos[c1]);
        pc2.r = q4(w.c_rot[c2]); pc2.t = v3(w.c_pos[c2]);
        composite_pair_update<CONVEX>(w, s, c1, c2, pc1, pc2, pose_inv_mul(pc1, pc2), np_lds, ws);
    }
}
// the slots freed by k_np_composite go onto the free stack (one workgroup; see aux_slot_alloc)
__global__ void k_cm_finish(DevWorld w) {
    if (collision_done(w)) return;
    const int nfreed = w.flags[FL_BP_NFREED], ftop = w.flags[FL_FREE_TOP];
    for (int k = threadIdx.x; k < nfreed; k += blockDim.x) w.free_stack[ftop + k] = w.free_pending[k];
    __syncthreads();
    if (threadIdx.x == 0 && nfreed > 0) { w.flags[FL_FREE_TOP] = ftop + nfreed; w.flags[FL_BP_NFREED] = 0; }
}
