/* ro_composite.h — composite shapes as ONE collider and contact clustering (TEST INFRASTRUCTURE, part of the oracle: included by
 * rapier_oracle.c, never by the product).
 *
 *   ColliderBuilder::compound   /root/reference/src/geometry/collider.rs:711   (parry Compound: parts = (pose, convex shape))
 *   ColliderBuilder::trimesh    collider.rs:944                                (parry TriMesh, no flags: plain triangles)
 *   ColliderBuilder::heightfield collider.rs:1089                              (parry HeightField: served as a triangle mesh, see ro_add_heightfield)
 *   cluster_manifolds_for_solver / carry_warmstart_data   /root/reference/src/geometry/contact_clustering.rs:33, :129
 *   the gate `use_clusters = contact_clustering && pair.manifolds.len() > 1`   /root/reference/src/geometry/narrow_phase/pair_update.rs:350
 *   solver manifolds 2+ of a pair go to the overflow colour                    /root/reference/src/geometry/narrow_phase/solver_graph.rs:534-547
 *
 * parry itself is not under /root/reference (Cargo dependency parry3d), so what it does for these shapes is restated from its published
 * algorithm and PARITY WITH PARRY IS UNPINNED, like every other manifold of this oracle:
 *   contact_manifolds_composite_shape_shape / _composite_shape_composite_shape / _trimesh_shape: one manifold per sub-shape (pair)
 *   whose AABB meets the other shape's AABB loosened by the prediction distance, each computed by the convex-convex dispatcher with
 *   the part's pose folded into the relative pose; `manifold.subshape_pos*` = the part's pose.
 * Deliberate simplifications (all stated in DESIGN.md section 5):
 *   * candidates are visited in ascending sub-shape index order (parry: BVH traversal order, which nothing outside parry reproduces);
 *   * the candidate set is recomputed from the tight loosened AABB every step (parry's TriMesh workspace keeps a fattened AABB and
 *     re-collects only when the shape leaves it: history-dependent);
 *   * sub-manifolds are persistent only while the pair has exactly ONE candidate (then the pair's manifold 0 is that sub-manifold and
 *     is tracked like a primitive pair's); with several candidates they are recomputed from scratch each step — the solver sees
 *     clusters then, whose warm-start data never comes from the sub-manifolds (contact_clustering.rs:93-95);
 *   * a triangle meets every shape through the support-map path (GJK / EPA + polygonal features; parry has a SAT special case for
 *     cuboid-triangle);
 *   * at most RO_MAX_CLUSTERS clusters per pair, RO_CLUSTER_PTS points per cluster while it is built, RO_MAX_SUBPAIRS candidates.
 */
#ifndef RO_COMPOSITE_H
#define RO_COMPOSITE_H

typedef struct { pose pos; Collider prim; Aabb aabb; } RoPart; /* pose / local AABB in the (recentred) composite frame; prim: shape fields only */
typedef struct RoComposite {
    int kind;                 /* RO_SHAPE_COMPOUND | RO_SHAPE_TRIMESH */
    int n;                    /* parts | triangles */
    RoPart *parts;            /* compound */
    v3 *verts; int nv; int *tris; Aabb *tri_aabb; /* triangle mesh (vertices recentred) */
    v3 centre, half;          /* the local AABB: its centre rides in the collider's pose, `half` is the collider's he */
} RoComposite;

static Aabb aabb_of_points(const v3 *p, int n) {
    Aabb a; a.mins = p[0]; a.maxs = p[0];
    for (int i = 1; i < n; ++i) {
        a.mins = V3(ro_minf(a.mins.x, p[i].x), ro_minf(a.mins.y, p[i].y), ro_minf(a.mins.z, p[i].z));
        a.maxs = V3(ro_maxf(a.maxs.x, p[i].x), ro_maxf(a.maxs.y, p[i].y), ro_maxf(a.maxs.z, p[i].z));
    }
    return a;
}
static Aabb aabb_merge(Aabb a, Aabb b) {
    Aabb r; r.mins = V3(ro_minf(a.mins.x, b.mins.x), ro_minf(a.mins.y, b.mins.y), ro_minf(a.mins.z, b.mins.z));
    r.maxs = V3(ro_maxf(a.maxs.x, b.maxs.x), ro_maxf(a.maxs.y, b.maxs.y), ro_maxf(a.maxs.z, b.maxs.z));
    return r;
}
static Aabb aabb_loosened(Aabb a, float l) { v3 d = V3(l, l, l); a.mins = vsub(a.mins, d); a.maxs = vadd(a.maxs, d); return a; }
/* Shape::compute_aabb(pos) of a primitive (the collider's own AABB rule at an arbitrary pose) */
static Aabb prim_aabb_at(const Collider *prim, pose at) {
    if (prim->shape == RO_SHAPE_TRIANGLE) { v3 q[3] = {pose_tp(at, prim->tri[0]), pose_tp(at, prim->tri[1]), pose_tp(at, prim->tri[2])}; return aabb_of_points(q, 3); }
    Collider c = *prim; c.pos = at;
    return collider_collision_aabb(&c, 0.0f);
}

static void comp_set_prim(Collider *c, const ro_world *w, const ro_collider_desc *d) { /* the shape fields of ro_add_collider */
    memset(c, 0, sizeof(*c));
    c->parent = -1;
    c->shape = ro_core_shape(d->shape); c->border = d->shape >= RO_SHAPE_ROUND_CUBOID && d->shape <= RO_SHAPE_ROUND_CONVEX_POLYHEDRON ? d->border_radius : 0.0f;
    c->he = V3(d->half_extents[0], d->half_extents[1], d->half_extents[2]);
    c->radius = d->half_extents[0];
    if (c->shape == RO_SHAPE_CAPSULE) { c->radius = d->half_extents[1]; c->axis = (int)d->half_extents[2]; if (c->axis < 0 || c->axis > 2) c->axis = 1; }
    if (c->shape == RO_SHAPE_CYLINDER || c->shape == RO_SHAPE_CONE) { c->radius = d->half_extents[1]; c->he = V3(c->radius, d->half_extents[0], c->radius); c->axis = 1; }
    if (c->shape == RO_SHAPE_CONVEX_POLYHEDRON) { c->poly = w->polys[(int)d->half_extents[0]]; c->he = c->poly->half; c->radius = 0.0f; c->axis = 1; }
}
static int32_t comp_register(ro_world *w, RoComposite *C) {
    w->comps = (RoComposite **)realloc(w->comps, sizeof(RoComposite *) * (size_t)(w->ncomps + 1));
    w->comps[w->ncomps] = C;
    return w->ncomps++;
}
int32_t ro_add_compound(ro_world *w, int32_t n_parts, const ro_collider_desc *parts) {
    if (n_parts < 1 || !parts) return -1;
    for (int i = 0; i < n_parts; ++i) {
        int sh = parts[i].shape;
        if (sh < RO_SHAPE_BALL || sh > RO_SHAPE_ROUND_CONVEX_POLYHEDRON || sh == RO_SHAPE_HALFSPACE) return -1; /* Compound::new: no composite / unbounded part */
        if (ro_core_shape(sh) == RO_SHAPE_CONVEX_POLYHEDRON && !((int)parts[i].half_extents[0] >= 0 && (int)parts[i].half_extents[0] < w->npolys)) return -1;
    }
    RoComposite *C = (RoComposite *)calloc(1, sizeof(RoComposite));
    C->kind = RO_SHAPE_COMPOUND; C->n = n_parts;
    C->parts = (RoPart *)calloc((size_t)n_parts, sizeof(RoPart));
    Aabb box;
    for (int i = 0; i < n_parts; ++i) {
        RoPart *P = &C->parts[i];
        comp_set_prim(&P->prim, w, &parts[i]);
        P->pos.t = V3(parts[i].translation[0], parts[i].translation[1], parts[i].translation[2]);
        P->pos.r = qnormalize(Q(parts[i].rotation[0], parts[i].rotation[1], parts[i].rotation[2], parts[i].rotation[3]));
        if (P->prim.shape == RO_SHAPE_CONVEX_POLYHEDRON) P->pos.t = vadd(qrot(P->pos.r, P->prim.poly->centre), P->pos.t);
        P->aabb = prim_aabb_at(&P->prim, P->pos);
        box = i == 0 ? P->aabb : aabb_merge(box, P->aabb);
    }
    C->centre = vmul(vadd(box.mins, box.maxs), 0.5f); C->half = vmul(vsub(box.maxs, box.mins), 0.5f);
    for (int i = 0; i < n_parts; ++i) { /* recentre on the local AABB (the centre is folded into the collider's pose, like a polyhedron's) */
        RoPart *P = &C->parts[i];
        P->pos.t = vsub(P->pos.t, C->centre);
        P->aabb = prim_aabb_at(&P->prim, P->pos);
    }
    return comp_register(w, C);
}
int32_t ro_add_trimesh(ro_world *w, int32_t nv, const float *xyz, int32_t nt, const uint32_t *idx) {
    if (nv < 3 || nt < 1 || !xyz || !idx) return -1;
    for (int i = 0; i < 3 * nt; ++i) if (idx[i] >= (uint32_t)nv) return -1;
    RoComposite *C = (RoComposite *)calloc(1, sizeof(RoComposite));
    C->kind = RO_SHAPE_TRIMESH; C->n = nt; C->nv = nv;
    C->verts = (v3 *)malloc(sizeof(v3) * (size_t)nv); C->tris = (int *)malloc(sizeof(int) * 3 * (size_t)nt); C->tri_aabb = (Aabb *)malloc(sizeof(Aabb) * (size_t)nt);
    for (int i = 0; i < nv; ++i) C->verts[i] = V3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    Aabb box = aabb_of_points(C->verts, nv);
    C->centre = vmul(vadd(box.mins, box.maxs), 0.5f); C->half = vmul(vsub(box.maxs, box.mins), 0.5f);
    for (int i = 0; i < nv; ++i) C->verts[i] = vsub(C->verts[i], C->centre);
    for (int t = 0; t < nt; ++t) {
        for (int k = 0; k < 3; ++k) C->tris[3 * t + k] = (int)idx[3 * t + k];
        v3 q[3] = {C->verts[C->tris[3 * t]], C->verts[C->tris[3 * t + 1]], C->verts[C->tris[3 * t + 2]]};
        C->tri_aabb[t] = aabb_of_points(q, 3);
    }
    return comp_register(w, C);
}
/* HeightField (heightfield3.rs): nrows x ncols heights over the unit square centred on the origin, scaled by `scale`; cell (r, c)
 * spans x in [c, c + 1] / (ncols - 1) - 0.5, z in [r, r + 1] / (nrows - 1) - 0.5 and is cut along its (r, c) -> (r + 1, c + 1)
 * diagonal into the triangles (p00, p10, p11) and (p00, p11, p01) — parry's default cell status.  Served by the triangle-mesh path. */
int32_t ro_add_heightfield(ro_world *w, int32_t nrows, int32_t ncols, const float *heights, const float scale[3]) {
    if (nrows < 2 || ncols < 2 || !heights || !scale) return -1;
    const int nv = nrows * ncols, nt = 2 * (nrows - 1) * (ncols - 1);
    float *xyz = (float *)malloc(sizeof(float) * 3 * (size_t)nv); uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * 3 * (size_t)nt);
    for (int r = 0; r < nrows; ++r) for (int c = 0; c < ncols; ++c) {
        const int i = r * ncols + c;
        xyz[3 * i] = ((float)c / (float)(ncols - 1) - 0.5f) * scale[0]; xyz[3 * i + 1] = heights[i] * scale[1]; xyz[3 * i + 2] = ((float)r / (float)(nrows - 1) - 0.5f) * scale[2];
    }
    int t = 0;
    for (int r = 0; r + 1 < nrows; ++r) for (int c = 0; c + 1 < ncols; ++c) {
        const uint32_t p00 = (uint32_t)(r * ncols + c), p01 = p00 + 1, p10 = p00 + (uint32_t)ncols, p11 = p10 + 1;
        idx[3 * t] = p00; idx[3 * t + 1] = p10; idx[3 * t + 2] = p11; ++t;
        idx[3 * t] = p00; idx[3 * t + 1] = p11; idx[3 * t + 2] = p01; ++t;
    }
    const int32_t id = ro_add_trimesh(w, nv, xyz, nt, idx);
    free(xyz); free(idx);
    return id;
}
static v3 comp_half(const RoComposite *C) { return C->half; }
static v3 comp_centre(const RoComposite *C) { return C->centre; }
static void comp_free(RoComposite *C) { if (!C) return; free(C->parts); free(C->verts); free(C->tris); free(C->tri_aabb); free(C); }

/* ---- sub-shapes of a collider ------------------------------------------------------------------------------------------------------ */
static int co_is_composite(const Collider *c) { return c->shape == RO_SHAPE_COMPOUND || c->shape == RO_SHAPE_TRIMESH; }
static int co_num_subs(const Collider *c) { return co_is_composite(c) ? c->comp->n : 1; }
static Aabb co_sub_aabb(const Collider *c, int i) { return c->shape == RO_SHAPE_COMPOUND ? c->comp->parts[i].aabb : c->comp->tri_aabb[i]; }
/* sub-shape i as a primitive + its pose in the collider's frame (has_pose = 0: identity — a triangle's vertices are in mesh space) */
static void co_sub(const Collider *c, int i, Collider *prim, pose *pos, int *has_pose) {
    *pos = pose_ident(); *has_pose = 0;
    if (c->shape == RO_SHAPE_COMPOUND) { *prim = c->comp->parts[i].prim; *pos = c->comp->parts[i].pos; *has_pose = 1; }
    else if (c->shape == RO_SHAPE_TRIMESH) {
        memset(prim, 0, sizeof(*prim)); prim->parent = -1; prim->shape = RO_SHAPE_TRIANGLE; prim->axis = 1;
        const RoComposite *C = c->comp;
        for (int k = 0; k < 3; ++k) prim->tri[k] = C->verts[C->tris[3 * i + k]];
    } else *prim = *c;
}
/* Compound::ccd_thickness (parry): the thinnest part's (Shape::ccd_thickness: ball / capsule radius, smallest half extent; + a round part's border) */
static float comp_ccd_thickness(const Collider *c) {
    float th = FLT_MAX;
    for (int k = 0; k < c->comp->n; ++k) {
        const Collider *q = &c->comp->parts[k].prim;
        float tq = q->shape == RO_SHAPE_BALL ? q->radius : q->shape == RO_SHAPE_CAPSULE ? q->radius : ro_minf(q->he.x, ro_minf(q->he.y, q->he.z));
        if (q->border > 0.0f) tq = tq + q->border;
        th = ro_minf(th, tq);
    }
    return th;
}
/* MassProperties of a compound = the sum of its parts' (MassProperties::from_compound); a triangle mesh on a fixed body weighs nothing */
static void comp_mass_props(const Collider *c, float density, ro_mp *out) {
    memset(out, 0, sizeof(*out)); out->frame[3] = 1.0f;
    if (c->shape != RO_SHAPE_COMPOUND) return;
    for (int i = 0; i < c->comp->n; ++i) {
        const RoPart *P = &c->comp->parts[i];
        ro_mp mp; v3 pi;
        shape_mass_props(&P->prim, density, &mp.mass, &pi, mp.frame, mp.com);
        mp.pi[0] = pi.x; mp.pi[1] = pi.y; mp.pi[2] = pi.z;
        const float t[3] = {P->pos.t.x, P->pos.t.y, P->pos.t.z}, q[4] = {P->pos.r.x, P->pos.r.y, P->pos.r.z, P->pos.r.w};
        ro_mp_transform(&mp, t, q);
        if (i == 0) *out = mp; else ro_mp_add(out, &mp);
    }
}

/* candidate sub-shape pairs of (co1, co2), ascending (i1, i2); returns the count (<= cap), *overflow = more existed */
static int comp_candidates(const Collider *co1, const Collider *co2, pose pos12, float prediction, int (*out)[2], int cap, int *overflow) {
    int n = 0; *overflow = 0;
    const int c1 = co_is_composite(co1), c2 = co_is_composite(co2);
    if (c1 && !c2) {
        const Aabb b = aabb_loosened(prim_aabb_at(co2, pos12), prediction); /* shape2.compute_aabb(pos12).loosened(prediction) */
        for (int i = 0; i < co1->comp->n; ++i) { Aabb a = co_sub_aabb(co1, i); if (aabb_intersects(&a, &b)) { if (n < cap) { out[n][0] = i; out[n][1] = -1; ++n; } else *overflow = 1; } }
    } else if (!c1 && c2) {
        const Aabb a = aabb_loosened(prim_aabb_at(co1, pose_inv(pos12)), prediction);
        for (int j = 0; j < co2->comp->n; ++j) { Aabb b = co_sub_aabb(co2, j); if (aabb_intersects(&a, &b)) { if (n < cap) { out[n][0] = -1; out[n][1] = j; ++n; } else *overflow = 1; } }
    } else {
        /* composite x composite: the parts of 1 that reach the box of 2, each against the parts of 2 it reaches */
        Collider whole2 = *co2; whole2.pos = pos12;
        const Aabb box2 = aabb_loosened(collider_collision_aabb(&whole2, 0.0f), prediction);
        const pose pos21 = pose_inv(pos12);
        for (int i = 0; i < co1->comp->n; ++i) {
            Aabb a = co_sub_aabb(co1, i);
            if (!aabb_intersects(&a, &box2)) continue;
            Collider prim; pose ppos; int hp; co_sub(co1, i, &prim, &ppos, &hp);
            const Aabb a2 = aabb_loosened(prim_aabb_at(&prim, hp ? pose_mul(pos21, ppos) : pos21), prediction);
            for (int j = 0; j < co2->comp->n; ++j) { Aabb b = co_sub_aabb(co2, j); if (aabb_intersects(&a2, &b)) { if (n < cap) { out[n][0] = i; out[n][1] = j; ++n; } else *overflow = 1; } }
        }
    }
    return n;
}

/* ---- contact clustering (contact_clustering.rs) ---------------------------------------------------------------------------------------- */
#define RO_COS_MERGE_ANGLE 0.996f
typedef struct { v3 n1, n2; int np; TrackedContact pts[RO_CLUSTER_PTS]; } ClusterTmp;
static int data_has_warmstart(const ContactData *d) { return d->impulse != 0.0f || d->warmstart_impulse != 0.0f; }
/* cluster_manifolds_for_solver :49-122 for ONE sub-manifold (already known to hold points); n1 / n2: its normals in the collider frames */
static void cluster_add_manifold(ClusterTmp *cl, int *ncl, const Manifold *m, v3 n1, v3 n2, const pose *pos1, const pose *pos2, float dedup_eps_sq) {
    int id = -1;
    for (int c = 0; c < *ncl; ++c) if (vdot(cl[c].n1, n1) >= RO_COS_MERGE_ANGLE) { id = c; break; }
    if (id < 0) {
        if (*ncl < RO_MAX_CLUSTERS) { id = (*ncl)++; cl[id].n1 = n1; cl[id].n2 = n2; cl[id].np = 0; }
        else { float best = -2.0f; for (int c = 0; c < *ncl; ++c) { float d = vdot(cl[c].n1, n1); if (d > best) { best = d; id = c; } } } /* (bound of this restatement) */
    }
    ClusterTmp *C = &cl[id];
    for (int i = 0; i < m->npoints; ++i) {
        TrackedContact pt = m->points[i];
        if (pos1) pt.local_p1 = pose_tp(*pos1, pt.local_p1);
        if (pos2) pt.local_p2 = pose_tp(*pos2, pt.local_p2);
        memset(&pt.data, 0, sizeof(pt.data));
        int ex = -1;
        for (int k = 0; k < C->np; ++k) if (vlen2(vsub(C->pts[k].local_p1, pt.local_p1)) < dedup_eps_sq) { ex = k; break; }
        if (ex >= 0) { if (pt.dist < C->pts[ex].dist) C->pts[ex] = pt; }
        else if (C->np < RO_CLUSTER_PTS) C->pts[C->np++] = pt;
        else { int sh = 0; for (int k = 1; k < C->np; ++k) if (C->pts[k].dist > C->pts[sh].dist) sh = k; if (pt.dist < C->pts[sh].dist) C->pts[sh] = pt; }
    }
}
/* carry_warmstart_data :129-174: every previous point with warm-start data goes to the nearest unclaimed target point (in shape 1's
 * frame) of a target whose normal agrees; tpos1[t] = the target's subshape_pos1 (NULL: none) */
static void carry_warmstart(const Manifold *const *prev, int nprev, v3 *tn1, TrackedContact **tpts, int *tnp, const pose *const *tpos1, int ntargets, float prediction) {
    const float match_eps_sq = prediction * prediction;
    for (int a = 0; a < nprev; ++a)
        for (int i = 0; i < prev[a]->npoints; ++i) {
            const TrackedContact *pp = &prev[a]->points[i];
            if (!data_has_warmstart(&pp->data)) continue;
            int bt = -1, bp = -1; float best = match_eps_sq;
            for (int t = 0; t < ntargets; ++t) {
                if (vdot(tn1[t], prev[a]->local_n1) < RO_COS_MERGE_ANGLE) continue;
                for (int k = 0; k < tnp[t]; ++k) {
                    if (data_has_warmstart(&tpts[t][k].data)) continue;
                    v3 p1 = tpos1[t] ? pose_tp(*tpos1[t], tpts[t][k].local_p1) : tpts[t][k].local_p1;
                    float d = vlen2(vsub(p1, pp->local_p1));
                    if (d < best) { best = d; bt = t; bp = k; }
                }
            }
            if (bt >= 0) tpts[bt][bp].data = pp->data;
        }
}
#endif
