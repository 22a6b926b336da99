// rp_ccd.h — the continuous-collision pass on the device (included at the end of rp_narrowphase.hip: it reuses the SAT and
// closest-point helpers of the manifold generators).
//
// DRIVER: CCDSolver::solve_continuous + apply_clamps (/root/reference/src/dynamics/ccd/ccd_solver.rs:158-340) and sweep_fast_body
// (sweeps.rs:470-640) with the tiers of sweeps.rs:22-41: a fast dynamic body (RigidBodyCcd::is_moving_fast_with_next_position,
// evaluated in body_writeback) sweeps the FIXED colliders; a `ccd_enabled` body — a bullet — sweeps every collider that is not on a
// bullet, targets standing at their (possibly just clamped) end-of-step pose.  The earliest solid impact fraction in (0, 1) clamps
// the body's pose (velocities untouched); sensors never stop a body (the paired intersection events of sensor crossings,
// ccd_solver.rs:265-320, are not raised).
//
// TIME OF IMPACT: the reference calls parry3d's query::sweep_toi (not under /root/reference).  In its place: conservative
// advancement over a lower bound of the distance — the largest separation over the SAT axes of the manifold generators, exact
// point projections for balls, segment / segment closest points for capsules, the support point for half-spaces — with Box2D's
// impact distance (core shapes within max(slop, total radius - slop), tolerance slop / 4, slop = allowed_linear_error):
//     t += (separation - target) / (approach speed along the separating direction + 4 tan(angle / 4) * rotation radius)
// until separation < target + tolerance (hit) or t >= the best fraction so far (miss); the iteration cap ends in a miss (a body that
// hovers within a few slops of a surface is the speculative contacts' job).  One workgroup per fast body: its lanes take
// (collider of the body, target collider) candidates, the earliest fraction is an atomicMin over positive float bits, so the result
// does not depend on the visiting order.  Every function mirrors the checker's restatement operation for operation.
#pragma once
#include "rp_grid.h"

#define RP_CCD_MAX_ITERS 48

struct CcdSweep { V3 c0, c1; Q4 q0, q1; V3 local_com; };
RP_DEV CcdSweep ccd_sweep_from_poses(Pose start, Pose end, V3 local_com) {
    CcdSweep s;
    s.c0 = pose_tp(start, local_com); s.c1 = pose_tp(end, local_com);
    s.q0 = start.r; s.q1 = end.r;
    if (qdot(s.q0, s.q1) < 0.0f) s.q1 = q4(-s.q1.x, -s.q1.y, -s.q1.z, -s.q1.w);
    s.local_com = local_com;
    return s;
}
RP_DEV Pose ccd_sweep_transform_at(const CcdSweep &s, float t) {
    V3 c = s.c0 + (s.c1 - s.c0) * t;
    Q4 q = qnormalize(q4(s.q0.x + (s.q1.x - s.q0.x) * t, s.q0.y + (s.q1.y - s.q0.y) * t, s.q0.z + (s.q1.z - s.q0.z) * t, s.q0.w + (s.q1.w - s.q0.w) * t));
    Pose p; p.r = q; p.t = c - qrot(q, s.local_com);
    return p;
}
// a collider's shape as the query sees it (c_shape / c_he of rp_world.h): he = cuboid half extents | half-space normal; capsule:
// he.x = half height, radius, axis; ball: radius
typedef SmShape CcdShape; // (rp_convex.h: the same record serves the support-mapped queries)
template <bool CONVEX> RP_DEV CcdShape ccd_shape_of(const DevWorld &w, int c) { // collider c as the query sees it (round shapes only exist in CONVEX worlds)
    const int sh = w.c_shape[c];
    float border = 0.0f;
    if constexpr (CONVEX) { if (sh >= RP_SHAPE_ROUND_CUBOID) border = w.c_mat[c].w; }
    return sm_shape_of(w, sh, w.c_he[c], border);
}
RP_DEV V3 ccd_clamp_box(V3 p, V3 he) { return v3(rp_clamp(p.x, -he.x, he.x), rp_clamp(p.y, -he.y, he.y), rp_clamp(p.z, -he.z, he.z)); }
RP_DEV float ccd_point_dir(V3 dv, V3 &dir) {
    float dist = len(dv);
    if (!(dist > 0.0f)) { dir = v3(0, 1, 0); return -1.0f; }
    dir = dv * (1.0f / dist);
    return dist;
}
RP_DEV float ccd_point_box(V3 p, V3 he, V3 &dir) { return ccd_point_dir(p - ccd_clamp_box(p, he), dir); }

template <bool CONVEX> __device__ float ccd_separation(const CcdShape &s1, const CcdShape &s2, Pose pos12, V3 &n1) {
    const Pose pos21 = pose_inv(pos12);
    if constexpr (CONVEX) {
        if (s1.shape >= RP_SHAPE_CYLINDER || s2.shape >= RP_SHAPE_CYLINDER || s1.border > 0.0f || s2.border > 0.0f) { // cylinders, cones, polyhedra, round shapes: the exact distance of the cores by GJK
            if (s1.shape == RP_SHAPE_HALFSPACE) {
                n1 = s1.he;
                return dot(s1.he, pose_tp(pos12, sm_support(s2, qrot_inv(pos12.r, -s1.he)))) - s2.border;
            }
            float d = sm_distance(s1, s2, pos12, n1);
            return d < 0.0f ? d : d - sm_border_radius(s1) - sm_border_radius(s2);
        }
    }
    if (s1.shape == RP_SHAPE_HALFSPACE) {
        const V3 n = s1.he;
        n1 = n;
        if (s2.shape == RP_SHAPE_BALL) return dot(n, pos12.t) - s2.radius;
        if (s2.shape == RP_SHAPE_CUBOID) {
            V3 n2 = qrot_inv(pos12.r, n);
            float ext = (fabsf(n2.x) * s2.he.x + fabsf(n2.y) * s2.he.y) + fabsf(n2.z) * s2.he.z;
            return dot(n, pos12.t) - ext;
        }
        V3 e = capsule_axis_dir(s2.axis) * s2.he.x;
        float da = dot(n, pose_tp(pos12, -e)), db = dot(n, pose_tp(pos12, e));
        return rp_min(da, db) - s2.radius;
    }
    if (s1.shape == RP_SHAPE_BALL) {
        if (s2.shape == RP_SHAPE_BALL) { float d = ccd_point_dir(pos12.t, n1); return d < 0.0f ? d : d - s1.radius - s2.radius; }
        if (s2.shape == RP_SHAPE_CUBOID) {
            V3 dir2; float d = ccd_point_box(pos21.t, s2.he, dir2);
            n1 = qrot(pos12.r, -dir2);
            return d < 0.0f ? d : d - s1.radius;
        }
        V3 e = capsule_axis_dir(s2.axis) * s2.he.x;
        V3 p = segment_project_point(pose_tp(pos12, -e), pose_tp(pos12, e), v3(0, 0, 0));
        float d = ccd_point_dir(p, n1);
        return d < 0.0f ? d : d - s1.radius - s2.radius;
    }
    if (s1.shape == RP_SHAPE_CUBOID) {
        if (s2.shape == RP_SHAPE_BALL) { float d = ccd_point_box(pos12.t, s1.he, n1); return d < 0.0f ? d : d - s2.radius; }
        if (s2.shape == RP_SHAPE_CUBOID) {
            V3 d1, d2, d3;
            float sa = sat_normal_oneway(s1.he, s2.he, pos12, d1);
            float sb = sat_normal_oneway(s2.he, s1.he, pos21, d2);
            float sc = sat_edge_twoway(s1.he, s2.he, pos12, d3);
            float sep = sa; n1 = d1;
            if (sb > sep) { sep = sb; n1 = qrot(pos12.r, -d2); }
            if (sc > sep) { sep = sc; n1 = d3; }
            return sep;
        }
        V3 e = capsule_axis_dir(s2.axis) * s2.he.x;
        V3 a2 = pose_tp(pos12, -e), b2 = pose_tp(pos12, e), d1, d3;
        float sa = sat_cuboid_segment_normal_oneway(s1.he, a2, b2, d1);
        float sc = sat_cuboid_segment_edge_twoway(s1.he, a2, b2, d3);
        float sep = sa; n1 = d1;
        if (sc > sep) { sep = sc; n1 = d3; }
        return sep - s2.radius;
    }
    {   // target capsule
        V3 e1 = capsule_axis_dir(s1.axis) * s1.he.x, a1 = -e1, b1 = e1;
        if (s2.shape == RP_SHAPE_BALL) {
            V3 p = segment_project_point(a1, b1, pos12.t);
            float d = ccd_point_dir(pos12.t - p, n1);
            return d < 0.0f ? d : d - s1.radius - s2.radius;
        }
        if (s2.shape == RP_SHAPE_CUBOID) {
            V3 a = pose_tp(pos21, a1), b = pose_tp(pos21, b1), d1, d3;
            float sa = sat_cuboid_segment_normal_oneway(s2.he, a, b, d1);
            float sc = sat_cuboid_segment_edge_twoway(s2.he, a, b, d3);
            float sep = sa; V3 dir2 = d1;
            if (sc > sep) { sep = sc; dir2 = d3; }
            n1 = qrot(pos12.r, -dir2);
            return sep - s1.radius;
        }
        V3 e2 = capsule_axis_dir(s2.axis) * s2.he.x;
        V3 a2 = pose_tp(pos12, -e2), b2 = pose_tp(pos12, e2);
        float s, t;
        closest_points_segment_segment(a1, b1, a2, b2, s, t);
        V3 p1 = a1 + (b1 - a1) * s, p2 = a2 + (b2 - a2) * t;
        float d = ccd_point_dir(p2 - p1, n1);
        return d < 0.0f ? d : d - s1.radius - s2.radius;
    }
}
RP_DEV float ccd_rot_radius(const CcdShape &s2, Pose pos_wrt_parent, V3 local_com) {
    V3 c = pos_wrt_parent.t - local_com;
    if (s2.shape == RP_SHAPE_BALL) return len(c);
    if (s2.shape == RP_SHAPE_CAPSULE) {
        V3 e = qrot(pos_wrt_parent.r, capsule_axis_dir(s2.axis) * s2.he.x);
        return rp_max(len(c - e), len(c + e));
    }
    return (len(c) + len(s2.he)) + s2.border;
}
template <bool CONVEX> __device__ float ccd_cast_pair(const CcdShape &s1, Pose target_pose, const CcdShape &s2, Pose pos_wrt_parent, const CcdSweep &sw, float rot_radius,
                               float max_fraction, float slop) {
    const float total_radius = sm_border_radius(s1) + sm_border_radius(s2); // balls, capsules, round shapes
    const float target = rp_max(slop, total_radius - slop) - total_radius, tol = 0.25f * slop;
    const V3 D = sw.c1 - sw.c0;
    const Q4 dq = qmul(sw.q1, qconj(sw.q0));
    const float sv = sqrtf((dq.x * dq.x + dq.y * dq.y) + dq.z * dq.z);
    const float rot_bound = (4.0f * sv / (1.0f + fabsf(dq.w))) * rot_radius;
    float t = 0.0f;
    for (int iter = 0; iter < RP_CCD_MAX_ITERS; ++iter) {
        Pose cp = pose_mul(ccd_sweep_transform_at(sw, t), pos_wrt_parent);
        Pose pos12 = pose_inv_mul(target_pose, cp);
        V3 n1;
        float sep = ccd_separation<CONVEX>(s1, s2, pos12, n1);
        if (sep < target + tol) return iter == 0 ? -2.0f : t; // -2: touching or overlapping at the start
        V3 nw = qrot(target_pose.r, n1);
        float approach = -dot(D, nw);
        if (approach < 0.0f) approach = 0.0f;
        float bound = approach + rot_bound;
        if (!(bound > 0.0f)) return -1.0f;
        t = t + (sep - target) / bound;
        if (!(t < max_fraction)) return -1.0f;
    }
    return -1.0f;
}
// The core of a fast piece (sweeps.rs:166-173 FastSubShape::local_centroid / min_extent, handed to the composite sweep at :386-391):
// a ball of CORE_FRACTION x the piece's smallest extent (its ccd_thickness) about the origin of its own frame.  A piece that starts
// a step touching or overlapping one sub-shape of a composite target — a thin slab lying across a mesh that has no inside — is
// swept again as this ball, so that its centre never crosses the sheet within a step (the reference's issue 524).  CORE_FRACTION
// is a parry constant that cannot be read here: 0.25, Box2D v3's (oracle/ro_ccd.h has the note).
#define RP_CCD_CORE_FRACTION 0.25f
RP_DEV CcdShape ccd_core_of(const CcdShape &s2) {
    float th = s2.shape == RP_SHAPE_BALL ? s2.radius : s2.shape == RP_SHAPE_CAPSULE ? s2.radius : rp_min(s2.he.x, rp_min(s2.he.y, s2.he.z));
    if (s2.border > 0.0f) th = th + s2.border;
    CcdShape c = s2;
    c.shape = RP_SHAPE_BALL; c.radius = RP_CCD_CORE_FRACTION * th; c.border = 0.0f; c.he = v3(c.radius, 0.0f, 0.0f);
    c.pts = nullptr; c.fn = nullptr; c.fl = nullptr; c.loop = nullptr; c.npts = 0; c.nfaces = 0;
    return c;
}
RP_DEV bool ccd_may_reach(V3 c0, V3 c1, float max_extent, V3 target_centre, float target_radius, float margin) {
    V3 p = segment_project_point(c0, c1, target_centre);
    float reach = (max_extent + target_radius) + margin;
    return len2(target_centre - p) <= reach * reach;
}
RP_DEV float ccd_bounding_radius_core(const DevWorld &w, int sh, float4 he);
RP_DEV float ccd_bounding_radius(const DevWorld &w, int c) { // Shape::compute_local_bounding_sphere of collider c (RoundShape: the inner sphere + the border)
    const int sh = w.c_shape[c];
    const float r = ccd_bounding_radius_core(w, sm_core_shape(sh), w.c_he[c]);
    return sh >= RP_SHAPE_ROUND_CUBOID ? r + w.c_mat[c].w : r;
}
RP_DEV float ccd_bounding_radius_core(const DevWorld &w, int sh, float4 he) {
    if (sh == RP_SHAPE_CUBOID) return len(v3(he));
    if (sh == RP_SHAPE_CAPSULE) return he.x + he.y;
    if (sh == RP_SHAPE_CONVEX_POLYHEDRON) return w.cv_pts[w.cv_hdr[__float_as_int(he.w)].x].w; // max |vertex| (every point row of the shape carries it)
    if (sh >= RP_SHAPE_CYLINDER) return sqrtf(he.x * he.x + he.y * he.y);
    return he.x;
}

#define CCD_MAX_FAST_COLLIDERS 64
// One workgroup per fast body of the list body_writeback filled this step (w.ccd_list, FL_CCD_N).  tier 0: non-bullets against fixed
// targets; tier 1: bullets against everything that is not on a bullet.
template <bool CONVEX> __global__ void __launch_bounds__(256) k_ccd(DevWorld w, int tier, int publish) {
    // (MULTI-mode steps: the hint buffer is published here, by the launch that follows the body write-back anyway — one launch less; the
    // two CCD counters reach the host's hints a step late, rp_counters_read reads the device)
    // (last kernel of a lean graph — rp_world.h "lean step graphs": a step that died is marked here for the graphs that follow and for
    // the host; the graph still counts in FL_SEQ.  The flags of lean_dead only ever stay or become non-zero here: every workgroup decides alike)
    const bool dead = lean_dead(w);
    if (dead && publish && blockIdx.x == 0) {
        // (a FULL graph in the one-launch form — DevWorld::lean without bit 0 — that died in its solver launch has already coloured this step's
        // begin-touch pairs: the resume must not colour them again; every other rebuild is gated by a flag its first run cleared)
        if (threadIdx.x == 0) { w.flags[FL_FAST_ABORT] = 2; w.flags[FL_SEQ] += 1; if (!(w.lean & 1)) w.flags[FL_TODO_COUNT] = 0; }
        __threadfence(); __syncthreads();
    }
    if (publish && blockIdx.x == 0) { for (int k = threadIdx.x; k < FL_COUNT; k += blockDim.x) { int v = __hip_atomic_load(&w.flags[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&w.host_flags[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } }
    if (dead) return;
    int n = w.flags[FL_CCD_N];
    if (n <= 0) return;
    if (n > w.n_bodies) n = w.n_bodies;
    __shared__ unsigned best;
    __shared__ int nfast, fast[CCD_MAX_FAST_COLLIDERS];
    const float slop = w.prm.p.normalized_allowed_linear_error * w.prm.p.length_unit; // IntegrationParameters::allowed_linear_error
    // the grid describes every collider (it follows them, and nothing edited the world since its last pass); only tier 0's FIXED targets stand where their fat AABBs say
    const bool grid_ok = tier == 0 && w.bp_incremental && w.flags[FL_BP_GRID_OK] != 0 && !w.flags[FL_BP_DIRTY];
    int n_large = w.flags[FL_N_LARGE]; if (n_large > w.large_cap) n_large = w.large_cap;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        const int bi = w.ccd_list[k];
        const int fl1 = w.b_flags[bi];
        const bool bullet = (fl1 & RP_BF_TYPE_MASK) == RP_BODY_DYNAMIC && (fl1 & RP_BF_CCD_ENABLED);
        __syncthreads();
        if ((bullet ? 1 : 0) != tier || (fl1 & RP_BF_SLEEPING)) continue; // (uniform over the workgroup)
        if (threadIdx.x == 0) { best = __float_as_uint(1.0f); nfast = -1; }
        __syncthreads();
        // the body's own colliders (enabled, not sensors), CCD_MAX_FAST_COLLIDERS at a time by attachment ordinal (c_ord): every one of
        // them is swept — a compound body of more than 64 colliders used to keep whichever 64 the atomics handed out — and
        // the chunks are the same in every run.  nfast: the largest ordinal attached to the body
        for (int c = threadIdx.x; c < w.n_colliders; c += blockDim.x) if (w.c_parent[c] == bi) atomicMax(&nfast, w.c_ord[c]);
        __syncthreads();
        const int max_ord = nfast;
        Pose start, end;
        start.t = v3(w.b_ccd0_pos[bi]); start.r = q4(w.b_ccd0_rot[bi]);
        end.t = v3(w.b_pos[bi]); end.r = q4(w.b_rot[bi]);
        const V3 lcom = v3(w.b_lcom_invm[bi]);
        const float max_extent = w.b_invpi[bi].w;
        const CcdSweep sw = ccd_sweep_from_poses(start, end, lcom);
        // tier 0, grid in service: the cells under the swept volume's box (see try_target below); too many cells: the plain walk
        const float ic = w.prm.inv_cell_size;
        const float reach = max_extent + 2.0f * slop;
        const V3 qmn = v3(fminf(sw.c0.x, sw.c1.x) - reach, fminf(sw.c0.y, sw.c1.y) - reach, fminf(sw.c0.z, sw.c1.z) - reach);
        const V3 qmx = v3(fmaxf(sw.c0.x, sw.c1.x) + reach, fmaxf(sw.c0.y, sw.c1.y) + reach, fmaxf(sw.c0.z, sw.c1.z) + reach);
        const int qlo[3] = {cell_coord(qmn.x, ic), cell_coord(qmn.y, ic), cell_coord(qmn.z, ic)};
        const int qnx = cell_coord(qmx.x, ic) - qlo[0] + 1, qny = cell_coord(qmx.y, ic) - qlo[1] + 1, qnz = cell_coord(qmx.z, ic) - qlo[2] + 1;
        const bool few_cells = qnx > 0 && qny > 0 && qnz > 0 && qnx <= 64 && qny <= 64 && qnz <= 64 && qnx * qny * qnz <= 2048;
        const int ncell = few_cells ? qnx * qny * qnz : 0;
        const bool use_grid = grid_ok && few_cells; // (uniform over the workgroup)
        for (int base = 0; base <= max_ord; base += CCD_MAX_FAST_COLLIDERS) {
          __syncthreads();
          for (int q = threadIdx.x; q < CCD_MAX_FAST_COLLIDERS; q += blockDim.x) fast[q] = -1;
          __syncthreads();
          for (int c = threadIdx.x; c < w.n_colliders; c += blockDim.x) {
              if (w.c_parent[c] != bi) continue;
              const int o = w.c_ord[c] - base;
              if (o < 0 || o >= CCD_MAX_FAST_COLLIDERS) continue;
              uint2 g = w.c_groups[c];
              if ((g.x == 0 && g.y == 0) || (__float_as_int(w.c_events[c].x) & RP_EVENTS_SENSOR_BIT)) continue;
              if (w.c_shape[c] == RP_SHAPE_TRIMESH) continue; // a mesh is never the fast shape (sweeps.rs:86-97: shape_never_ccd_swept)
              fast[o] = c;
          }
          __syncthreads();
          for (int f = 0; f < CCD_MAX_FAST_COLLIDERS; ++f) {
            const int c1 = fast[f];
            if (c1 < 0) continue; // (uniform over the workgroup)
            // a compound is swept child by child (FastShapeKind::Compound, sweeps.rs:337-345): each part with its own pose on the body
            const bool comp1 = shape_is_composite(w.c_shape[c1]);
            const int nparts1 = comp1 ? co_num_subs(w, c1) : 1; // (uniform over the workgroup)
            for (int part = 0; part < nparts1; ++part) {
            CcdShape s2 = ccd_shape_of<CONVEX>(w, c1);
            Pose pwp; pwp.t = v3(w.c_lpos[c1]); pwp.r = q4(w.c_lrot[c1]);
            if (comp1) { const SubShape a = co_sub(w, c1, part); s2 = sm_shape_of(w, a.sh, a.he, a.bd); pwp = pose_mul(pwp, a.pos); }
            const float rot_radius = ccd_rot_radius(s2, pwp, lcom);
            const uint2 g1 = w.c_groups[c1];
            // one candidate target: the filters of sweep_fast_body, the reach pre-filter, the cast
            auto try_target = [&](int c2) {
                const int p2 = w.c_parent[c2];
                if (c2 == c1 || p2 == bi) return;
                if (w.n_sub > 1 && w.c_sub[c2] != w.c_sub[c1]) return; // another sub-world (rp_world_begin_subworld)
                const uint2 g2 = w.c_groups[c2];
                if ((g2.x == 0 && g2.y == 0) || (__float_as_int(w.c_events[c2].x) & RP_EVENTS_SENSOR_BIT)) return;
                const int fl2 = p2 >= 0 ? w.b_flags[p2] : RP_BODY_FIXED;
                const bool fixed2 = (fl2 & RP_BF_TYPE_MASK) == RP_BODY_FIXED;
                // tier_allows (sweeps.rs:35-41)
                if (tier) { if ((fl2 & RP_BF_TYPE_MASK) == RP_BODY_DYNAMIC && (fl2 & RP_BF_CCD_ENABLED)) return; } else if (!fixed2) return;
                if (!((g1.x & g2.y) != 0 && (g2.x & g1.y) != 0)) return; // collision_groups.test
                Pose tp = collider_world_pose(w, c2); // target_collider_pose (:97-102): bodies already stand at their end-of-step pose
                const int sh2 = w.c_shape[c2];
                if (shape_is_composite(sh2)) {
                    // a composite target (sweeps.rs:255-262, :384-400: sweep_time_of_impact_composite): every sub-shape whose box meets the
                    // swept volume's box, taken into the composite's frame, is a target of its own; the earliest fraction wins (an atomicMin:
                    // no order dependence).  One lane walks the sub-shapes of ITS target (round 5: composites used to be skipped)
                    if constexpr (CONVEX) {
                        const V3 qc = (qmn + qmx) * 0.5f, qh = (qmx - qmn) * 0.5f;
                        const V3 lc = pose_itp(tp, qc);
                        float m[3][3]; quat_to_mat(tp.r, m);
                        const V3 lh = v3(fabsf(m[0][0]) * qh.x + fabsf(m[1][0]) * qh.y + fabsf(m[2][0]) * qh.z,
                                         fabsf(m[0][1]) * qh.x + fabsf(m[1][1]) * qh.y + fabsf(m[2][1]) * qh.z,
                                         fabsf(m[0][2]) * qh.x + fabsf(m[1][2]) * qh.y + fabsf(m[2][2]) * qh.z);
                        const int first = co_first_sub(w, c2), nsub = co_num_subs(w, c2);
                        for (int i = 0; i < nsub; ++i) {
                            const float4 amn = w.cm_min[first + i], amx = w.cm_max[first + i];
                            if (amn.x > lc.x + lh.x || amx.x < lc.x - lh.x || amn.y > lc.y + lh.y || amx.y < lc.y - lh.y || amn.z > lc.z + lh.z || amx.z < lc.z - lh.z) continue;
                            const SubShape b = co_sub(w, c2, i);
                            const Pose tpose = b.has_pose ? pose_mul(tp, b.pos) : tp;
                            if (b.has_pose && !ccd_may_reach(sw.c0, sw.c1, max_extent, tpose.t, ccd_bounding_radius_core(w, sm_core_shape(b.sh), b.he) + b.bd, 2.0f * slop)) continue;
                            CcdShape s1 = sm_shape_of(w, b.sh, b.he, b.bd);
                            s1.tri[0] = b.tri[0]; s1.tri[1] = b.tri[1]; s1.tri[2] = b.tri[2];
                            const float cur = __uint_as_float(__hip_atomic_load(&best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                            float hit = ccd_cast_pair<CONVEX>(s1, tpose, s2, pwp, sw, rot_radius, cur, slop);
                            if (hit == -2.0f) { // starts on this sub-shape: the core ball's turn
                                const CcdShape core = ccd_core_of(s2);
                                hit = ccd_cast_pair<CONVEX>(s1, tpose, core, pwp, sw, ccd_rot_radius(core, pwp, lcom), cur, slop);
                            }
                            if (hit > 0.0f && hit < cur) atomicMin(&best, __float_as_uint(hit));
                        }
                    }
                    return;
                }
                if (sh2 != RP_SHAPE_HALFSPACE && !ccd_may_reach(sw.c0, sw.c1, max_extent, tp.t, ccd_bounding_radius(w, c2), 2.0f * slop)) return;
                const CcdShape s1 = ccd_shape_of<CONVEX>(w, c2);
                const float cur = __uint_as_float(__hip_atomic_load(&best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                const float hit = ccd_cast_pair<CONVEX>(s1, tp, s2, pwp, sw, rot_radius, cur, slop);
                if (hit > 0.0f && hit < cur) atomicMin(&best, __float_as_uint(hit));
            };
            if (use_grid) {
                // tier 0 sweeps FIXED colliders: their fat AABBs are where they stand, and the broad phase's grid (which follows every
                // collider: rp_grid.h) finds the ones whose box meets the swept volume's box — the centre-of-mass segment inflated by
                // max_extent + 2 slop, what ccd_may_reach measures against — instead of a walk over every collider of the world per fast
                // body (23 % of a step in which thousands of small shapes land: profiles/r04_shapes_rain_kernel_stats.txt).  The large
                // list (slabs, walls, half-spaces) is walked whole; a collider met through several cells is taken from the cell that
                // holds the min corner of (its box ∩ the query box).  The earliest fraction is an atomicMin: no order dependence.
                { int q0, q1; large_range_of(w, c1, n_large, q0, q1); for (int q = q0 + threadIdx.x; q < q1; q += blockDim.x) try_target(w.large_list[q]); }
                const int gcur = BP_GPAR(w);
                for (int idx = threadIdx.x; idx < ncell * RP_BP_BUCKET; idx += blockDim.x) {
                    const int cell = idx / RP_BP_BUCKET, e = idx % RP_BP_BUCKET;
                    const int x = qlo[0] + cell % qnx, y = qlo[1] + (cell / qnx) % qny, z = qlo[2] + cell / (qnx * qny);
                    const int h = (int)(rp_hash64(cell_key_of(w, c1, x, y, z)) & (unsigned long long)(w.grid_cap - 1));
                    int nb = w.bk_cnt[gcur][h]; if (nb > RP_BP_BUCKET) nb = RP_BP_BUCKET;
                    if (e >= nb) continue;
                    const int it = w.bk_items[gcur][(size_t)h * RP_BP_BUCKET + e], j = it & 0xffffff;
                    if (w.c_inlarge[j] || !bp_entry_is_cell(w, it, x, y, z)) continue;
                    const float4 jmn = w.c_fatmin[j];
                    if (cell_coord(fmaxf(jmn.x, qmn.x), ic) != x || cell_coord(fmaxf(jmn.y, qmn.y), ic) != y || cell_coord(fmaxf(jmn.z, qmn.z), ic) != z) continue;
                    try_target(j);
                }
            } else {
                for (int c2 = threadIdx.x; c2 < w.n_colliders; c2 += blockDim.x) try_target(c2);
            }
            } // (the parts of a compound fast collider)
          }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float fraction = __uint_as_float(best);
            if (fraction < 1.0f) { // apply_clamps (ccd_solver.rs:325-339) + what advance_to_final_positions derives from the pose
                const Pose p = ccd_sweep_transform_at(sw, fraction);
                w.b_pos[bi] = f4(p.t, 0.0f); w.b_rot[bi] = f4(p.r);
                w.b_wcom[bi] = f4(qrot(p.r, lcom) + p.t, 0.0f);
                Sym3 ii = world_inv_inertia(v3(w.b_invpi[bi]), q4(w.b_pframe[bi]), p.r);
                apply_locked_rotations((fl1 >> RP_BF_LOCK_SHIFT) & 0x3f, ii);
                w.b_eii0[bi] = make_float4(ii.m11, ii.m12, ii.m13, ii.m22);
                w.b_eii1[bi] = make_float4(ii.m23, ii.m33, 0.0f, 0.0f);
                atomicAdd(&w.flags[FL_CCD_CLAMPS], 1);
            }
        }
    }
}
// true = rp_launch_ccd will run for this world (then it can carry the hint publication: rp_ccd_launches / `publish`)
bool rp_ccd_launches(const DevWorld &w) { return !(w.prm.p.max_ccd_substeps == 0 || w.n_bodies == 0 || w.n_colliders == 0); }
void rp_launch_ccd(const DevWorld &w, hipStream_t st, int has_bullets, int publish) {
    if (!rp_ccd_launches(w)) return;
    if (w.has_convex) { // worlds with a cylinder / cone: the distance of such a pair is a GJK run (rp_convex.h)
        hipLaunchKernelGGL(k_ccd<true>, dim3(240), dim3(256), 0, st, w, 0, publish);
        if (has_bullets) hipLaunchKernelGGL(k_ccd<true>, dim3(64), dim3(256), 0, st, w, 1, 0);
        return;
    }
    hipLaunchKernelGGL(k_ccd<false>, dim3(64), dim3(256), 0, st, w, 0, publish);
    if (has_bullets) hipLaunchKernelGGL(k_ccd<false>, dim3(64), dim3(256), 0, st, w, 1, 0);
}
