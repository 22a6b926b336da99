// rp_joints.h — impulse joints on the global solver path: locked linear and angular axes
// (spherical, revolute, prismatic and fixed joints), limits and motors of the free axes, rows rebuilt from the current poses
// every substep, solved before the contacts in every pass, coloured in the contacts' colour space.
//
// Restates (one joint per thread instead of one 4-lane chunk):
//   JointConstraintBuilder::update            solver/joint_constraint/joint_constraint_builder.rs:76-150
//   JointConstraint::<Real,1>::update         solver/joint_constraint/joint_velocity_constraint.rs:144-353
//   JointConstraintHelper::{new, lock_linear, lock_angular, finalize_constraints}
//                                             solver/joint_constraint/joint_constraint_helper.rs:95-164, 411-458, 628-673, 676-720
//   JointConstraintHelper::{limit_linear, limit_angular, motor_linear, motor_angular}   :166-208, 468-564, 285-331, 566-625
//   JointMotor::motor_params, MotorModel::combine_coefficients   joint/generic_joint.rs:236-250, joint/motor_model.rs:39-58
//   JointConstraint::{solve_generic, warmstart_generic, remove_bias_from_rhs}   joint_velocity_constraint.rs:97-142
// The reference's wide (SIMD) path uses nalgebra's quaternion->matrix / rotate formulas and its scalar
// path glam's, and the two emit the lock rows in different orders (wide: linear then angular; scalar: angular then
// linear, joint_velocity_constraint.rs:253-283 vs :438-468); like the oracle this file uses the scalar path's forms
// and row order for every joint.
#pragma once
#include "rp_world.h"

// row planes: JR[plane][joint]; planes 0, 1 = the two inverse masses, row r uses planes 2 + 6*r .. 2 + 6*r + 5.  A joint has at
// most 12 rows: one per motorised free axis, then one per locked or limited axis.
enum { JR_IM1 = 0, JR_IM2 = 1, JR_ROW0 = 2, JR_LIN = 0, JR_A1, JR_A2, JR_I1, JR_I2, JR_BND, JR_ROW_PLANES = 6, JR_MAX_ROWS = 12, JR_COUNT = JR_ROW0 + JR_ROW_PLANES * JR_MAX_ROWS };
static_assert(JR_COUNT == RP_JR_COUNT, "DevWorld::JR plane count");
#define JRP(plane, j) w.JR[(size_t)(plane) * w.n_joints + (j)]
#define JRR(r, plane, j) JRP(JR_ROW0 + JR_ROW_PLANES * (r) + (plane), j)

// select_active_interactions (impulse_joint_set.rs:504-572): a joint takes part in the step when it has a non-fixed side and
// none of its dynamic / kinematic bodies sleeps (removed joints have no side left)
RP_DEV bool joint_live(const DevWorld &w, int j) {
    int b1 = w.j_b1[j], b2 = w.j_b2[j];
    if (b1 < 0 && b2 < 0) return false;
    if (!w.sleep_enabled) return true;
    return !(b1 >= 0 && (w.b_flags[b1] & RP_BF_SLEEPING)) && !(b2 >= 0 && (w.b_flags[b2] & RP_BF_SLEEPING));
}

struct JointRow { V3 lin_jac, ang_jac1, ang_jac2, ii1, ii2; float impulse, inv_lhs, rhs, rhs_wo_bias, cfm_gain, cfm_coeff, bmin, bmax; };
#define JR_UNBOUNDED 3.402823466e+38f // impulse_bounds of a lock row: [-f32::MAX, f32::MAX]

// The two words of a row that a SWEEP changes — the accumulated impulse (JR_LIN.w) and, when the bias is removed, the right-hand side
// (JR_A2.w).  Worlds that may run their sweeps on LDS tiles (rp_tiles.hip) keep them in DevWorld::jm instead, two copies: a tile sweep
// reads copy c_par and its owner instances write the other one (a halo instance must not see the owner's result of the same sweep);
// every other kernel works in place on the current copy.  jm == null: the words live in the row planes, as ever.
RP_DEV float2 jm_get(const DevWorld &w, int j, int r, int par) { return w.jm[((size_t)par * JR_MAX_ROWS + r) * w.n_joints + j]; }
RP_DEV void jm_put(const DevWorld &w, int j, int r, int par, float impulse, float rhs) { w.jm[((size_t)par * JR_MAX_ROWS + r) * w.n_joints + j] = make_float2(impulse, rhs); }
RP_DEV float jrow_impulse(const DevWorld &w, int j, int r) { return w.jm ? jm_get(w, j, r, w.c_par).x : JRR(r, JR_LIN, j).w; }

RP_DEV void jrow_load(const DevWorld &w, int j, int r, JointRow &c) {
    float4 a = JRR(r, JR_LIN, j), b = JRR(r, JR_A1, j), d = JRR(r, JR_A2, j), e = JRR(r, JR_I1, j), f = JRR(r, JR_I2, j);
    c.lin_jac = v3(a); c.impulse = a.w; c.ang_jac1 = v3(b); c.inv_lhs = b.w; c.ang_jac2 = v3(d); c.rhs = d.w;
    c.ii1 = v3(e); c.rhs_wo_bias = e.w; c.ii2 = v3(f); c.cfm_gain = f.w;
    float4 g = JRR(r, JR_BND, j); c.bmin = g.x; c.bmax = g.y;
    if (w.jm) { const float2 m = jm_get(w, j, r, w.c_par); c.impulse = m.x; c.rhs = m.y; }
}
RP_DEV void jrow_store(const DevWorld &w, int j, int r, const JointRow &c) {
    JRR(r, JR_LIN, j) = f4(c.lin_jac, c.impulse); JRR(r, JR_A1, j) = f4(c.ang_jac1, c.inv_lhs);
    JRR(r, JR_A2, j) = f4(c.ang_jac2, c.rhs); JRR(r, JR_I1, j) = f4(c.ii1, c.rhs_wo_bias); JRR(r, JR_I2, j) = f4(c.ii2, c.cfm_gain);
    JRR(r, JR_BND, j) = make_float4(c.bmin, c.bmax, 0.0f, 0.0f);
    if (w.jm) jm_put(w, j, r, w.c_par, c.impulse, c.rhs);
}
// the planes of a row without its two sweep words (tile sweeps that rebuild rows themselves: the words go to the other copy of jm)
RP_DEV void jrow_store_planes(const DevWorld &w, int j, int r, const JointRow &c) {
    JRR(r, JR_LIN, j) = f4(c.lin_jac, c.impulse); JRR(r, JR_A1, j) = f4(c.ang_jac1, c.inv_lhs);
    JRR(r, JR_A2, j) = f4(c.ang_jac2, c.rhs); JRR(r, JR_I1, j) = f4(c.ii1, c.rhs_wo_bias); JRR(r, JR_I2, j) = f4(c.ii2, c.cfm_gain);
    JRR(r, JR_BND, j) = make_float4(c.bmin, c.bmax, 0.0f, 0.0f);
}
// rows of a joint: the motors of its free axes (GenericJoint::motor_axes & !locked_axes), its locked axes, then the limits of
// its free axes (limit_axes & !locked_axes)
// (GenericJoint::coupled_axes rides in bits 8..13 of the `limited` word — j_limited — so that every caller that hands the three masks on
// hands it on too: JR_COUPLED(limited).)  Per-axis rows skip the coupled axes; the coupled linear axes add ONE motor row and ONE limit
// row (carried by the first coupled axis), two coupled angular axes one limit row (joint_velocity_constraint.rs:159-352).
#define JR_COUPLED(limited_word) (((limited_word) >> 8) & 0x3f)
RP_DEV int joint_row_count(int locked, int limited, int motor) {
    const int coupled = JR_COUPLED(limited);
    locked &= 0x3f; limited &= 0x3f & ~locked; motor &= 0x3f & ~locked;
    int n = __popc((unsigned)locked) + __popc((unsigned)(limited & ~coupled)) + __popc((unsigned)(motor & ~coupled));
    if (coupled) {
        if (motor & coupled & 7) n++;
        if ((coupled & 0x38) && (limited & (1 << (__ffs(coupled & 0x38) - 1)))) n++;
        if ((coupled & 7) && (limited & (1 << (__ffs(coupled & 7) - 1)))) n++;
    }
    return n;
}
// WritebackId of row k (scalar update order, joint_velocity_constraint.rs:186-352): Motor(3 + a), Motor(i), the coupled linear motor,
// Dof(3 + a) for the locked angular axes, Dof(i) for the locked linear ones, then Limit(3 + a), Limit(i), the coupled angular limit, the
// coupled linear limit; Dof = axis, Limit = 6 + axis, Motor = 12 + axis
RP_DEV int joint_row_dof(int locked, int limited, int motor, int k) {
    const int coupled = JR_COUPLED(limited);
    locked &= 0x3f; limited &= 0x3f & ~locked; motor &= 0x3f & ~locked;
    const int lim1 = limited & ~coupled, mot1 = motor & ~coupled;
    for (int a = 0; a < 3; ++a) if (mot1 & (8 << a)) { if (k == 0) return 12 + 3 + a; --k; }
    for (int i = 0; i < 3; ++i) if (mot1 & (1 << i)) { if (k == 0) return 12 + i; --k; }
    if (motor & coupled & 7) { if (k == 0) return 12 + __ffs(coupled & 7) - 1; --k; }
    for (int a = 0; a < 3; ++a) if (locked & (8 << a)) { if (k == 0) return 3 + a; --k; }
    for (int i = 0; i < 3; ++i) if (locked & (1 << i)) { if (k == 0) return i; --k; }
    for (int a = 0; a < 3; ++a) if (lim1 & (8 << a)) { if (k == 0) return 6 + 3 + a; --k; }
    for (int i = 0; i < 3; ++i) if (lim1 & (1 << i)) { if (k == 0) return 6 + i; --k; }
    if ((coupled & 0x38) && (limited & (1 << (__ffs(coupled & 0x38) - 1)))) { if (k == 0) return 6 + __ffs(coupled & 0x38) - 1; --k; }
    if ((coupled & 7) && (limited & (1 << (__ffs(coupled & 7) - 1)))) { if (k == 0) return 6 + __ffs(coupled & 7) - 1; --k; }
    return 0;
}
// impulse written back by the last step for a WritebackId (warm start of substep 0)
RP_DEV float joint_seed_impulse(const DevWorld &w, int j, int dof) {
    int g = dof / 3, c = dof - 3 * g;
    float4 v = g == 0 ? w.j_imp[j] : g == 1 ? w.j_imp_ang[j] : g == 2 ? w.j_imp_lim[j] : g == 3 ? w.j_imp_lim_ang[j] : g == 4 ? w.j_imp_mot[j] : w.j_imp_mot_ang[j];
    return c == 0 ? v.x : c == 1 ? v.y : v.z;
}
// JointMotor::motor_params(dt) of axis `axis`
struct MotorParams { float erp_inv_dt, cfm_coeff, cfm_gain, target_pos, target_vel, max_impulse; };
RP_DEV MotorParams joint_motor_params(const DevWorld &w, int j, int axis, float dt) {
    float4 a = w.j_mot[(size_t)(2 * axis) * w.n_joints + j], b = w.j_mot[(size_t)(2 * axis + 1) * w.n_joints + j]; // (target_vel, target_pos, stiffness, damping), (max_force, model)
    MotorParams p;
    p.erp_inv_dt = a.z * rp_inv(dt * a.z + a.w);
    float c = rp_inv(dt * dt * a.z + dt * a.w);
    bool acc = b.y == 0.0f; // MotorModel::AccelerationBased
    p.cfm_coeff = acc ? c : 0.0f; p.cfm_gain = acc ? 0.0f : c;
    p.target_pos = a.y; p.target_vel = a.x; p.max_impulse = b.x * dt;
    return p;
}
// JointConstraintHelper::finalize_constraints over one block of rows (modified Gram-Schmidt; rows with bounded impulses — limits,
// motors — are not removed from the others), the warm-start carry of JointConstraintBuilder::update, and the store
RP_DEV void joint_finalize_store(const DevWorld &w, int j, JointRow *rows, const int *dof, int len, int base, V3 imsum, int substep_id) {
    for (int a = 0; a < len; ++a) {
        JointRow &cj = rows[a];
        float dot_jj = dot(cj.lin_jac, cmul(imsum, cj.lin_jac)) + dot(cj.ii1, cj.ang_jac1) + dot(cj.ii2, cj.ang_jac2);
        float cfm_gain = dot_jj * cj.cfm_coeff + cj.cfm_gain;
        float inv_dot_jj = rp_inv(dot_jj);
        cj.inv_lhs = rp_inv(dot_jj + cfm_gain);
        cj.cfm_gain = cfm_gain;
        if (cj.bmin != -JR_UNBOUNDED || cj.bmax != JR_UNBOUNDED) continue;
        for (int b = a + 1; b < len; ++b) {
            JointRow &ci = rows[b];
            float dot_ij = dot(ci.lin_jac, cmul(imsum, cj.lin_jac)) + dot(ci.ii1, cj.ang_jac1) + dot(ci.ii2, cj.ang_jac2);
            float coeff = dot_ij * inv_dot_jj;
            ci.lin_jac = ci.lin_jac - cj.lin_jac * coeff;
            ci.ang_jac1 = ci.ang_jac1 - cj.ang_jac1 * coeff;
            ci.ang_jac2 = ci.ang_jac2 - cj.ang_jac2 * coeff;
            ci.ii1 = ci.ii1 - cj.ii1 * coeff;
            ci.ii2 = ci.ii2 - cj.ii2 * coeff;
            ci.rhs_wo_bias = ci.rhs_wo_bias - cj.rhs_wo_bias * coeff;
            ci.rhs = ci.rhs - cj.rhs * coeff;
        }
    }
    const bool ws = w.prm.p.warmstart_joints != 0;
    for (int k = 0; k < len; ++k) {
        // the previous substep's impulse of the same row is still in its plane (the row count of a joint is constant within a step)
        if (ws) rows[k].impulse = (substep_id == 0 ? joint_seed_impulse(w, j, dof[k]) : jrow_impulse(w, j, base + k)) * w.prm.p.warmstart_coefficient;
        jrow_store(w, j, base + k, rows[k]);
    }
}

// The same block with a compile-time row count and dofs 0..LEN-1 (joints that lock the three linear axes and nothing else — every
// joint of b3d_joint_grid): every loop unrolls, the rows stay in registers.  The run-time-indexed form above keeps its rows in scratch
// memory (720 B per lane of the dataflow kernel), a memory round trip per access on a path that is rebuilt every substep.
template <int LEN>
RP_DEV void joint_finalize_static(const DevWorld &w, int j, JointRow (&rows)[LEN], V3 imsum, int substep_id) {
#pragma unroll
    for (int a = 0; a < LEN; ++a) {
        JointRow &cj = rows[a];
        float dot_jj = dot(cj.lin_jac, cmul(imsum, cj.lin_jac)) + dot(cj.ii1, cj.ang_jac1) + dot(cj.ii2, cj.ang_jac2);
        float cfm_gain = dot_jj * cj.cfm_coeff + cj.cfm_gain;
        float inv_dot_jj = rp_inv(dot_jj);
        cj.inv_lhs = rp_inv(dot_jj + cfm_gain);
        cj.cfm_gain = cfm_gain;
#pragma unroll
        for (int b = a + 1; b < LEN; ++b) { // (lock rows are unbounded: none is skipped)
            JointRow &ci = rows[b];
            float dot_ij = dot(ci.lin_jac, cmul(imsum, cj.lin_jac)) + dot(ci.ii1, cj.ang_jac1) + dot(ci.ii2, cj.ang_jac2);
            float coeff = dot_ij * inv_dot_jj;
            ci.lin_jac = ci.lin_jac - cj.lin_jac * coeff;
            ci.ang_jac1 = ci.ang_jac1 - cj.ang_jac1 * coeff;
            ci.ang_jac2 = ci.ang_jac2 - cj.ang_jac2 * coeff;
            ci.ii1 = ci.ii1 - cj.ii1 * coeff;
            ci.ii2 = ci.ii2 - cj.ii2 * coeff;
            ci.rhs_wo_bias = ci.rhs_wo_bias - cj.rhs_wo_bias * coeff;
            ci.rhs = ci.rhs - cj.rhs * coeff;
        }
    }
    const bool ws = w.prm.p.warmstart_joints != 0;
#pragma unroll
    for (int k = 0; k < LEN; ++k)
        if (ws) rows[k].impulse = (substep_id == 0 ? joint_seed_impulse(w, j, k) : jrow_impulse(w, j, k)) * w.prm.p.warmstart_coefficient;
}
// where the rebuilt rows of a spherical joint go: to their planes (every launch path), or straight into the solve of the tile sweep that
// rebuilt them (rp_tiles.hip: TileJointBuild keeps them in registers, the owner instance also stores them for the sweeps that follow)
struct JointRowsToPlanes {
    RP_DEV void take3(const DevWorld &w, int j, JointRow (&r3)[3], V3 im1, V3 im2) const {
#pragma unroll
        for (int k = 0; k < 3; ++k) jrow_store(w, j, k, r3[k]);
        JRP(JR_IM1, j) = f4(im1, 0.0f); JRP(JR_IM2, j) = f4(im2, 0.0f);
    }
};

// How a joint reaches the solver bodies: plain HBM arrays on the per-stage launch path (PlainBodyIO), tagged write-through
// records on the dataflow path (rp_flow.hip).  `side` = 0 / 1 for body1 / body2.
struct PlainBodyIO {
    const DevWorld &w;
    RP_DEV void bodies(const DevWorld &w_, int j, int &b1, int &b2) const { b1 = w_.j_b1[j]; b2 = w_.j_b2[j]; }
    RP_DEV void pose(int side, int b, Pose &p) const { p.r = q4(w.s_rot[b]); p.t = v3(w.s_trans[b]); }
    RP_DEV void load_vel(int side, int b, V3 &l, V3 &a) const { l = v3(w.s_lin[b]); a = v3(w.s_ang[b]); }
    RP_DEV void store_vel(int side, int b, V3 l, V3 a) const { w.s_lin[b] = f4(l, 0.0f); w.s_ang[b] = f4(a, 0.0f); }
    RP_DEV int jm_out(const DevWorld &w_) const { return w_.c_par; } // the sweep's mutable row words go back in place
    RP_DEV bool jm_store() const { return true; }
};

// JointConstraintBuilder::update for joint j (rows rebuilt from the solver poses s_rot / s_trans).
// ONLY_SPHERICAL: the caller knows that every joint it passes locks the three linear axes and nothing else (DevWorld::joints_spherical):
// the run-time-indexed general form is compiled out
template <class IO, class SINK = JointRowsToPlanes, bool ONLY_SPHERICAL = false>
RP_DEV void joint_update_one_t(const DevWorld &w, const IO &io, int j, int substep_id, const SINK &sink = SINK()) {
    // (ONLY_SPHERICAL: the masks are known; an IO that already knows the joint's bodies — the tile sweeps: from the cone entry — says so)
    int b1, b2; io.bodies(w, j, b1, b2);
    const int locked = ONLY_SPHERICAL ? 0x7 : w.j_locked[j], limited_word = ONLY_SPHERICAL ? 0 : w.j_limited[j];
    const int coupled = ONLY_SPHERICAL ? 0 : JR_COUPLED(limited_word);   // GenericJoint::coupled_axes
    const int limited_all = limited_word & 0x3f & ~locked, motor_all = ONLY_SPHERICAL ? 0 : (w.j_motor[j] & 0x3f & ~locked);
    const int limited = limited_all & ~coupled, motor = motor_all & ~coupled; // the per-axis rows skip the coupled axes
    Pose p1, p2; p1.r = q4(0, 0, 0, 1); p1.t = v3(0, 0, 0); p2 = p1;
    V3 im1 = v3(0, 0, 0), im2 = im1; Sym3 ii1 = {0, 0, 0, 0, 0, 0}, ii2 = ii1;
    if (b1 >= 0) { io.pose(0, b1, p1); im1 = v3(w.b_eim[b1]); float4 a = w.b_eii0[b1], b = w.b_eii1[b1]; ii1.m11 = a.x; ii1.m12 = a.y; ii1.m13 = a.z; ii1.m22 = a.w; ii1.m23 = b.x; ii1.m33 = b.y; }
    if (b2 >= 0) { io.pose(1, b2, p2); im2 = v3(w.b_eim[b2]); float4 a = w.b_eii0[b2], b = w.b_eii1[b2]; ii2.m11 = a.x; ii2.m12 = a.y; ii2.m13 = a.z; ii2.m22 = a.w; ii2.m23 = b.x; ii2.m33 = b.y; }
    Pose lf1, lf2; lf1.r = q4(w.j_f1r[j]); lf1.t = v3(w.j_f1t[j]); lf2.r = q4(w.j_f2r[j]); lf2.t = v3(w.j_f2t[j]);
    Pose frame1 = pose_mul(p1, lf1), frame2 = pose_mul(p2, lf2);
    V3 world_com1 = p1.t, world_com2 = p2.t;
    float m[3][3]; quat_to_mat(frame1.r, m);
    V3 col[3] = {v3(m[0][0], m[1][0], m[2][0]), v3(m[0][1], m[1][1], m[2][1]), v3(m[0][2], m[1][2], m[2][2])};
    V3 lin_err = frame2.t - frame1.t;
    V3 new_center1 = frame2.t;
#pragma unroll
    for (int i = 0; i < 3; ++i) if (locked & (1 << i)) new_center1 = new_center1 - col[i] * dot(lin_err, col[i]);
    frame1.t = new_center1;
    V3 r1 = frame1.t - world_com1, r2 = frame2.t - world_com2;
    V3 c1x = v3(0.0f, r1.z, -r1.y), c1y = v3(-r1.z, 0.0f, r1.x), c1z = v3(r1.y, -r1.x, 0.0f);
    V3 c2x = v3(0.0f, r2.z, -r2.y), c2y = v3(-r2.z, 0.0f, r2.x), c2z = v3(r2.y, -r2.x, 0.0f);
    V3 imsum = im1 + im2;
    if (ONLY_SPHERICAL || (locked == 0x7 && !motor_all && !limited_all && !coupled)) { // three locked linear axes, nothing else (a spherical joint): rows in registers
        JointRow r3[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            JointRow &c = r3[i];
            c.impulse = 0.0f;
            c.lin_jac = col[i];
            c.ang_jac1 = c1x * col[i].x + c1y * col[i].y + c1z * col[i].z;
            c.ang_jac2 = c2x * col[i].x + c2y * col[i].y + c2z * col[i].z;
            float rhs_wo_bias = 0.0f;
            float rhs_bias = dot(c.lin_jac, lin_err) * w.prm.joint_erp_inv_dt;
            c.ii1 = sym_mul(ii1, c.ang_jac1);
            c.ii2 = sym_mul(ii2, c.ang_jac2);
            c.inv_lhs = 0.0f; c.cfm_coeff = w.prm.joint_cfm_coeff; c.cfm_gain = 0.0f; c.bmin = -JR_UNBOUNDED; c.bmax = JR_UNBOUNDED;
            c.rhs = rhs_wo_bias + rhs_bias; c.rhs_wo_bias = rhs_wo_bias;
        }
        joint_finalize_static<3>(w, j, r3, imsum, substep_id);
        sink.take3(w, j, r3, im1, im2);
        return;
    }
    if (ONLY_SPHERICAL) return;
    JointRow rows[6];
    int dof[6] = {0, 0, 0, 0, 0, 0};
    int len = 0, base = 0;
    if (motor_all) {
        // motor rows come first and are finalised as a block of their own (joint_velocity_constraint.rs:186-246): motor_angular for
        // the angular axes (joint_constraint_helper.rs:566-625), then motor_linear (:285-331)
        const float dt = w.prm.dt_sub;
        Q4 q1 = frame1.r, q2 = frame2.r;
        float sgn = copysignf(1.0f, qdot(q1, q2));
        Q4 ang_err = qmul(qconj(q1), q2);
        float imag[3] = {ang_err.x * sgn, ang_err.y * sgn, ang_err.z * sgn};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (!(motor & (8 << a))) continue;
            MotorParams mp = joint_motor_params(w, j, 3 + a, dt);
            V3 ang_jac = col[a];
            float rhs_wo_bias = 0.0f;
            if (mp.erp_inv_dt != 0.0f) {
                float ang_dist = rp_asin_portable(rp_clamp(imag[a], -1.0f, 1.0f)) * 2.0f;
                // utils::smallest_abs_diff_between_angles (utils/mod.rs:217-224)
                float s_err = ang_dist - mp.target_pos;
                float s_err_complement = s_err - copysignf(1.0f, s_err) * 6.28318530717958647692f;
                rhs_wo_bias += (fabsf(s_err) < fabsf(s_err_complement) ? s_err : s_err_complement) * mp.erp_inv_dt;
            }
            rhs_wo_bias += -mp.target_vel;
            JointRow &c = rows[len];
            c.impulse = 0.0f; c.bmin = -mp.max_impulse; c.bmax = mp.max_impulse;
            c.lin_jac = v3(0, 0, 0); c.ang_jac1 = ang_jac; c.ang_jac2 = ang_jac;
            c.ii1 = sym_mul(ii1, ang_jac);
            c.ii2 = sym_mul(ii2, ang_jac);
            c.inv_lhs = 0.0f; c.cfm_coeff = mp.cfm_coeff; c.cfm_gain = mp.cfm_gain;
            c.rhs = rhs_wo_bias; c.rhs_wo_bias = rhs_wo_bias;
            dof[len] = 12 + 3 + a;
            len++;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (!(motor & (1 << i))) continue;
            MotorParams mp = joint_motor_params(w, j, i, dt);
            JointRow &c = rows[len];
            c.impulse = 0.0f; c.bmin = -mp.max_impulse; c.bmax = mp.max_impulse;
            c.lin_jac = col[i];
            c.ang_jac1 = c1x * col[i].x + c1y * col[i].y + c1z * col[i].z;
            c.ang_jac2 = c2x * col[i].x + c2y * col[i].y + c2z * col[i].z;
            c.ii1 = sym_mul(ii1, c.ang_jac1);
            c.ii2 = sym_mul(ii2, c.ang_jac2);
            float rhs_wo_bias = 0.0f;
            float dist = dot(lin_err, c.lin_jac);
            if (mp.erp_inv_dt != 0.0f) rhs_wo_bias += (dist - mp.target_pos) * mp.erp_inv_dt;
            float target_vel = mp.target_vel;
            if (limited & (1 << i)) { float4 lp = w.j_lim[(size_t)i * w.n_joints + j]; float inv_dt = rp_inv(dt); target_vel = rp_clamp(target_vel, (lp.x - dist) * inv_dt, (lp.y - dist) * inv_dt); }
            rhs_wo_bias += -target_vel;
            c.inv_lhs = 0.0f; c.cfm_coeff = mp.cfm_coeff; c.cfm_gain = mp.cfm_gain;
            c.rhs = rhs_wo_bias; c.rhs_wo_bias = rhs_wo_bias;
            dof[len] = 12 + i;
            len++;
        }
        // (a coupled ANGULAR motor is a no-op in the reference: "TODO: coupled angular motor constraint")
        if (motor_all & coupled & 7) {
            // motor_linear_coupled (joint_constraint_helper.rs:333-408): ONE row along the combined error of the coupled linear axes; the
            // motor and limits of the first coupled linear axis (SpringJoint: LinX)
            const int fa = __ffs(coupled & 7) - 1;
            MotorParams mp = joint_motor_params(w, j, fa, dt);
            V3 lj = v3(0, 0, 0), aj1 = lj, aj2 = lj;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (!(coupled & (1 << i))) continue;
                const float coeff = dot(col[i], lin_err);
                lj = lj + col[i] * coeff;
                aj1 = aj1 + (c1x * col[i].x + c1y * col[i].y + c1z * col[i].z) * coeff;
                aj2 = aj2 + (c2x * col[i].x + c2y * col[i].y + c2z * col[i].z) * coeff;
            }
            const float dist = sqrtf(dot(lj, lj)), inv_dist = rp_inv(dist);
            lj = lj * inv_dist; aj1 = aj1 * inv_dist; aj2 = aj2 * inv_dist;
            float rhs_wo_bias = 0.0f;
            if (mp.erp_inv_dt != 0.0f) rhs_wo_bias += (dist - mp.target_pos) * mp.erp_inv_dt;
            float target_vel = mp.target_vel;
            if (limited_all & (1 << fa)) { float4 lp = w.j_lim[(size_t)fa * w.n_joints + j]; float inv_dt = rp_inv(dt); target_vel = rp_clamp(target_vel, (lp.x - dist) * inv_dt, (lp.y - dist) * inv_dt); }
            rhs_wo_bias += -target_vel;
            JointRow &c = rows[len];
            c.impulse = 0.0f; c.bmin = -mp.max_impulse; c.bmax = mp.max_impulse;
            c.lin_jac = lj; c.ang_jac1 = aj1; c.ang_jac2 = aj2;
            c.ii1 = sym_mul(ii1, aj1); c.ii2 = sym_mul(ii2, aj2);
            c.inv_lhs = 0.0f; c.cfm_coeff = mp.cfm_coeff; c.cfm_gain = mp.cfm_gain;
            c.rhs = rhs_wo_bias; c.rhs_wo_bias = rhs_wo_bias;
            dof[len] = 12 + fa;
            len++;
        }
        joint_finalize_store(w, j, rows, dof, len, 0, imsum, substep_id);
        base = len; len = 0;
    }
    if (locked & 0x38) {
        // locked angular axes — JointConstraintHelper::new (:129-139): ang_basis = diff_conj1_2_tr(q1, q2) * sgn, ang_err =
        // (q1^-1 q2) * sgn, sgn = copysign(1, q1 . q2); lock_angular (:628-673); RotationOps::diff_conj1_2 (utils/rotation_ops.rs:121-135)
        Q4 q1 = frame1.r, q2 = frame2.r;
        V3 v1 = v3(q1.x, q1.y, q1.z), v2 = v3(q2.x, q2.y, q2.z);
        float w1 = q1.w, w2 = q2.w;
        V3 u = v1 * w2 + v2 * w1;
        V3 cu[3] = {v3(0.0f, u.z, -u.y), v3(-u.z, 0.0f, u.x), v3(u.y, -u.x, 0.0f)};
        V3 ca[3] = {v3(0.0f, v1.z, -v1.y), v3(-v1.z, 0.0f, v1.x), v3(v1.y, -v1.x, 0.0f)};
        V3 cb[3] = {v3(0.0f, v2.z, -v2.y), v3(-v2.z, 0.0f, v2.x), v3(v2.y, -v2.x, 0.0f)};
        float d = w1 * w2;
        V3 dg[3] = {v3(d, 0.0f, 0.0f), v3(0.0f, d, 0.0f), v3(0.0f, 0.0f, d)};
        float v2c[3] = {v2.x, v2.y, v2.z};
        V3 M[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            V3 kron = v1 * v2c[c];
            V3 prod = ca[0] * cb[c].x + ca[1] * cb[c].y + ca[2] * cb[c].z;
            M[c] = (((kron + dg[c]) - cu[c]) + prod) * 0.5f;
        }
        float sgn = copysignf(1.0f, qdot(q1, q2));
        Q4 ang_err = qmul(qconj(q1), q2);
        float ang_err_imag[3] = {ang_err.x * sgn, ang_err.y * sgn, ang_err.z * sgn};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (!(locked & (8 << a))) continue;
            V3 ang_jac = a == 0 ? v3(M[0].x, M[1].x, M[2].x) : a == 1 ? v3(M[0].y, M[1].y, M[2].y) : v3(M[0].z, M[1].z, M[2].z);
            ang_jac = ang_jac * sgn;
            JointRow &c = rows[len];
            c.impulse = 0.0f;
            c.lin_jac = v3(0, 0, 0); c.ang_jac1 = ang_jac; c.ang_jac2 = ang_jac;
            float rhs_wo_bias = 0.0f;
            float rhs_bias = ang_err_imag[a] * w.prm.joint_erp_inv_dt;
            c.ii1 = sym_mul(ii1, ang_jac);
            c.ii2 = sym_mul(ii2, ang_jac);
            c.inv_lhs = 0.0f; c.cfm_coeff = w.prm.joint_cfm_coeff; c.cfm_gain = 0.0f; c.bmin = -JR_UNBOUNDED; c.bmax = JR_UNBOUNDED;
            c.rhs = rhs_wo_bias + rhs_bias; c.rhs_wo_bias = rhs_wo_bias;
            dof[len] = 3 + a;
            len++;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (!(locked & (1 << i))) continue;
        JointRow &c = rows[len];
        c.impulse = 0.0f;
        c.lin_jac = col[i];
        c.ang_jac1 = c1x * col[i].x + c1y * col[i].y + c1z * col[i].z;
        c.ang_jac2 = c2x * col[i].x + c2y * col[i].y + c2z * col[i].z;
        float rhs_wo_bias = 0.0f;
        float rhs_bias = dot(c.lin_jac, lin_err) * w.prm.joint_erp_inv_dt;
        c.ii1 = sym_mul(ii1, c.ang_jac1);
        c.ii2 = sym_mul(ii2, c.ang_jac2);
        c.inv_lhs = 0.0f; c.cfm_coeff = w.prm.joint_cfm_coeff; c.cfm_gain = 0.0f; c.bmin = -JR_UNBOUNDED; c.bmax = JR_UNBOUNDED;
        c.rhs = rhs_wo_bias + rhs_bias; c.rhs_wo_bias = rhs_wo_bias;
        dof[len] = i;
        len++;
    }
    if (limited_all) {
        // limited (free) axes: limit_angular rows, then limit_linear rows, then the coupled ones (joint_constraint_helper.rs:166-208, 468-564)
        const float maxcv = w.prm.max_corrective_velocity, erp = w.prm.joint_erp_inv_dt;
        const float inf = __int_as_float(0x7f800000);
        if (limited & 0x38) {
            Q4 q1 = frame1.r, q2 = frame2.r;
            float sgn = copysignf(1.0f, qdot(q1, q2));
            Q4 ang_err = qmul(qconj(q1), q2);
            float imag[3] = {ang_err.x * sgn, ang_err.y * sgn, ang_err.z * sgn}, real = ang_err.w * sgn;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (!(limited & (8 << a))) continue;
                float4 lp = w.j_lim[(size_t)(3 + a) * w.n_joints + j]; // c_cos, c_sin, half_range
                float x = imag[a];
                float sin_half = lp.x * x - lp.y * real;
                float cos_half = lp.x * real + lp.y * x;
                float half = rp_atan2_portable(sin_half, cos_half);
                float shift = copysignf(3.14159265358979323846f, half);
                float wrapped_half = fabsf(half) > 1.5707963267948966f ? half - shift : half;
                float ang = wrapped_half * 2.0f;
                bool min_enabled = ang <= -lp.z, max_enabled = lp.z <= ang;
                V3 ang_jac = col[a];
                JointRow &c = rows[len];
                c.impulse = 0.0f; c.bmin = min_enabled ? -inf : 0.0f; c.bmax = max_enabled ? inf : 0.0f;
                c.lin_jac = v3(0, 0, 0); c.ang_jac1 = ang_jac; c.ang_jac2 = ang_jac;
                float rhs_wo_bias = 0.0f;
                float rhs_bias = rp_clamp((rp_max(ang - lp.z, 0.0f) - rp_max(-lp.z - ang, 0.0f)) * erp, -maxcv, maxcv);
                c.ii1 = sym_mul(ii1, ang_jac);
                c.ii2 = sym_mul(ii2, ang_jac);
                c.inv_lhs = 0.0f; c.cfm_coeff = w.prm.joint_cfm_coeff; c.cfm_gain = 0.0f;
                c.rhs = rhs_wo_bias + rhs_bias; c.rhs_wo_bias = rhs_wo_bias;
                dof[len] = 6 + 3 + a;
                len++;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (!(limited & (1 << i))) continue;
            float4 lp = w.j_lim[(size_t)i * w.n_joints + j]; // min, max
            JointRow &c = rows[len];
            c.impulse = 0.0f;
            c.lin_jac = col[i];
            c.ang_jac1 = c1x * col[i].x + c1y * col[i].y + c1z * col[i].z;
            c.ang_jac2 = c2x * col[i].x + c2y * col[i].y + c2z * col[i].z;
            c.ii1 = sym_mul(ii1, c.ang_jac1);
            c.ii2 = sym_mul(ii2, c.ang_jac2);
            float dist = dot(lin_err, c.lin_jac);
            bool min_enabled = dist <= lp.x, max_enabled = lp.y <= dist;
            float rhs_wo_bias = 0.0f;
            float rhs_bias = rp_clamp((rp_max(dist - lp.y, 0.0f) - rp_max(lp.x - dist, 0.0f)) * erp, -maxcv, maxcv);
            c.inv_lhs = 0.0f; c.cfm_coeff = w.prm.joint_cfm_coeff; c.cfm_gain = 0.0f;
            c.rhs = rhs_wo_bias + rhs_bias; c.rhs_wo_bias = rhs_wo_bias;
            c.bmin = min_enabled ? -inf : 0.0f; c.bmax = max_enabled ? inf : 0.0f;
            dof[len] = 6 + i;
            len++;
        }
        if ((coupled & 0x38) && (limited_all & (1 << (__ffs(coupled & 0x38) - 1)))) {
            // limit_angular_coupled (joint_constraint_helper.rs:725-790): exactly two coupled angular axes; the angle between the two
            // frames' copies of the THIRD axis is limited (glam Quat::from_rotation_arc + to_axis_angle restated, as in the oracle)
            const int fa = __ffs(coupled & 0x38) - 1, ca = (coupled >> 3) & 7;
            const int nc = (ca & 1) ? ((ca & 2) ? 2 : 1) : 0;                       // trailing_ones: the angular axis that is NOT coupled
            float m2[3][3]; quat_to_mat(frame2.r, m2);
            const V3 axis1 = nc == 0 ? col[0] : nc == 1 ? col[1] : col[2];
            const V3 axis2 = nc == 0 ? v3(m2[0][0], m2[1][0], m2[2][0]) : nc == 1 ? v3(m2[0][1], m2[1][1], m2[2][1]) : v3(m2[0][2], m2[1][2], m2[2][2]);
            Q4 rot; const float d = dot(axis1, axis2);
            const float one_minus_eps = 1.0f - 2.0f * 1.1920929e-7f;
            if (d > one_minus_eps) rot = q4(0.0f, 0.0f, 0.0f, 1.0f);
            else if (d < -one_minus_eps) { // from_axis_angle(from.any_orthonormal_vector(), PI)
                const float sign = copysignf(1.0f, axis1.z), a = -1.0f / (sign + axis1.z), b = axis1.x * axis1.y * a;
                const V3 o = v3(b, sign + axis1.y * axis1.y * a, -axis1.y);
                rot = q4(o.x * 1.0f, o.y * 1.0f, o.z * 1.0f, -4.371139e-08f);
            } else {
                const V3 cr = cross(axis1, axis2);
                const Q4 q = q4(cr.x, cr.y, cr.z, 1.0f + d);
                const float inv = 1.0f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
                rot = q4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
            }
            const V3 rv = v3(rot.x, rot.y, rot.z);
            V3 ang_jac; float angle;
            const float rl = sqrtf(dot(rv, rv));
            if (rl >= 1.0e-8f) { angle = 2.0f * rp_atan2_portable(rl, rot.w); ang_jac = rv * (1.0f / rl); } else { ang_jac = v3(1.0f, 0.0f, 0.0f); angle = 0.0f; }
            if (angle == 0.0f) { // axis1.orthonormal_basis()[0] (utils/orthonormal_basis.rs:37-50)
                const float sign = copysignf(1.0f, axis1.z), a = -1.0f / (sign + axis1.z), b = axis1.x * axis1.y * a;
                ang_jac = v3(1.0f + sign * axis1.x * axis1.x * a, sign * b, -sign * axis1.x);
            }
            const float4 lm = w.j_lim[(size_t)fa * w.n_joints + j]; // (min, max): a COUPLED angular axis keeps its raw limits there (no per-axis row ever reads the recentred form)
            const bool min_enabled = angle <= lm.x, max_enabled = lm.y <= angle;
            JointRow &c = rows[len];
            c.impulse = 0.0f; c.bmin = min_enabled ? -inf : 0.0f; c.bmax = max_enabled ? inf : 0.0f;
            c.lin_jac = v3(0, 0, 0); c.ang_jac1 = ang_jac; c.ang_jac2 = ang_jac;
            const float rhs_bias = rp_clamp((rp_max(angle - lm.y, 0.0f) - rp_max(lm.x - angle, 0.0f)) * erp, -maxcv, maxcv);
            c.ii1 = sym_mul(ii1, ang_jac); c.ii2 = sym_mul(ii2, ang_jac);
            c.inv_lhs = 0.0f; c.cfm_coeff = w.prm.joint_cfm_coeff; c.cfm_gain = 0.0f;
            c.rhs = 0.0f + rhs_bias; c.rhs_wo_bias = 0.0f;
            dof[len] = 6 + fa;
            len++;
        }
        if ((coupled & 7) && (limited_all & (1 << (__ffs(coupled & 7) - 1)))) {
            // limit_linear_coupled (joint_constraint_helper.rs:210-283): the distance along the combined error of the coupled linear
            // axes against the MAX limit of the first coupled axis (RopeJoint; the reference: "FIXME: handle min limit too")
            const int fa = __ffs(coupled & 7) - 1;
            V3 lj = v3(0, 0, 0), aj1 = lj, aj2 = lj;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (!(coupled & (1 << i))) continue;
                const float coeff = dot(col[i], lin_err);
                lj = lj + col[i] * coeff;
                aj1 = aj1 + (c1x * col[i].x + c1y * col[i].y + c1z * col[i].z) * coeff;
                aj2 = aj2 + (c2x * col[i].x + c2y * col[i].y + c2z * col[i].z) * coeff;
            }
            const float dist = sqrtf(dot(lj, lj)), inv_dist = rp_inv(dist);
            lj = lj * inv_dist; aj1 = aj1 * inv_dist; aj2 = aj2 * inv_dist;
            const float lmax = w.j_lim[(size_t)fa * w.n_joints + j].y;
            const float rhs_wo_bias = rp_min(dist - lmax, 0.0f) * rp_inv(w.prm.dt_sub);
            const float rhs_bias = rp_clamp(rp_max(dist - lmax, 0.0f) * erp, -maxcv, maxcv);
            JointRow &c = rows[len];
            c.impulse = 0.0f; c.bmin = 0.0f; c.bmax = inf;
            c.lin_jac = lj; c.ang_jac1 = aj1; c.ang_jac2 = aj2;
            c.ii1 = sym_mul(ii1, aj1); c.ii2 = sym_mul(ii2, aj2);
            c.inv_lhs = 0.0f; c.cfm_coeff = w.prm.joint_cfm_coeff; c.cfm_gain = 0.0f;
            c.rhs = rhs_wo_bias + rhs_bias; c.rhs_wo_bias = rhs_wo_bias;
            dof[len] = 6 + fa;
            len++;
        }
    }
    joint_finalize_store(w, j, rows, dof, len, base, imsum, substep_id);
    JRP(JR_IM1, j) = f4(im1, 0.0f); JRP(JR_IM2, j) = f4(im2, 0.0f);
}
RP_DEV void joint_update_one(const DevWorld &w, int j, int substep_id) { PlainBodyIO io = {w}; joint_update_one_t(w, io, j, substep_id); }

// All rows of joint j: [remove bias] [warm start] solve — solve_joint, staged_island_solver/solve.rs:31-47.
// The rows are walked CH at a time: the CH rows of a chunk are fetched together before the first of them is solved (a row's store
// would otherwise hold back the next row's loads: one exposed round trip per row), and a caller that knows its joint early
// (the dataflow launch: while it waits for its bodies' tickets) fetches the first chunk itself — jrows_load + joint_solve_fetched.
template <int CH> struct JointRowsT { JointRow c[CH]; };
template <int CH>
RP_DEV void jrows_load(const DevWorld &w, int j, int r0, int nrows, JointRowsT<CH> &R) {
#pragma unroll
    for (int q = 0; q < CH; ++q) if (r0 + q < nrows) jrow_load(w, j, r0 + q, R.c[q]);
}
template <int CH>
RP_DEV void jrows_solve(const DevWorld &w, int j, int r0, int nrows, JointRowsT<CH> &R, V3 im1, V3 im2, int b1, int b2, V3 &l1, V3 &a1, V3 &l2, V3 &a2, bool wo_bias, bool warmstart, int jm_out, bool jm_store) {
#pragma unroll
    for (int q = 0; q < CH; ++q) {
        if (r0 + q >= nrows) break;
        const int r = r0 + q;
        JointRow &c = R.c[q];
        if (wo_bias) c.rhs = c.rhs_wo_bias;
        if (warmstart) {
            V3 lin_impulse = c.lin_jac * c.impulse;
            V3 i1 = c.ii1 * c.impulse, i2 = c.ii2 * c.impulse;
            l1 = l1 + cmul(lin_impulse, im1); a1 = a1 + i1;
            l2 = l2 - cmul(lin_impulse, im2); a2 = a2 - i2;
            // a world-attached side reloads zeros for every row (gather of solver id u32::MAX)
            if (b1 < 0) { l1 = v3(0, 0, 0); a1 = l1; }
            if (b2 < 0) { l2 = v3(0, 0, 0); a2 = l2; }
        }
        float dlinvel = dot(c.lin_jac, l2 - l1);
        float dangvel = dot(c.ang_jac2, a2) - dot(c.ang_jac1, a1);
        float rhs = dlinvel + dangvel + c.rhs;
        float total = c.impulse + c.inv_lhs * (rhs - c.cfm_gain * c.impulse);
        total = rp_clamp(total, c.bmin, c.bmax); // lock rows: +-f32::MAX; limit rows: one-sided or zero
        float delta = total - c.impulse;
        c.impulse = total;
        V3 lin_impulse = c.lin_jac * delta;
        V3 i1 = c.ii1 * delta, i2 = c.ii2 * delta;
        l1 = l1 + cmul(lin_impulse, im1); a1 = a1 + i1;
        l2 = l2 - cmul(lin_impulse, im2); a2 = a2 - i2;
        if (b1 < 0) { l1 = v3(0, 0, 0); a1 = l1; }
        if (b2 < 0) { l2 = v3(0, 0, 0); a2 = l2; }
        // only the mutable words of the row go back
        if (w.jm) { if (jm_store) jm_put(w, j, r, jm_out, c.impulse, c.rhs); }
        else { JRR(r, JR_LIN, j).w = c.impulse; if (wo_bias) JRR(r, JR_A2, j).w = c.rhs; }
    }
}
// rows [0, CH) already fetched into R0 (with nrows, im1, im2) by the caller
// (b1, b2: the joint's bodies, already known to the caller — a tile sweep fetches them with the rows, ahead of the stage loop)
template <class IO, int CH>
RP_DEV void joint_solve_fetched(const DevWorld &w, const IO &io, int j, int b1, int b2, int nrows, V3 im1, V3 im2, JointRowsT<CH> &R0, bool wo_bias, bool warmstart) {
    V3 l1 = v3(0, 0, 0), a1 = l1, l2 = l1, a2 = l1;
    if (b1 >= 0) io.load_vel(0, b1, l1, a1);
    if (b2 >= 0) io.load_vel(1, b2, l2, a2);
    const int jo = io.jm_out(w); const bool js = io.jm_store();
    jrows_solve<CH>(w, j, 0, nrows, R0, im1, im2, b1, b2, l1, a1, l2, a2, wo_bias, warmstart, jo, js);
    for (int r0 = CH; r0 < nrows; r0 += CH) { JointRowsT<CH> R; jrows_load<CH>(w, j, r0, nrows, R); jrows_solve<CH>(w, j, r0, nrows, R, im1, im2, b1, b2, l1, a1, l2, a2, wo_bias, warmstart, jo, js); }
    if (b1 >= 0) io.store_vel(0, b1, l1, a1);
    if (b2 >= 0) io.store_vel(1, b2, l2, a2);
}
template <class IO, int CH = 1>
RP_DEV void joint_solve_one_t(const DevWorld &w, const IO &io, int j, bool wo_bias, bool warmstart) {
    int nrows = joint_row_count(w.j_locked[j], w.j_limited[j], w.j_motor[j]);
    V3 im1 = v3(JRP(JR_IM1, j)), im2 = v3(JRP(JR_IM2, j));
    JointRowsT<CH> R0; jrows_load<CH>(w, j, 0, nrows, R0);
    joint_solve_fetched<IO, CH>(w, io, j, w.j_b1[j], w.j_b2[j], nrows, im1, im2, R0, wo_bias, warmstart);
}
RP_DEV void joint_solve_one(const DevWorld &w, int j, bool wo_bias, bool warmstart) { PlainBodyIO io = {w}; joint_solve_one_t(w, io, j, wo_bias, warmstart); }

// JointConstraint::writeback_impulses — joint_velocity_constraint.rs:346-353
RP_DEV void joint_writeback_one(const DevWorld &w, int j) {
    int locked = w.j_locked[j], limited = w.j_limited[j] & ~locked, motor = w.j_motor[j] & ~locked;
    float imp[18];
    float4 old = w.j_imp[j], olda = w.j_imp_ang[j];
    imp[0] = old.x; imp[1] = old.y; imp[2] = old.z; imp[3] = olda.x; imp[4] = olda.y; imp[5] = olda.z;
    { float4 l = w.j_imp_lim[j], la = w.j_imp_lim_ang[j]; imp[6] = l.x; imp[7] = l.y; imp[8] = l.z; imp[9] = la.x; imp[10] = la.y; imp[11] = la.z; }
    { float4 m = w.j_imp_mot[j], ma = w.j_imp_mot_ang[j]; imp[12] = m.x; imp[13] = m.y; imp[14] = m.z; imp[15] = ma.x; imp[16] = ma.y; imp[17] = ma.z; }
    int nrows = joint_row_count(locked, limited, motor);
    for (int k = 0; k < nrows; ++k) imp[joint_row_dof(locked, limited, motor, k)] = jrow_impulse(w, j, k);
    w.j_imp[j] = make_float4(imp[0], imp[1], imp[2], 0.0f);
    w.j_imp_ang[j] = make_float4(imp[3], imp[4], imp[5], 0.0f);
    if (limited) { w.j_imp_lim[j] = make_float4(imp[6], imp[7], imp[8], 0.0f); w.j_imp_lim_ang[j] = make_float4(imp[9], imp[10], imp[11], 0.0f); } // JointLimits::impulse
    if (motor) { w.j_imp_mot[j] = make_float4(imp[12], imp[13], imp[14], 0.0f); w.j_imp_mot_ang[j] = make_float4(imp[15], imp[16], imp[17], 0.0f); } // JointMotor::impulse
}

// One sweep over the joints inside a single workgroup (SINGLE mode and the serial tail): parallel joint
// colours from `first` one after the other, then the overflow list on lane 0.
RP_DEV void joint_tail_sweep(const DevWorld &w, int first, bool wo_bias, bool warmstart) {
    if (w.n_joints == 0) return;
    int nst = w.flags[FL_NJ_STAGES];
    for (int st = first; st < nst; ++st) {
        int beg = w.j_stage_begin[st], cnt = w.j_stage_count[st];
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) joint_solve_one(w, w.j_order[beg + i], wo_bias, warmstart);
        __threadfence();
        __syncthreads();
    }
    int ob = w.flags[FL_NJ_OVF_BEGIN], oc = w.flags[FL_NJ_OVF_COUNT];
    if (oc > 0) {
        if (threadIdx.x == 0) for (int i = 0; i < oc; ++i) { joint_solve_one(w, w.j_order[ob + i], wo_bias, warmstart); __threadfence(); }
        __threadfence();
        __syncthreads();
    }
}
