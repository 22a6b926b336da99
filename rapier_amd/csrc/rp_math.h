// rp_math.h — device-side f32 vector / quaternion helpers for the gfx950 kernels.
// IEEE semantics on purpose: the library is built with -ffp-contract=off (the reference forbids
// FMA/recip in the solver, /root/reference/run-ci-checks.sh:74-83) and HIP's default correctly
// rounded f32 divide/sqrt.  `rp_inv` = utils::simd_inv (/root/reference/src/utils/mod.rs:131-146).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RP_DEV __device__ __forceinline__
#define RP_HD __host__ __device__ __forceinline__

struct V3 { float x, y, z; };
struct Q4 { float x, y, z, w; };
struct Sym3 { float m11, m12, m13, m22, m23, m33; };

RP_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RP_HD V3 v3(const float4 &f) { V3 r; r.x = f.x; r.y = f.y; r.z = f.z; return r; }
RP_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RP_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RP_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RP_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RP_HD V3 cmul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RP_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RP_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RP_HD float len2(V3 a) { return dot(a, a); }
RP_HD float len(V3 a) { return sqrtf(dot(a, a)); }
RP_HD float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
RP_HD float4 f4(V3 a, float w) { return make_float4(a.x, a.y, a.z, w); }
RP_HD float rp_inv(float x) { return (x > -1.0e-20f && x < 1.0e-20f) ? 0.0f : 1.0f / x; }
RP_HD float rp_max(float a, float b) { return a > b ? a : b; }
// utils::canonicalize_zero (utils/mod.rs:80-102): x + 0.0 turns -0.0 into +0.0, every other value is untouched (IEEE: not foldable)
RP_HD float rp_canon0(float x) { return x + 0.0f; }
RP_HD float rp_min(float a, float b) { return a < b ? a : b; }
RP_HD float rp_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// OrthonormalBasis (Pixar) — /root/reference/src/utils/orthonormal_basis.rs:77-93
RP_HD V3 orthonormal_vector(V3 n) {
    float sign = copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    return v3(b, sign + n.y * n.y * a, -n.y);
}
RP_HD void orthonormal_basis(V3 n, V3 &b0, V3 &b1) {
    float sign = copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    b0 = v3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    b1 = v3(b, sign + n.y * n.y * a, -n.y);
}

RP_HD Q4 q4(float x, float y, float z, float w) { Q4 q; q.x = x; q.y = y; q.z = z; q.w = w; return q; }
RP_HD Q4 q4(const float4 &f) { return q4(f.x, f.y, f.z, f.w); }
RP_HD float4 f4(Q4 q) { return make_float4(q.x, q.y, q.z, q.w); }
RP_HD Q4 qmul(Q4 a, Q4 b) {
    return q4(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
              a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
RP_HD Q4 qconj(Q4 a) { return q4(-a.x, -a.y, -a.z, a.w); }
RP_HD float qdot(Q4 a, Q4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
RP_HD Q4 qnormalize(Q4 a) {
    float inv = 1.0f / sqrtf(qdot(a, a));
    return q4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
}
RP_HD V3 qrot(Q4 q, V3 v) {
    V3 b = v3(q.x, q.y, q.z);
    float b2 = dot(b, b);
    return v * (q.w * q.w - b2) + b * (dot(v, b) * 2.0f) + cross(b, v) * (q.w * 2.0f);
}
RP_HD V3 qrot_inv(Q4 q, V3 v) { return qrot(qconj(q), v); }

struct Pose { Q4 r; V3 t; };
RP_HD Pose pose_mul(Pose a, Pose b) { Pose r; r.r = qmul(a.r, b.r); r.t = qrot(a.r, b.t) + a.t; return r; }
RP_HD Pose pose_inv(Pose a) { Pose r; r.r = qconj(a.r); r.t = qrot(r.r, -a.t); return r; }
RP_HD Pose pose_inv_mul(Pose a, Pose b) { Pose r; Q4 ai = qconj(a.r); r.r = qmul(ai, b.r); r.t = qrot(ai, b.t - a.t); return r; }
RP_HD V3 pose_tp(Pose a, V3 p) { return qrot(a.r, p) + a.t; }
RP_HD V3 pose_itp(Pose a, V3 p) { return qrot_inv(a.r, p - a.t); }

// translation / rotation locking of update_world_mass_properties (rigid_body_components.rs:533-571); `la` = LockedAxes bits
RP_HD void apply_locked_rotations(int la, Sym3 &ii) {
    if (la & 8) { ii.m11 = 0.0f; ii.m12 = 0.0f; ii.m13 = 0.0f; }
    if (la & 16) { ii.m22 = 0.0f; ii.m12 = 0.0f; ii.m23 = 0.0f; }
    if (la & 32) { ii.m33 = 0.0f; ii.m13 = 0.0f; ii.m23 = 0.0f; }
}

// Portable single-precision atan (Cephes atanf scheme, only + - * /): identical to oracle/ro_math.h so both sides
// agree bit for bit (libm's and ocml's atan2f do not).
RP_HD float rp_atan_portable(float x) {
    float sign = 1.0f; if (x < 0.0f) { sign = -1.0f; x = -x; }
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return sign * y;
}
RP_HD float rp_atan2_pos(float y, float x) { // y >= 0
    if (x > 0.0f) return rp_atan_portable(y / x);
    if (x < 0.0f) return 3.14159265358979323846f + rp_atan_portable(y / x);
    return y > 0.0f ? 1.5707963267948966f : 0.0f;
}
RP_HD float rp_atan2_portable(float y, float x) { // full range
    if (x > 0.0f) return rp_atan_portable(y / x);
    if (x < 0.0f) return y >= 0.0f ? rp_atan_portable(y / x) + 3.14159265358979323846f : rp_atan_portable(y / x) - 3.14159265358979323846f;
    return y > 0.0f ? 1.5707963267948966f : (y < 0.0f ? -1.5707963267948966f : 0.0f);
}
// direction of a capsule's segment (ColliderBuilder::capsule_x / capsule_y / capsule_z)
RP_HD V3 capsule_axis_dir(int axis) { return axis == 0 ? v3(1, 0, 0) : axis == 2 ? v3(0, 0, 1) : v3(0, 1, 0); }
// asin(x) for |x| <= 1 from the portable atan: atan2(x, sqrt((1 - x)(1 + x)))
RP_HD float rp_asin_portable(float x) { return rp_atan2_portable(x, sqrtf((1.0f - x) * (1.0f + x))); }
// Quat::to_scaled_axis: axis * angle, angle = 2 atan2(|v|, w)
RP_HD V3 quat_to_scaled_axis(Q4 q) {
    V3 v = v3(q.x, q.y, q.z);
    float length = len(v);
    if (length >= 1.0e-8f) { float angle = 2.0f * rp_atan2_pos(length, q.w); return (v * (1.0f / length)) * angle; }
    return v3(0, 0, 0);
}

RP_HD V3 sym_mul(Sym3 m, V3 v) {
    return v3(m.m11 * v.x + m.m12 * v.y + m.m13 * v.z, m.m12 * v.x + m.m22 * v.y + m.m23 * v.z,
              m.m13 * v.x + m.m23 * v.y + m.m33 * v.z);
}
RP_HD void quat_to_mat(Q4 q, float r[3][3]) {
    float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2;
    float yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
    float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    r[0][0] = 1.0f - (yy + zz); r[0][1] = xy - wz; r[0][2] = xz + wy;
    r[1][0] = xy + wz; r[1][1] = 1.0f - (xx + zz); r[1][2] = yz - wx;
    r[2][0] = xz - wy; r[2][1] = yz + wx; r[2][2] = 1.0f - (xx + yy);
}
// parry MassProperties::world_inv_inertia: R diag(inv_pi) R^T
RP_HD Sym3 world_inv_inertia(V3 inv_pi, Q4 frame, Q4 rot) {
    Sym3 r = {0, 0, 0, 0, 0, 0};
    if (inv_pi.x == 0.0f && inv_pi.y == 0.0f && inv_pi.z == 0.0f) return r;
    float m[3][3];
    quat_to_mat(qmul(rot, frame), m);
#define RP_E(i, j) (m[i][0] * inv_pi.x * m[j][0] + m[i][1] * inv_pi.y * m[j][1] + m[i][2] * inv_pi.z * m[j][2])
    r.m11 = RP_E(0, 0); r.m12 = RP_E(0, 1); r.m13 = RP_E(0, 2); r.m22 = RP_E(1, 1); r.m23 = RP_E(1, 2); r.m33 = RP_E(2, 2);
#undef RP_E
    return r;
}
