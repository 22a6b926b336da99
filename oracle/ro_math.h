/*
 * oracle/ro_math.h — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference's math).
 *
 * Scalar f32 vector / quaternion / pose helpers used by the oracle.  They restate the
 * plain-IEEE semantics of the reference's math stack (glamx / nalgebra):
 * no FMA (build with -ffp-contract=off), IEEE div and sqrt, `inv(x)=0 if |x|<1e-20`
 * (/root/reference/src/utils/mod.rs:131-146), Pixar orthonormal vector
 * (/root/reference/src/utils/orthonormal_basis.rs:77-93).
 *
 * Nothing under rapier_amd/ may include this file.
 */
#ifndef RO_MATH_H
#define RO_MATH_H
#include <math.h>
#include <stdint.h>

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } quat;            /* glam layout: (x,y,z,w) */
typedef struct { quat r; v3 t; } pose;                /* rotation then translation */
typedef struct { float m11, m12, m13, m22, m23, m33; } sym3; /* parry SdpMatrix3 */

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
/* utils::canonicalize_zero (utils/mod.rs:80-102): x + 0.0 turns -0.0 into +0.0 and leaves every other value untouched */
static inline float canon0(float x) { volatile float z = 0.0f; return x + z; }
static inline v3 vcanon(v3 a) { return V3(canon0(a.x), canon0(a.y), canon0(a.z)); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 vcmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vlen2(v3 a) { return vdot(a, a); }
static inline float vlen(v3 a) { return sqrtf(vdot(a, a)); }
static inline float vget(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline void vset(v3 *a, int i, float v) { if (i == 0) a->x = v; else if (i == 1) a->y = v; else a->z = v; }

/* utils::inv / simd_inv — /root/reference/src/utils/mod.rs:131-146 */
static inline float ro_inv(float x) { return (x > -1.0e-20f && x < 1.0e-20f) ? 0.0f : 1.0f / x; }

/* OrthonormalBasis::orthonormal_vector — orthonormal_basis.rs:88-93 */
static inline v3 orthonormal_vector(v3 n) {
    float sign = copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    return V3(b, sign + n.y * n.y * a, -n.y);
}
/* OrthonormalBasis::orthonormal_basis — orthonormal_basis.rs:77-86 */
static inline void orthonormal_basis(v3 n, v3 out[2]) {
    float sign = copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    out[0] = V3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    out[1] = V3(b, sign + n.y * n.y * a, -n.y);
}

static inline quat Q(float x, float y, float z, float w) { quat q = {x, y, z, w}; return q; }
static inline quat qident(void) { return Q(0, 0, 0, 1); }
static inline quat qmul(quat a, quat b) {
    return Q(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
             a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w,
             a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
static inline quat qconj(quat a) { return Q(-a.x, -a.y, -a.z, a.w); }
static inline float qdot(quat a, quat b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline quat qnormalize(quat a) {
    float inv = 1.0f / sqrtf(qdot(a, a));
    return Q(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
}
/* glam Quat::mul_vec3 (scalar path) */
static inline v3 qrot(quat q, v3 v) {
    v3 b = V3(q.x, q.y, q.z);
    float b2 = vdot(b, b);
    return vadd(vadd(vmul(v, q.w * q.w - b2), vmul(b, vdot(v, b) * 2.0f)), vmul(vcross(b, v), q.w * 2.0f));
}
static inline v3 qrot_inv(quat q, v3 v) { return qrot(qconj(q), v); }

static inline pose pose_mul(pose a, pose b) { pose r; r.r = qmul(a.r, b.r); r.t = vadd(qrot(a.r, b.t), a.t); return r; }
static inline pose pose_inv(pose a) { pose r; r.r = qconj(a.r); r.t = qrot(r.r, vneg(a.t)); return r; }
/* a^-1 * b */
static inline pose pose_inv_mul(pose a, pose b) {
    pose r; quat ai = qconj(a.r);
    r.r = qmul(ai, b.r); r.t = qrot(ai, vsub(b.t, a.t)); return r;
}
static inline v3 pose_tp(pose a, v3 p) { return vadd(qrot(a.r, p), a.t); }
static inline v3 pose_itp(pose a, v3 p) { return qrot_inv(a.r, vsub(p, a.t)); }
static inline pose pose_ident(void) { pose p; p.r = qident(); p.t = V3(0, 0, 0); return p; }

/* Portable single-precision atan (Cephes atanf scheme: two range reductions + a degree-9 odd polynomial, only
 * + - * /), so that the oracle and the HIP kernels (rp_math.h: rp_atan_portable) agree bit for bit; libm's and
 * ocml's atan2f are each within 1 ulp of it but not of each other. */
static inline float ro_atan_portable(float x) {
    float sign = 1.0f; if (x < 0.0f) { sign = -1.0f; x = -x; }
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return sign * y;
}
/* atan2(y, x) for y >= 0 */
static inline float ro_atan2_pos(float y, float x) {
    if (x > 0.0f) return ro_atan_portable(y / x);
    if (x < 0.0f) return 3.14159265358979323846f + ro_atan_portable(y / x);
    return y > 0.0f ? 1.5707963267948966f : 0.0f;
}
/* atan2(y, x), full range, from the portable atan */
static inline float ro_atan2_portable(float y, float x) {
    if (x > 0.0f) return ro_atan_portable(y / x);
    if (x < 0.0f) return y >= 0.0f ? ro_atan_portable(y / x) + 3.14159265358979323846f : ro_atan_portable(y / x) - 3.14159265358979323846f;
    return y > 0.0f ? 1.5707963267948966f : (y < 0.0f ? -1.5707963267948966f : 0.0f);
}
/* asin(x) for |x| <= 1 from the portable atan: atan2(x, sqrt((1 - x)(1 + x))) */
static inline float ro_asin_portable(float x) { return ro_atan2_portable(x, sqrtf((1.0f - x) * (1.0f + x))); }
/* Quat::to_scaled_axis: axis * angle, angle = 2 atan2(|v|, w) */
static inline v3 quat_to_scaled_axis(quat q) {
    v3 v = V3(q.x, q.y, q.z);
    float length = vlen(v);
    if (length >= 1.0e-8f) { float angle = 2.0f * ro_atan2_pos(length, q.w); return vmul(vmul(v, 1.0f / length), angle); }
    return V3(0, 0, 0);
}

static inline v3 sym3_mul(sym3 m, v3 v) {
    return V3(m.m11 * v.x + m.m12 * v.y + m.m13 * v.z,
              m.m12 * v.x + m.m22 * v.y + m.m23 * v.z,
              m.m13 * v.x + m.m23 * v.y + m.m33 * v.z);
}
/* glam Mat3::from_quat, returned as rows r[i][j] */
static inline void quat_to_mat(quat q, float r[3][3]) {
    float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2;
    float yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
    float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    r[0][0] = 1.0f - (yy + zz); r[0][1] = xy - wz;          r[0][2] = xz + wy;
    r[1][0] = xy + wz;          r[1][1] = 1.0f - (xx + zz); r[1][2] = yz - wx;
    r[2][0] = xz - wy;          r[2][1] = yz + wx;          r[2][2] = 1.0f - (xx + yy);
}
static inline float ro_clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline float ro_maxf(float a, float b) { return a > b ? a : b; }
static inline float ro_minf(float a, float b) { return a < b ? a : b; }
#endif
