// ik_math.hpp -- device-side f64 Lie-group math for the batched IK kernels (gfx950).
//
// One configuration per lane; every value lives in VGPRs, chain constants come
// from LDS.  The operation ORDER of each function is part of the contract: it
// is the same sequence of IEEE-754 + - * / sqrt the CPU oracle executes
// (oracle/optik_oracle.c), compiled with -ffp-contract=off on both sides, so a
// kernel result can be compared bit-for-bit with the oracle.  Identical
// sub-expressions the reference recomputes (so3::log three times per
// evaluation, so3::right_jacobian twice) are computed once here.
//
// Reference being restated: /root/reference/crates/optik/src/math.rs,
// kinematics.rs:123-196, 243-255, objective.rs:7-110.
#pragma once

#include "ik_platform.hpp"  // (the HIP runtime header -- or, for tests/emu only, its host emulation)

namespace optik {

#define OPTIK_DEV __device__ __forceinline__

struct V3 { double x, y, z; };
struct Q4 { double i, j, k, w; };      // nalgebra storage order [i, j, k, w]
struct Pose { V3 t; Q4 q; };
struct M3 { double m[3][3]; };         // row-major

OPTIK_DEV V3 cross(const V3 a, const V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// nalgebra Quaternion product.
OPTIK_DEV Q4 qmul(const Q4 a, const Q4 b) {
    Q4 o;
    o.w = a.w * b.w - a.i * b.i - a.j * b.j - a.k * b.k;
    o.i = a.w * b.i + a.i * b.w + a.j * b.k - a.k * b.j;
    o.j = a.w * b.j - a.i * b.k + a.j * b.w + a.k * b.i;
    o.k = a.w * b.k + a.i * b.j - a.j * b.i + a.k * b.w;
    return o;
}

OPTIK_DEV Q4 qconj(const Q4 a) { return Q4{-a.i, -a.j, -a.k, a.w}; }

// nalgebra UnitQuaternion * Vector3:  t = 2 (v x r);  r' = w t + v x t + r.
OPTIK_DEV V3 qrot(const Q4 q, const V3 r) {
    const V3 v{q.i, q.j, q.k};
    V3 t = cross(v, r);
    t.x *= 2.0; t.y *= 2.0; t.z *= 2.0;
    const V3 c = cross(v, t);
    return V3{t.x * q.w + c.x + r.x, t.y * q.w + c.y + r.y, t.z * q.w + c.z + r.z};
}

OPTIK_DEV Pose pose_mul(const Pose a, const Pose b) {
    const V3 s = qrot(a.q, b.t);
    Pose o;
    o.t = V3{a.t.x + s.x, a.t.y + s.y, a.t.z + s.z};
    o.q = qmul(a.q, b.q);
    return o;
}

// Isometry3::inv_mul (objective.rs:49,70).
OPTIK_DEV Pose pose_inv_mul(const Pose a, const Pose b) {
    const Q4 qc = qconj(a.q);
    const V3 d{b.t.x - a.t.x, b.t.y - a.t.y, b.t.z - a.t.z};
    Pose o;
    o.t = qrot(qc, d);
    o.q = qmul(qc, b.q);
    return o;
}

// Small per-lane vectors are LLVM vector values, not arrays: element selection by a
// per-lane index then stays a chain of v_cndmask on registers.  (With C arrays the
// optimiser rewrites select(load a[i], load a[j]) into a load through a selected
// pointer, which forces the whole array into scratch / LDS.)
typedef double dvec8 __attribute__((ext_vector_type(8)));

OPTIK_DEV double vpick(const dvec8 a, int idx) {  // a[idx-1], idx per lane, 1-based
    double v = a[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) v = (idx == i + 1) ? a[i] : v;
    return v;
}

OPTIK_DEV void vput(dvec8 &a, int idx, double v) {  // a[idx-1] = v
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (idx == i + 1) ? v : a[i];
}

OPTIK_DEV dvec8 vsel(bool c, const dvec8 a, const dvec8 b) {
    dvec8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = c ? a[i] : b[i];
    return o;
}

// ... and for the 9 rows of an 8-DoF chain's dual problem (single-kernel path only)
typedef double dvec16 __attribute__((ext_vector_type(16)));

OPTIK_DEV double vpick(const dvec16 a, int idx) {
    double v = a[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) v = (idx == i + 1) ? a[i] : v;
    return v;
}

OPTIK_DEV void vput(dvec16 &a, int idx, double v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (idx == i + 1) ? v : a[i];
}

OPTIK_DEV dvec16 vsel(bool c, const dvec16 a, const dvec16 b) {
    dvec16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = c ? a[i] : b[i];
    return o;
}

template <bool SMALL> struct RowVecOf { typedef dvec8 type; };
template <> struct RowVecOf<false> { typedef dvec16 type; };

// ---- elementary functions (same operation sequence as the oracle) ---------

OPTIK_DEV double k_sin(double x, double y) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    const double w = z * z;
    const double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    const double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

OPTIK_DEV double k_cos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    double w = z * z;
    const double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    const double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

// sin/cos of a joint-sized angle: Cody-Waite with a three-part pi/2, then the
// fdlibm kernels; quadrant by select (no divergence).
OPTIK_DEV void sincos_dev(double x, double &s, double &c) {
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const double big = 6755399441055744.0;  // 1.5 * 2^52
    double fn = x * invpio2 + big;
    fn = fn - big;
    const int n = (int)fn;
    double r = x - fn * pio2_1;
    const double t = r;
    const double w2 = fn * pio2_2;
    r = t - w2;
    const double w = fn * pio2_2t - ((t - r) - w2);
    const double y0 = r - w;
    const double y1 = (r - y0) - w;
    const double ks = k_sin(y0, y1), kc = k_cos(y0, y1);
    const bool swap = (n & 1) != 0;
    const double sv = swap ? kc : ks;
    const double cv = swap ? ks : kc;
    s = (n & 2) ? -sv : sv;
    c = (((n + 1) & 2) != 0) ? -cv : cv;
}

// atan2(y, x), y > 0, x >= 0 (math.rs:54 after the w >= 0 flip).
OPTIK_DEV double atan2_q1(double y, double x) {
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
    const double t = y / x;
    double num, den, hi, lo;
    const bool direct = t < 0.4375;
    if (direct) { num = t; den = 1.0; hi = 0.0; lo = 0.0; }
    else if (t < 0.6875) { num = 2.0 * t - 1.0; den = 2.0 + t;
        hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
    else if (t < 1.1875) { num = t - 1.0; den = t + 1.0;
        hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
    else if (t < 2.4375) { num = t - 1.5; den = 1.0 + 1.5 * t;
        hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
    else { num = -1.0; den = t;
        hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
    const double u = num / den;
    const double z = u * u;
    const double w = z * z;
    const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    return direct ? (u - u * (s1 + s2)) : (hi - ((u * (s1 + s2) - lo) - u));
}

// ---- math.rs ----------------------------------------------------------------

constexpr double EPSILON = 1e-6;  // math.rs:7

// so3::hat (math.rs:13) and hat_2 (math.rs:19-31), row-major.
OPTIK_DEV M3 hat(const V3 w) {
    M3 M;
    M.m[0][0] = 0.0;  M.m[0][1] = -w.z; M.m[0][2] = w.y;
    M.m[1][0] = w.z;  M.m[1][1] = 0.0;  M.m[1][2] = -w.x;
    M.m[2][0] = -w.y; M.m[2][1] = w.x;  M.m[2][2] = 0.0;
    return M;
}

OPTIK_DEV M3 hat_2(const V3 w) {
    const double w11 = w.x * w.x, w12 = w.x * w.y, w13 = w.x * w.z;
    const double w22 = w.y * w.y, w23 = w.y * w.z, w33 = w.z * w.z;
    M3 M;
    M.m[0][0] = -w22 - w33; M.m[0][1] = w12;        M.m[0][2] = w13;
    M.m[1][0] = w12;        M.m[1][1] = -w11 - w33; M.m[1][2] = w23;
    M.m[2][0] = w13;        M.m[2][1] = w23;        M.m[2][2] = -w11 - w22;
    return M;
}

// so3::log, math.rs:40-63.
OPTIK_DEV V3 so3_log(const Q4 q) {
    const bool pos = q.w >= 0.0;
    const double w = pos ? q.w : -q.w;
    const V3 v{pos ? q.i : -q.i, pos ? q.j : -q.j, pos ? q.k : -q.k};
    const double v_norm_2 = v.x * v.x + v.y * v.y + v.z * v.z;
    double theta_over_v_norm;
    if (v_norm_2 > EPSILON) {
        const double v_norm = __builtin_sqrt(v_norm_2);
        theta_over_v_norm = atan2_q1(v_norm, w) / v_norm;
    } else {
        theta_over_v_norm = 1. / w - 1. / (3. * (w * w * w)) * v_norm_2
                            + 1. / (5. * (w * w * w * w * w)) * (v_norm_2 * v_norm_2);
    }
    return V3{2.0 * v.x * theta_over_v_norm, 2.0 * v.y * theta_over_v_norm,
              2.0 * v.z * theta_over_v_norm};
}

// Everything se3::log, so3::right_jacobian and the q-matrix share for one error
// rotation w: theta, sin, cos are computed once (bit-identical to recomputing).
struct RotTerms {
    V3 w;
    double theta_2, theta, s, c;
};

OPTIK_DEV RotTerms rot_terms(const V3 w) {
    RotTerms r;
    r.w = w;
    r.theta_2 = w.x * w.x + w.y * w.y + w.z * w.z;
    r.theta = __builtin_sqrt(r.theta_2);
    sincos_dev(r.theta, r.s, r.c);
    return r;
}

// so3::right_jacobian, math.rs:72-94 (theta_2 == 0 uses the limit 1/6, quirk Q3).
OPTIK_DEV M3 so3_right_jacobian(const RotTerms &r) {
    const double theta_2 = r.theta_2, theta_4 = theta_2 * theta_2;
    const bool big = theta_2 > EPSILON;
    const double a = big ? r.s / r.theta : 1. - 1. / 6. * theta_2 + 1. / 120.0 * theta_4;
    const double b = big ? (1. - r.c) / theta_2 : 1. / 2. - 1. / 24. * theta_2 + 1. / 720. * theta_4;
    const double cc = (theta_2 > 0.0) ? (1. - a) / theta_2 : 1. / 6.;
    const double e = (b - 2. * cc) / (2. * a);
    const M3 H = hat(r.w), H2 = hat_2(r.w);
    M3 J;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            J.m[i][k] = ((i == k) ? 1.0 : 0.0) + 0.5 * H.m[i][k] + e * H2.m[i][k];
    return J;
}

// se3::log, math.rs:107-124: returns the linear part V^-1 t (angular part = w).
OPTIK_DEV V3 se3_log_linear(const RotTerms &r, const V3 t) {
    const double theta_sq = r.theta_2;
    double p;
    if (r.theta > EPSILON) p = 0.5 * (r.theta * r.s) / (1. - r.c);
    else p = 1. - theta_sq / 12. - theta_sq * theta_sq / 720.;
    const double k = (theta_sq > 0.0) ? 1. / theta_sq * (1. - p) : 1. / 12.;
    const M3 H = hat(r.w), H2 = hat_2(r.w);
    double e[3];
    const double tv[3] = {t.x, t.y, t.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double m = ((i == c) ? 1.0 : 0.0) - 0.5 * H.m[i][c] + k * H2.m[i][c];
            acc += m * tv[c];
        }
        e[i] = acc;
    }
    return V3{e[0], e[1], e[2]};
}

// se3::right_jacobian_q_matrix, math.rs:135-170.  E = so3::right_jacobian(w).
OPTIK_DEV M3 se3_q_matrix(const RotTerms &r, const V3 v, const M3 &E) {
    // math.rs:139-141: theta = w.norm(); theta_2 = theta^2 (re-squared)
    const double theta = r.theta;
    const double theta_2 = theta * theta;
    const double theta_4 = theta_2 * theta_2;
    double a, b;
    if (theta_2 > EPSILON) {
        const double s_t = r.s / theta;
        const double inv_1mc = 1. / (2. * (1. - r.c));
        a = 1. / theta_2 - s_t * inv_1mc;
        b = -2. / theta_4 + (1. + s_t) * inv_1mc / theta_2;
    } else {
        a = 1. / 12. + theta_2 / 720.;
        b = 1. / 360.;
    }
    const double wv[3] = {r.w.x, r.w.y, r.w.z};
    const double vv[3] = {v.x, v.y, v.z};
    const double d = wv[0] * vv[0] + wv[1] * vv[1] + wv[2] * vv[2];
    double cv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) cv[i] = b * d * wv[i] - (theta_2 * b + 2. * a) * vv[i];
    const M3 Hv = hat(v);
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            C.m[i][c] = 0.5 * Hv.m[i][c] + cv[i] * wv[c] + a * wv[i] * vv[c]
                        + ((i == c) ? d * a : 0.0);
    M3 Q;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc += C.m[i][k] * E.m[k][c];
            Q.m[i][c] = acc;
        }
    return Q;
}

// apply_weighting on one 3-block (objective.rs:13-23 / 25-35): R' diag(w) R e.
OPTIK_DEV V3 weight_block(const Q4 tq, const V3 e, const double *w3) {
    const V3 ew = qrot(tq, e);
    const V3 s{ew.x * w3[0], ew.y * w3[1], ew.z * w3[2]};
    return qrot(qconj(tq), s);
}

}  // namespace optik
