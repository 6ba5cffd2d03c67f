/*
 * optik_oracle.c -- CPU oracle (test infrastructure, NOT product code).
 *
 * Plain-C f64 restatement of the random-restart IK hot path of kylc/optik:
 *   math.rs, kinematics.rs (FK / Jacobian half), objective.rs, lib.rs:241-415,
 * plus the un-vendored third-party arithmetic that path depends on:
 *   NLopt SLSQP (Kraft 1988/1994; nlopt 0.8.1 via kylc/rust-nlopt@8e731e3),
 *   rand_core 0.9.3 seed_from_u64, rand_chacha 0.9.0 ChaCha8, rand 0.9.2 uniform f64.
 * See optik_oracle.h for the parity status.  Citations are relative to
 * /root/reference/.  Build with -ffp-contract=off: rustc never contracts a*b+c.
 */
#include "optik_oracle.h"

#include <math.h>
#ifndef sincos
void sincos(double x, double *s, double *c); /* glibc (GNU extension; not declared under -std=c11) */
#endif
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================= */
/* nalgebra 0.34 primitives (Isometry3 / UnitQuaternion), as the reference  */
/* uses them.  Quaternion storage [i,j,k,w].                                */
/* ======================================================================= */

static void v3_cross(const double a[3], const double b[3], double o[3]) {
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}

/* nalgebra Quaternion * Quaternion. */
static void q_mul(const double a[4], const double b[4], double o[4]) {
    double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    double i = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    double j = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    double k = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[0] = i; o[1] = j; o[2] = k; o[3] = w;
}

static void q_conj(const double a[4], double o[4]) {
    o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3];
}

/* nalgebra UnitQuaternion * Vector3:  t = 2 (v x r);  r' = w t + v x t + r. */
static void q_rot(const double q[4], const double r[3], double o[3]) {
    double t[3], c[3];
    v3_cross(q, r, t);
    t[0] *= 2.0; t[1] *= 2.0; t[2] *= 2.0;
    v3_cross(q, t, c);
    double x = t[0] * q[3] + c[0] + r[0];
    double y = t[1] * q[3] + c[1] + r[1];
    double z = t[2] * q[3] + c[2] + r[2];
    o[0] = x; o[1] = y; o[2] = z;
}

static void q_inv_rot(const double q[4], const double r[3], double o[3]) {
    double qc[4];
    q_conj(q, qc);
    q_rot(qc, r, o);
}

/* Isometry3 * Isometry3 = (t1 + q1*t2, q1 q2). */
static void pose_mul(const ok_pose *a, const ok_pose *b, ok_pose *o) {
    double s[3], q[4];
    q_rot(a->q, b->t, s);
    q_mul(a->q, b->q, q);
    o->t[0] = a->t[0] + s[0]; o->t[1] = a->t[1] + s[1]; o->t[2] = a->t[2] + s[2];
    memcpy(o->q, q, sizeof q);
}

/* Isometry3::inv_mul(a, b) = (q_a^-1 (t_b - t_a), q_a^-1 q_b)  (objective.rs:49,70). */
static void pose_inv_mul(const ok_pose *a, const ok_pose *b, ok_pose *o) {
    double qc[4], d[3];
    q_conj(a->q, qc);
    d[0] = b->t[0] - a->t[0]; d[1] = b->t[1] - a->t[1]; d[2] = b->t[2] - a->t[2];
    q_rot(qc, d, o->t);
    q_mul(qc, b->q, o->q);
}

static void pose_identity(ok_pose *p) {
    p->t[0] = p->t[1] = p->t[2] = 0.0;
    p->q[0] = p->q[1] = p->q[2] = 0.0; p->q[3] = 1.0;
}

/* ======================================================================= */
/* Elementary functions.  The reference calls f64::sin_cos / atan2 / sqrt   */
/* (platform libm).  The oracle and the HIP kernel both use the SAME        */
/* operation sequence below, built from IEEE + - * / only (correctly        */
/* rounded on x86-64 and on gfx950), so that GPU results can be compared    */
/* BIT-FOR-BIT with the oracle.  The algorithms restate Sun fdlibm /        */
/* FreeBSD msun (k_sin.c, k_cos.c, e_rem_pio2.c medium path, s_atan.c) --   */
/* the libm Rust itself ships for targets without a system libm; each is    */
/* accurate to < 1 ulp (tests/test_oracle_math.py checks against mpmath).   */
/* ======================================================================= */

static double k_sin(double x, double y) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double w = z * z;
    double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

static double k_cos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double w = z * z;
    double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

/* sin and cos of x, |x| < ~1e6 (joint angles).  Cody-Waite reduction with
 * the three-part pi/2 of e_rem_pio2.c, second iteration always taken. */
void ok_sincos(double x, double *s, double *c) {
#ifdef OK_PLATFORM_LIBM
    /* The libm-sensitivity study only (tests/test_oracle_libm_sensitivity.py): what the
     * reference itself calls on Linux -- f64::sin_cos binds glibc's sincos (math.rs:76,113,144; nalgebra's
     * from_axis_angle under kinematics.rs:245-248).  The GPU kernels cannot reproduce glibc's bits; the default
     * build below is the sequence they share with this file. */
    sincos(x, s, c);
    return;
#endif
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
                 pio2_2t = 2.02226624879595063154e-21;
    const double big = 6755399441055744.0; /* 1.5 * 2^52 */
    double fn = x * invpio2 + big;
    fn = fn - big;
    int n = (int)fn;
    double r = x - fn * pio2_1;
    double w = fn * pio2_1t;
    (void)w;
    double t = r;
    double w2 = fn * pio2_2;
    r = t - w2;
    w = fn * pio2_2t - ((t - r) - w2);
    double y0 = r - w;
    double y1 = (r - y0) - w;
    double ks = k_sin(y0, y1), kc = k_cos(y0, y1);
    switch (n & 3) {
    case 0: *s = ks; *c = kc; break;
    case 1: *s = kc; *c = -ks; break;
    case 2: *s = -ks; *c = -kc; break;
    default: *s = -kc; *c = ks; break;
    }
}

/* atan2(y, x) for y > 0, x >= 0 (the only use: math.rs:54 after the w >= 0
 * flip).  s_atan.c argument reduction + polynomial on t = y / x. */
double ok_atan2_q1(double y, double x) {
#ifdef OK_PLATFORM_LIBM
    return atan2(y, x); /* f64::atan2 (math.rs:54): glibc's, for the libm-sensitivity study only */
#endif
    static const double aT[11] = {
        3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01,
        -1.11111104054623557880e-01, 9.09088713343650656196e-02, -7.69187620504482999495e-02,
        6.66107313738753120669e-02, -5.83357013379057348645e-02, 4.97687799461593236017e-02,
        -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    double t = y / x;
    double num, den, hi, lo;
    int direct = 0;
    if (t < 0.4375) { num = t; den = 1.0; hi = 0.0; lo = 0.0; direct = 1; }
    else if (t < 0.6875) { num = 2.0 * t - 1.0; den = 2.0 + t;
        hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
    else if (t < 1.1875) { num = t - 1.0; den = t + 1.0;
        hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
    else if (t < 2.4375) { num = t - 1.5; den = 1.0 + 1.5 * t;
        hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
    else { num = -1.0; den = t;
        hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
    double u = num / den;
    double z = u * u;
    double w = z * z;
    double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (direct) return u - u * (s1 + s2);
    return hi - ((u * (s1 + s2) - lo) - u);
}

/* ======================================================================= */
/* math.rs                                                                  */
/* ======================================================================= */

static const double OK_EPSILON = 1e-6; /* math.rs:7 */

/* so3::hat, math.rs:13 (row-major 3x3 here). */
static void so3_hat(const double w[3], double M[3][3]) {
    M[0][0] = 0.0;   M[0][1] = -w[2]; M[0][2] = w[1];
    M[1][0] = w[2];  M[1][1] = 0.0;   M[1][2] = -w[0];
    M[2][0] = -w[1]; M[2][1] = w[0];  M[2][2] = 0.0;
}

/* so3::hat_2, math.rs:19-31. */
static void so3_hat_2(const double w[3], double M[3][3]) {
    double w11 = w[0] * w[0], w12 = w[0] * w[1], w13 = w[0] * w[2];
    double w22 = w[1] * w[1], w23 = w[1] * w[2], w33 = w[2] * w[2];
    M[0][0] = -w22 - w33; M[0][1] = w12;        M[0][2] = w13;
    M[1][0] = w12;        M[1][1] = -w11 - w33; M[1][2] = w23;
    M[2][0] = w13;        M[2][1] = w23;        M[2][2] = -w11 - w22;
}

/* so3::log, math.rs:40-63. */
void ok_so3_log(const double q[4], double out[3]) {
    double w, v[3];
    if (q[3] >= 0.0) { w = q[3]; v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; }
    else { w = -q[3]; v[0] = -q[0]; v[1] = -q[1]; v[2] = -q[2]; }
    double v_norm_2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double theta_over_v_norm;
    if (v_norm_2 > OK_EPSILON) {
        double v_norm = sqrt(v_norm_2);
        theta_over_v_norm = ok_atan2_q1(v_norm, w) / v_norm;
    } else {
        /* math.rs:57-59 */
        theta_over_v_norm = 1. / w - 1. / (3. * (w * w * w)) * v_norm_2
                            + 1. / (5. * (w * w * w * w * w)) * (v_norm_2 * v_norm_2);
    }
    out[0] = 2.0 * v[0] * theta_over_v_norm;
    out[1] = 2.0 * v[1] * theta_over_v_norm;
    out[2] = 2.0 * v[2] * theta_over_v_norm;
}

/* so3::right_jacobian, math.rs:72-94.  Row-major internal helper.
 * Deviation (quirk Q3): at theta_2 == 0 exactly the reference computes
 * c = (1-a)/0 = NaN; the oracle (and the kernel) use the limit c = 1/6. */
static void so3_right_jacobian_rm(const double w[3], double J[3][3]) {
    double theta_2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double theta_4 = theta_2 * theta_2;
    double theta = sqrt(theta_2);
    double s, c;
    ok_sincos(theta, &s, &c);
    double a, b;
    if (theta_2 > OK_EPSILON) a = s / theta;
    else a = 1. - 1. / 6. * theta_2 + 1. / 120.0 * theta_4;
    if (theta_2 > OK_EPSILON) b = (1. - c) / theta_2;
    else b = 1. / 2. - 1. / 24. * theta_2 + 1. / 720. * theta_4;
    double cc = (theta_2 > 0.0) ? (1. - a) / theta_2 : 1. / 6.;
    double e = (b - 2. * cc) / (2. * a);
    double H[3][3], H2[3][3];
    so3_hat(w, H);
    so3_hat_2(w, H2);
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k)
            J[r][k] = ((r == k) ? 1.0 : 0.0) + 0.5 * H[r][k] + e * H2[r][k];
}

void ok_so3_right_jacobian(const double w[3], double J[9]) {
    double M[3][3];
    so3_right_jacobian_rm(w, M);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) J[c * 3 + r] = M[r][c];
}

/* se3::log, math.rs:107-124.  Deviation (Q3): theta_sq == 0 uses the limit
 * (1-p)/theta_sq -> 1/12 instead of the reference's 0 * inf = NaN. */
void ok_se3_log(const ok_pose *X, double e[6]) {
    double w[3];
    ok_so3_log(X->q, w);
    double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double theta = sqrt(theta_sq);
    double p;
    if (theta > OK_EPSILON) {
        double s, c;
        ok_sincos(theta, &s, &c);
        p = 0.5 * (theta * s) / (1. - c);
    } else {
        p = 1. - theta_sq / 12. - theta_sq * theta_sq / 720.;
    }
    double k = (theta_sq > 0.0) ? 1. / theta_sq * (1. - p) : 1. / 12.;
    double H[3][3], H2[3][3];
    so3_hat(w, H);
    so3_hat_2(w, H2);
    for (int r = 0; r < 3; ++r) {
        double acc = 0.0;
        for (int c = 0; c < 3; ++c) {
            double m = ((r == c) ? 1.0 : 0.0) - 0.5 * H[r][c] + k * H2[r][c];
            acc += m * X->t[c];
        }
        e[r] = acc;
    }
    e[3] = w[0]; e[4] = w[1]; e[5] = w[2];
}

/* se3::right_jacobian_q_matrix, math.rs:135-170 (row-major out). */
static void se3_q_matrix(const double v[3], const double w[3], double Q[3][3]) {
    double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double theta_2 = theta * theta;
    double theta_4 = theta_2 * theta_2;
    double a, b;
    if (theta_2 > OK_EPSILON) {
        double s, c;
        ok_sincos(theta, &s, &c);
        double s_t = s / theta;
        double inv_1mc = 1. / (2. * (1. - c));
        a = 1. / theta_2 - s_t * inv_1mc;
        b = -2. / theta_4 + (1. + s_t) * inv_1mc / theta_2;
    } else {
        a = 1. / 12. + theta_2 / 720.;
        b = 1. / 360.;
    }
    double d = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
    double cv[3];
    for (int i = 0; i < 3; ++i) cv[i] = b * d * w[i] - (theta_2 * b + 2. * a) * v[i];
    double Hv[3][3], C[3][3], E[3][3];
    so3_hat(v, Hv);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            C[r][c] = 0.5 * Hv[r][c] + cv[r] * w[c] + a * w[r] * v[c] + ((r == c) ? d * a : 0.0);
    so3_right_jacobian_rm(w, E);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += C[r][k] * E[k][c];
            Q[r][c] = acc;
        }
}

/* se3::right_jacobian, math.rs:191-203 (row-major 6x6 internal). */
static void se3_right_jacobian_rm(const ok_pose *X, double U[6][6]) {
    double w[3], J[3][3], Q[3][3];
    ok_so3_log(X->q, w);
    so3_right_jacobian_rm(w, J);
    se3_q_matrix(X->t, w, Q);
    memset(U, 0, 36 * sizeof(double));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            U[r][c] = J[r][c];
            U[r][c + 3] = Q[r][c];
            U[r + 3][c + 3] = J[r][c];
        }
}

void ok_se3_right_jacobian(const ok_pose *X, double J[36]) {
    double U[6][6];
    se3_right_jacobian_rm(X, U);
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) J[c * 6 + r] = U[r][c];
}

/* ======================================================================= */
/* kinematics.rs                                                            */
/* ======================================================================= */

/* urdf_to_tfm, kinematics.rs:263-267; nalgebra UnitQuaternion::from_euler_angles. */
void ok_pose_from_rpy(const double xyz[3], const double rpy[3], ok_pose *out) {
    double sr, cr, sp, cp, sy, cy;
    ok_sincos(rpy[0] * 0.5, &sr, &cr);
    ok_sincos(rpy[1] * 0.5, &sp, &cp);
    ok_sincos(rpy[2] * 0.5, &sy, &cy);
    out->q[3] = cr * cp * cy + sr * sp * sy;
    out->q[0] = sr * cp * cy - cr * sp * sy;
    out->q[1] = cr * sp * cy + sr * cp * sy;
    out->q[2] = cr * cp * sy - sr * sp * cy;
    out->t[0] = xyz[0]; out->t[1] = xyz[1]; out->t[2] = xyz[2];
}

/* ---- target matrix -> pose, the two readings of the reference's bindings ------------------------------
 * nalgebra 0.34.0 (Cargo.lock:579-581; not vendored: restated from the crate's published source, PARITY
 * UNPINNED).  m: 3x3 rotation block, column-major (m[3 c + r]); q out as [i, j, k, w]. */

/* UnitQuaternion::from_rotation_matrix: what try_convert::<Matrix4, Isometry3> ends in
 * (optik-py/src/lib.rs:8-15). */
static void quat_from_rot(const double *m, double q[4]) {
#define RM(r, c) m[3 * (c) + (r)]
    const double tr = RM(0, 0) + RM(1, 1) + RM(2, 2);
    double den;
    if (tr > 0.0) {
        den = sqrt(tr + 1.0) * 2.0;
        q[3] = 0.25 * den;
        q[0] = (RM(2, 1) - RM(1, 2)) / den; q[1] = (RM(0, 2) - RM(2, 0)) / den; q[2] = (RM(1, 0) - RM(0, 1)) / den;
    } else if (RM(0, 0) > RM(1, 1) && RM(0, 0) > RM(2, 2)) {
        den = sqrt(1.0 + RM(0, 0) - RM(1, 1) - RM(2, 2)) * 2.0;
        q[3] = (RM(2, 1) - RM(1, 2)) / den;
        q[0] = 0.25 * den; q[1] = (RM(0, 1) + RM(1, 0)) / den; q[2] = (RM(0, 2) + RM(2, 0)) / den;
    } else if (RM(1, 1) > RM(2, 2)) {
        den = sqrt(1.0 + RM(1, 1) - RM(0, 0) - RM(2, 2)) * 2.0;
        q[3] = (RM(0, 2) - RM(2, 0)) / den;
        q[0] = (RM(0, 1) + RM(1, 0)) / den; q[1] = 0.25 * den; q[2] = (RM(1, 2) + RM(2, 1)) / den;
    } else {
        den = sqrt(1.0 + RM(2, 2) - RM(0, 0) - RM(1, 1)) * 2.0;
        q[3] = (RM(1, 0) - RM(0, 1)) / den;
        q[0] = (RM(0, 2) + RM(2, 0)) / den; q[1] = (RM(1, 2) + RM(2, 1)) / den; q[2] = 0.25 * den;
    }
#undef RM
}

/* Rotation3::from_axis_angle, column-major out. */
static void rot_axis_angle(const double u[3], double angle, double *o) {
    if (angle == 0.0) {
        for (int e = 0; e < 9; ++e) o[e] = (e % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double sx = u[0] * u[0], sy = u[1] * u[1], sz = u[2] * u[2];
    double sn, cs; /* f64::sin_cos: platform libm -- ONE sincos call on both sides (glibc's sin + cos differ from its
                    * sincos in the last bit for ~0.5 % of arguments, and compilers merge the pair or not as they like) */
    sincos(angle, &sn, &cs);
    const double omc = 1.0 - cs;
    o[0] = sx + (1.0 - sx) * cs;            o[3] = u[0] * u[1] * omc - u[2] * sn; o[6] = u[0] * u[2] * omc + u[1] * sn;
    o[1] = u[0] * u[1] * omc + u[2] * sn;   o[4] = sy + (1.0 - sy) * cs;          o[7] = u[1] * u[2] * omc - u[0] * sn;
    o[2] = u[0] * u[2] * omc - u[1] * sn;   o[5] = u[1] * u[2] * omc + u[0] * sn; o[8] = sz + (1.0 - sz) * cs;
}

/* o = a * b (nalgebra gemm for small static matrices: column by column, axpy over k in order). */
static void m3_mul(const double *a, const double *b, double *o) {
    double t[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            double y = a[r] * b[3 * c];
            y = a[3 + r] * b[3 * c + 1] + y;
            y = a[6 + r] * b[3 * c + 2] + y;
            t[3 * c + r] = y;
        }
    memcpy(o, t, sizeof t);
}

static double m3_diff_norm2(const double *a, const double *b) {
    double res = 0.0;
    for (int c = 0; c < 3; ++c) {
        const double d0 = a[3 * c] - b[3 * c], d1 = a[3 * c + 1] - b[3 * c + 1], d2 = a[3 * c + 2] - b[3 * c + 2];
        res += d0 * d0 + d1 * d1 + d2 * d2;
    }
    return res;
}

/* iterative != 0: UnitQuaternion::from_matrix = Rotation3::from_matrix_eps(m, EPSILON, 0, identity) then
 * from_rotation_matrix -- the C path, optik-cpp/src/lib.rs:141-142; iterative == 0: from_rotation_matrix alone. */
void ok_quat_from_matrix(const double m[9], int iterative, double q[4]) {
    if (!iterative) { quat_from_rot(m, q); return; }
    const double eps = 2.220446049250313e-16;
    const double sq_eps = sqrt(eps), eps2 = eps * eps;
    const double disturb = sq_eps > eps2 ? sq_eps : eps2;
    double ax[3] = {1.0, 0.0, 0.0};
    double rot[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (long it = 0; it < 100000; ++it) { /* (nalgebra: usize::MAX) */
        double axis[3], denom;
        for (int c = 0; c < 3; ++c) {
            const double *a = rot + 3 * c, *b = m + 3 * c;
            double cr[3];
            v3_cross(a, b, cr);
            const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
            if (c == 0) { axis[0] = cr[0]; axis[1] = cr[1]; axis[2] = cr[2]; denom = d; }
            else { axis[0] += cr[0]; axis[1] += cr[1]; axis[2] += cr[2]; denom += d; }
        }
        const double dv = fabs(denom) + eps;
        const double aa[3] = {axis[0] / dv, axis[1] / dv, axis[2] / dv};
        const double sq = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
        if (sq > eps * eps) { /* Unit::try_new_and_get */
            const double nrm = sqrt(sq);
            const double u[3] = {aa[0] / nrm, aa[1] / nrm, aa[2] / nrm};
            double rd[9];
            rot_axis_angle(u, nrm, rd);
            m3_mul(rd, rot, rot);
        } else {
            double pert[9];
            memcpy(pert, rot, sizeof pert);
            const double n0 = m3_diff_norm2(m, rot);
            double n1;
            for (;;) {
                double rp[9];
                rot_axis_angle(ax, disturb, rp);
                m3_mul(pert, rp, pert);
                n1 = m3_diff_norm2(m, pert);
                if (!(fabs(n0 - n1) <= eps)) break; /* abs_diff_ne!: true for NaN too */
            }
            if (n0 < n1) break;
            const double t = ax[0]; ax[0] = ax[1]; ax[1] = ax[2]; ax[2] = t; /* yzx */
            memcpy(rot, pert, sizeof pert);
        }
    }
    quat_from_rot(rot, q);
}

/* JointType::local_transform, kinematics.rs:243-255. */
static void local_transform(const ok_chain *c, int j, double qj, ok_pose *o) {
    pose_identity(o);
    if (c->type[j] == OK_JOINT_REVOLUTE) {
        /* UnitQuaternion::from_axis_angle: (sin, cos)(angle/2) */
        double s, co;
        ok_sincos(qj / 2.0, &s, &co);
        o->q[0] = c->axis[j][0] * s; o->q[1] = c->axis[j][1] * s; o->q[2] = c->axis[j][2] * s;
        o->q[3] = co;
    } else if (c->type[j] == OK_JOINT_PRISMATIC) {
        o->t[0] = c->axis[j][0] * qj; o->t[1] = c->axis[j][1] * qj; o->t[2] = c->axis[j][2] * qj;
    }
}

/* forward_kinematics_mut, kinematics.rs:123-164. */
void ok_fk(const ok_chain *c, const double *q, const ok_pose *ee_offset, ok_pose *joint_tfms,
           ok_pose *ee_tfm) {
    ok_pose state, local, jt, next;
    pose_identity(&state);
    int qidx = 0;
    for (int j = 0; j < c->n_joints; ++j) {
        int nq = (c->type[j] == OK_JOINT_FIXED) ? 0 : 1;
        local_transform(c, j, nq ? q[qidx] : 0.0, &local);
        pose_mul(&c->origin[j], &local, &jt);   /* joint.origin * local_transform(q) */
        pose_mul(&state, &jt, &next);           /* state.tfm *= ... */
        state = next;
        qidx += nq;
        joint_tfms[j] = state;
    }
    pose_mul(&joint_tfms[c->n_joints - 1], ee_offset, ee_tfm); /* :163 */
}

/* joint_jacobian, kinematics.rs:166-196.  6 x n column-major, rows 0-2 linear. */
void ok_joint_jacobian(const ok_chain *c, const ok_pose *joint_tfms, const ok_pose *ee,
                       double *J) {
    int col = 0;
    for (int j = 0; j < c->n_joints; ++j) {
        if (c->type[j] == OK_JOINT_REVOLUTE) {
            double angular[3], d[3], linear[3], al[3], ll[3];
            q_rot(joint_tfms[j].q, c->axis[j], angular);
            d[0] = ee->t[0] - joint_tfms[j].t[0];
            d[1] = ee->t[1] - joint_tfms[j].t[1];
            d[2] = ee->t[2] - joint_tfms[j].t[2];
            v3_cross(angular, d, linear);
            q_inv_rot(ee->q, angular, al);
            q_inv_rot(ee->q, linear, ll);
            J[col * 6 + 0] = ll[0]; J[col * 6 + 1] = ll[1]; J[col * 6 + 2] = ll[2];
            J[col * 6 + 3] = al[0]; J[col * 6 + 4] = al[1]; J[col * 6 + 5] = al[2];
            col += 1;
        } else if (c->type[j] == OK_JOINT_PRISMATIC) {
            /* kinematics.rs:185 is todo!(): the reference panics.  The oracle
             * writes NaN so any use is visible. */
            for (int r = 0; r < 6; ++r) J[col * 6 + r] = NAN;
            col += 1;
        }
    }
}

/* ======================================================================= */
/* objective.rs                                                             */
/* ======================================================================= */

/* approx::relative_eq!(a, b, epsilon = eps) with default max_relative = f64::EPSILON. */
static int relative_eq(double a, double b, double eps) {
    if (a == b) return 1;
    if (isinf(a) || isinf(b)) return 0;
    double d = fabs(a - b);
    if (d <= eps) return 1;
    double la = fabs(a), lb = fabs(b);
    double largest = (lb > la) ? lb : la;
    return d <= largest * 2.220446049250313e-16;
}

/* nalgebra Matrix::is_identity on a 3x1 vector: element 0 ~ 1, the others ~ 0
 * (objective.rs:13,25). */
static int vec3_is_identity(const double w[3], double eps) {
    return relative_eq(w[0], 1.0, eps) && relative_eq(w[1], 0.0, eps) && relative_eq(w[2], 0.0, eps);
}

/* apply_weighting, objective.rs:7-38. */
static void apply_weighting(double e[6], const ok_pose *target, const double wl[3],
                            const double wa[3]) {
    const double IDENTITY_EPS = 1e-20;
    if (!vec3_is_identity(wl, IDENTITY_EPS)) {
        double w[3], s[3];
        q_rot(target->q, e, w);
        s[0] = w[0] * wl[0]; s[1] = w[1] * wl[1]; s[2] = w[2] * wl[2];
        q_inv_rot(target->q, s, e);
    }
    if (!vec3_is_identity(wa, IDENTITY_EPS)) {
        double w[3], s[3];
        q_rot(target->q, e + 3, w);
        s[0] = w[0] * wa[0]; s[1] = w[1] * wa[1]; s[2] = w[2] * wa[2];
        q_inv_rot(target->q, s, e + 3);
    }
}

/* objective, objective.rs:40-57. */
double ok_objective(const ok_chain *c, const ok_pose *target, const ok_pose *joint_tfms,
                    const ok_pose *ee, const double wl[3], const double wa[3]) {
    (void)c; (void)joint_tfms;
    ok_pose X;
    double e[6];
    pose_inv_mul(target, ee, &X);
    ok_se3_log(&X, e);
    apply_weighting(e, target, wl, wa);
    double acc = 0.0;
    for (int i = 0; i < 6; ++i) acc += e[i] * e[i];
    return acc;
}

/* objective_grad, objective.rs:60-110. */
void ok_objective_grad(const ok_chain *c, const ok_pose *target, const ok_pose *joint_tfms,
                       const ok_pose *ee, const double wl[3], const double wa[3], double *g) {
    ok_pose X;
    double Jq[6 * OK_MAX_DOF], Jlog[6][6], Jtask[6][OK_MAX_DOF], e[6];
    int n = c->n_pos;
    pose_inv_mul(target, ee, &X);
    ok_joint_jacobian(c, joint_tfms, ee, Jq);
    se3_right_jacobian_rm(&X, Jlog);
    for (int r = 0; r < 6; ++r)
        for (int k = 0; k < n; ++k) {
            double acc = 0.0;
            for (int m = 0; m < 6; ++m) acc += Jlog[r][m] * Jq[k * 6 + m];
            Jtask[r][k] = acc;
        }
    ok_se3_log(&X, e);
    double wl2[3] = {wl[0] * wl[0], wl[1] * wl[1], wl[2] * wl[2]};
    double wa2[3] = {wa[0] * wa[0], wa[1] * wa[1], wa[2] * wa[2]};
    apply_weighting(e, target, wl2, wa2);
    for (int k = 0; k < n; ++k) {
        double acc = 0.0;
        for (int r = 0; r < 6; ++r) acc += (2.0 * e[r]) * Jtask[r][k];
        g[k] = acc;
    }
}

/* The NLopt callback body, lib.rs:305-337 (without the early-exit test). */
double ok_eval(const ok_chain *c, const ok_pose *target, const ok_pose *ee_offset,
               const double wl[3], const double wa[3], const double *q, double *g) {
    ok_pose jt[OK_MAX_JOINTS], ee;
    ok_fk(c, q, ee_offset, jt, &ee);
    if (g) ok_objective_grad(c, target, jt, &ee, wl, wa, g);
    return ok_objective(c, target, jt, &ee, wl, wa);
}

/* ======================================================================= */
/* RNG: rand_core::SeedableRng::seed_from_u64 (PCG32), rand_chacha ChaCha8, */
/* rand::distr::Uniform<f64> inclusive.  [EXT] -- SURVEY appendix C.        */
/* ======================================================================= */

static uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

#define OK_QR(a, b, c, d)            \
    a += b; d ^= a; d = rotl32(d, 16); \
    c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  \
    c += d; b ^= c; b = rotl32(b, 7);

/* DJB ChaCha block: words 12-13 = 64-bit block counter, 14-15 = 64-bit stream. */
void ok_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds,
                     uint32_t out[16]) {
    uint32_t s[16], x[16];
    s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
    for (int i = 0; i < 8; ++i) s[4 + i] = key[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32);
    s[14] = (uint32_t)stream;  s[15] = (uint32_t)(stream >> 32);
    memcpy(x, s, sizeof x);
    for (int r = 0; r < rounds; r += 2) {
        OK_QR(x[0], x[4], x[8], x[12]) OK_QR(x[1], x[5], x[9], x[13])
        OK_QR(x[2], x[6], x[10], x[14]) OK_QR(x[3], x[7], x[11], x[15])
        OK_QR(x[0], x[5], x[10], x[15]) OK_QR(x[1], x[6], x[11], x[12])
        OK_QR(x[2], x[7], x[8], x[13]) OK_QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

/* rand_core 0.9 SeedableRng::seed_from_u64: PCG32 output, little-endian words. */
void ok_seed_from_u64(uint64_t state, uint32_t key[8]) {
    const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
    for (int i = 0; i < 8; ++i) {
        state = state * MUL + INC;
        uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
}

/* rand 0.9.2 `Rng::random_range(lb..=ub)` for f64 (lib.rs:89).  Call chain in the
 * crate (src/rng.rs, src/distr/uniform.rs, src/distr/uniform_float.rs):
 *   Rng::random_range(range)            -> range.sample_single(self).unwrap()
 *   SampleRange for RangeInclusive<f64> -> UniformFloat::<f64>::sample_single_inclusive(lo, hi, rng)
 *   sample_single_inclusive             -> scale = high - low;  (no division, no decrease loop)
 *                                          value1_2 = (next_u64 >> 12) with exponent 0;
 *                                          value0_1 = value1_2 - 1.0;  value0_1 * scale + low
 * Rule 0 (OK_RANGE_SINGLE_INCLUSIVE, the default) is that chain.  Rule 1
 * (OK_RANGE_NEW_INCLUSIVE) is `Uniform::new_inclusive(lo, hi).sample(rng)` -- what rand
 * 0.8.5's default `sample_single_inclusive` did: scale = (high - low) / (1 - eps) with the
 * 1-ulp decrease loop -- kept so that either reading can be checked against a real
 * `cargo run`.  The crate is not vendored: see the header. */
static int g_range_rule = OK_RANGE_SINGLE_INCLUSIVE;
void ok_set_range_rule(int rule) { g_range_rule = rule; }
int ok_get_range_rule(void) { return g_range_rule; }

double ok_uniform_scale(double low, double high, int rule) {
    if (rule == OK_RANGE_SINGLE_INCLUSIVE) return high - low;
    const double max_rand = 1.0 - 2.220446049250313e-16;
    double scale = (high - low) / max_rand;
    while (scale * max_rand + low > high) {
        uint64_t u;
        memcpy(&u, &scale, 8);
        u -= 1;
        memcpy(&scale, &u, 8);
    }
    return scale;
}

double ok_uniform_inclusive(double low, double high, uint64_t bits) {
    const double scale = ok_uniform_scale(low, high, g_range_rule);
    uint64_t m = (bits >> 12) | 0x3ff0000000000000ull; /* into_float_with_exponent(0) */
    double value1_2;
    memcpy(&value1_2, &m, 8);
    double value0_1 = value1_2 - 1.0;
    return value0_1 * scale + low;
}

/* lib.rs:358-370 + random_configuration lib.rs:86-91:
 * ChaCha8Rng::seed_from_u64(42); set_stream(i); one next_u64 per joint. */
void ok_restart_seed(const ok_chain *c, uint64_t restart_index, double *q0) {
    uint32_t key[8], blk[16];
    ok_seed_from_u64(42, key);
    int n = c->n_pos;
    uint64_t block = 0;
    int word = 16;
    for (int k = 0; k < n; ++k) {
        if (word >= 16) { ok_chacha_block(key, block++, restart_index, 8, blk); word = 0; }
        uint64_t lo = blk[word], hi = blk[word + 1];
        word += 2;
        q0[k] = ok_uniform_inclusive(c->lb[k], c->ub[k], lo | (hi << 32));
    }
}

/* ======================================================================= */
/* NLopt SLSQP (Kraft), box bounds only (m = meq = 0).  [EXT]               */
/* Restated from the published algorithm: D. Kraft, "A software package for */
/* sequential quadratic programming", DFVLR-FB 88-28 (1988); Lawson &       */
/* Hanson, "Solving Least Squares Problems" (1974) ch. 23 for H12 / NNLS /  */
/* LDP; Fletcher & Powell (1974) composite-t LDL' rank-one update.  NLopt's */
/* modifications (S. G. Johnson 2010) are called out where they apply.      */
/* ======================================================================= */

#define SQ_N OK_MAX_DOF

static const double EPMACH = 2.220446049250313e-16;

static double bl_dot(int n, const double *x, int incx, const double *y, int incy) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += x[i * incx] * y[i * incy];
    return s;
}

/* NLopt's dnrm2: scaled by the max magnitude. */
static double bl_nrm2(int n, const double *x, int incx) {
    double xmax = 0.0;
    for (int i = 0; i < n; ++i) { double a = fabs(x[i * incx]); if (a > xmax) xmax = a; }
    if (xmax == 0.0) return 0.0;
    double scale = 1.0 / xmax, sum = 0.0;
    for (int i = 0; i < n; ++i) { double xs = scale * x[i * incx]; sum += xs * xs; }
    return xmax * sqrt(sum);
}

/* Lawson-Hanson H12: construct (mode 1) / apply (mode 2) a Householder
 * transformation.  u is the pivot vector with stride iue; columns of c have
 * element stride ice and vector stride icv.  1-based lpivot, l1, m. */
static void h12(int mode, int lpivot, int l1, int m, double *u, int iue, double *up, double *c,
                int ice, int icv, int ncv) {
    if (0 >= lpivot || lpivot >= l1 || l1 > m) return;
#define U(j) u[((j) - 1) * iue]
    double cl = fabs(U(lpivot));
    if (mode != 2) {
        for (int j = l1; j <= m; ++j) { double sm = fabs(U(j)); if (sm > cl) cl = sm; }
        if (cl <= 0.0) return;
        double clinv = 1.0 / cl;
        double d = U(lpivot) * clinv;
        double sm = d * d;
        for (int j = l1; j <= m; ++j) { d = U(j) * clinv; sm += d * d; }
        cl *= sqrt(sm);
        if (U(lpivot) > 0.0) cl = -cl;
        *up = U(lpivot) - cl;
        U(lpivot) = cl;
    } else if (cl <= 0.0) {
        return;
    }
    if (ncv <= 0) return;
    double b = *up * U(lpivot);
    if (b >= 0.0) return;
    b = 1.0 / b;
    int i2 = 1 - icv + ice * (lpivot - 1);
    int incr = ice * (l1 - lpivot);
    for (int j = 1; j <= ncv; ++j) {
        i2 += icv;
        int i3 = i2 + incr, i4 = i3;
        double sm = c[i2 - 1] * *up;
        for (int i = l1; i <= m; ++i) { sm += c[i3 - 1] * U(i); i3 += ice; }
        if (sm == 0.0) continue;
        sm *= b;
        c[i2 - 1] += sm * *up;
        for (int i = l1; i <= m; ++i) { c[i4 - 1] += sm * U(i); i4 += ice; }
    }
#undef U
}

/* BLAS drotg as restated in NLopt's slsqp.c (dsrotg). */
static void rotg(double *da, double *db, double *c, double *s) {
    double roe = (fabs(*da) > fabs(*db)) ? *da : *db;
    double scale = fabs(*da) + fabs(*db);
    double r, z;
    if (scale == 0.0) {
        *c = 1.0; *s = 0.0; r = 0.0;
    } else {
        double a = *da / scale, b = *db / scale;
        r = scale * sqrt(a * a + b * b);
        if (roe < 0.0) r = -r;
        *c = *da / r;
        *s = *db / r;
    }
    z = *s;
    if (fabs(*c) > 0.0 && fabs(*c) <= *s) z = 1.0 / *c;
    *da = r;
    *db = z;
}

static void rot(int n, double *x, int incx, double *y, int incy, double c, double s) {
    for (int i = 0; i < n; ++i) {
        double xi = x[i * incx], yi = y[i * incy];
        x[i * incx] = c * xi + s * yi;
        y[i * incy] = c * yi - s * xi;
    }
}

/* Diagnostic: how the NNLS calls ended -- [iter][nsetp] counts, iter capped at 15 (ok_nnls_hist reads and clears them;
 * tools/nnls_pass_hist.py).  Not part of any result. */
static unsigned long long g_nnls_hist[16][16];
void ok_nnls_hist(unsigned long long *out256) {
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) out256[i * 16 + j] = __atomic_exchange_n(&g_nnls_hist[i][j], 0ull, __ATOMIC_RELAXED);
}

/* Lawson-Hanson NNLS:  min ||A x - b||  s.t. x >= 0.   A is m x n column-major,
 * leading dimension mda.  Returns mode: 1 ok, 2 bad dims, 3 iteration count. */
static int nnls(double *a, int mda, int m, int n, double *b, double *x, double *rnorm, double *w,
                double *z, int *indx) {
#define A(i, j) a[((j) - 1) * mda + ((i) - 1)]
    const double factor = 0.01;
    if (m <= 0 || n <= 0) return 2;
    int mode = 1, iter = 0, itmax = 3 * n;
    for (int i = 1; i <= n; ++i) indx[i - 1] = i;
    int iz1 = 1, iz2 = n, nsetp = 0, npp1 = 1;
    int izmax = 0, j, jj = 0;
    double up = 0.0;
    for (int i = 0; i < n; ++i) x[i] = 0.0;

    for (;;) { /* step two */
        if (iz1 > iz2 || nsetp >= m) break;
        for (int iz = iz1; iz <= iz2; ++iz) {
            j = indx[iz - 1];
            w[j - 1] = bl_dot(m - nsetp, &A(npp1, j), 1, &b[npp1 - 1], 1);
        }
        int found = 0;
        for (;;) { /* step three */
            double wmax = 0.0;
            for (int iz = iz1; iz <= iz2; ++iz) {
                j = indx[iz - 1];
                if (w[j - 1] <= wmax) continue;
                wmax = w[j - 1];
                izmax = iz;
            }
            if (wmax <= 0.0) break; /* step four: KKT satisfied */
            int iz = izmax;
            j = indx[iz - 1];
            /* step five */
            double asave = A(npp1, j);
            h12(1, npp1, npp1 + 1, m, &A(1, j), 1, &up, z, 1, 1, 0);
            double unorm = bl_nrm2(nsetp, &A(1, j), 1);
            double t = factor * fabs(A(npp1, j));
            double d1 = unorm + t;
            if (d1 - unorm > 0.0) {
                memcpy(z, b, (size_t)m * sizeof(double));
                h12(2, npp1, npp1 + 1, m, &A(1, j), 1, &up, z, 1, 1, 1);
                if (z[npp1 - 1] / A(npp1, j) > 0.0) { found = 1; }
            }
            if (found) {
                memcpy(b, z, (size_t)m * sizeof(double));
                indx[iz - 1] = indx[iz1 - 1];
                indx[iz1 - 1] = j;
                ++iz1;
                nsetp = npp1;
                ++npp1;
                for (int jz = iz1; jz <= iz2; ++jz) {
                    jj = indx[jz - 1];
                    h12(2, nsetp, npp1, m, &A(1, j), 1, &up, &A(1, jj), 1, mda, 1);
                }
                w[j - 1] = 0.0;
                for (int i = npp1; i <= m; ++i) A(i, j) = 0.0;
                break;
            }
            A(npp1, j) = asave;
            w[j - 1] = 0.0;
        }
        if (!found) break; /* wmax <= 0 -> done */

        /* step six: solve the triangular system; then steps seven..eleven */
        int resolve = 1;
        while (resolve) {
            for (int ip = nsetp; ip >= 1; --ip) {
                if (ip != nsetp) {
                    for (int i = 0; i < ip; ++i) z[i] -= z[ip] * A(i + 1, jj);
                }
                jj = indx[ip - 1];
                z[ip - 1] /= A(ip, jj);
            }
            ++iter;
            if (iter > itmax) { mode = 3; goto done; }
            double alpha = 1.0;
            jj = 0;
            for (int ip = 1; ip <= nsetp; ++ip) {
                if (z[ip - 1] > 0.0) continue;
                int l = indx[ip - 1];
                double t = -x[l - 1] / (z[ip - 1] - x[l - 1]);
                if (alpha < t) continue;
                alpha = t;
                jj = ip;
            }
            for (int ip = 1; ip <= nsetp; ++ip) {
                int l = indx[ip - 1];
                x[l - 1] = (1.0 - alpha) * x[l - 1] + alpha * z[ip - 1];
            }
            if (jj == 0) { resolve = 0; break; } /* back to step two */
            /* step eleven: move coefficient i from set P to set Z */
            int i = indx[jj - 1];
            for (;;) {
                x[i - 1] = 0.0;
                ++jj;
                for (j = jj; j <= nsetp; ++j) {
                    int ii = indx[j - 1];
                    indx[j - 2] = ii;
                    double c, s;
                    rotg(&A(j - 1, ii), &A(j, ii), &c, &s);
                    double t = A(j - 1, ii);
                    rot(n, &A(j - 1, 1), mda, &A(j, 1), mda, c, s);
                    A(j - 1, ii) = t;
                    A(j, ii) = 0.0;
                    rot(1, &b[j - 2], 1, &b[j - 1], 1, c, s);
                }
                npp1 = nsetp;
                --nsetp;
                --iz1;
                indx[iz1 - 1] = i;
                if (nsetp <= 0) { mode = 3; goto done; }
                int again = 0;
                for (jj = 1; jj <= nsetp; ++jj) {
                    i = indx[jj - 1];
                    if (x[i - 1] <= 0.0) { again = 1; break; }
                }
                if (!again) break;
            }
            memcpy(z, b, (size_t)m * sizeof(double));
        }
    }
done: {
        int k = (npp1 < m) ? npp1 : m;
        *rnorm = bl_nrm2(m - nsetp, &b[k - 1], 1);
        if (npp1 > m) for (int i = 0; i < n; ++i) w[i] = 0.0;
    }
    __atomic_fetch_add(&g_nnls_hist[iter < 15 ? iter : 15][nsetp < 15 ? nsetp : 15], 1ull, __ATOMIC_RELAXED);
    return mode;
#undef A
}

/* Lawson-Hanson LDP:  min ||x||  s.t.  G x >= h.   G is m x n column-major (ld mg).
 * w workspace >= (n+1)*(m+2) + 2m; on success w[0..m) holds the multipliers. */
static int ldp(const double *g, int mg, int m, int n, const double *h, double *x, double *xnorm,
               double *w, int *indx) {
    if (n <= 0) return 2;
    for (int i = 0; i < n; ++i) x[i] = 0.0;
    *xnorm = 0.0;
    if (m == 0) return 1;
    int iw = 0;
    for (int j = 0; j < m; ++j) {
        for (int i = 0; i < n; ++i) w[iw++] = g[i * mg + j];
        w[iw++] = h[j];
    }
    int if_ = iw;
    for (int i = 0; i < n; ++i) w[iw++] = 0.0;
    w[iw] = 1.0;
    int n1 = n + 1;
    int iz = iw + 1, iy = iz + n1, iwdual = iy + m;
    double rnorm;
    int mode = nnls(w, n1, n1, m, &w[if_], &w[iy], &rnorm, &w[iwdual], &w[iz], indx);
    if (mode != 1) return mode;
    if (rnorm <= 0.0) return 4;
    double fac = 1.0 - bl_dot(m, h, 1, &w[iy], 1);
    double d1 = 1.0 + fac;
    if (d1 - 1.0 <= 0.0) return 4;
    fac = 1.0 / fac;
    for (int j = 0; j < n; ++j) x[j] = fac * bl_dot(m, &g[j * mg], 1, &w[iy], 1);
    *xnorm = bl_nrm2(n, x, 1);
    for (int i = 0; i < m; ++i) w[i] = 0.0;
    for (int i = 0; i < m; ++i) w[i] += fac * w[iy + i];
    return 1;
}

/* Kraft LSI:  min ||E x - f||  s.t.  G x >= h.   E is me x n (ld le), G mg x n (ld lg). */
static int lsi(double *e, double *f, double *g, double *h, int le, int me, int lg, int mg, int n,
               double *x, double *xnorm, double *w, int *jw) {
#define E(i, j) e[((j) - 1) * le + ((i) - 1)]
#define G(i, j) g[((j) - 1) * lg + ((i) - 1)]
    double t;
    /* QR factors of E and application to f */
    for (int i = 1; i <= n; ++i) {
        int j = (i + 1 < n) ? i + 1 : n;
        h12(1, i, i + 1, me, &E(1, i), 1, &t, &E(1, j), 1, le, n - i);
        h12(2, i, i + 1, me, &E(1, i), 1, &t, f, 1, 1, 1);
    }
    /* transform G and h to get the least distance problem */
    for (int i = 1; i <= mg; ++i) {
        for (int j = 1; j <= n; ++j) {
            if (!(fabs(E(j, j)) >= EPMACH)) return 5;
            G(i, j) = (G(i, j) - bl_dot(j - 1, &G(i, 1), lg, &E(1, j), 1)) / E(j, j);
        }
        h[i - 1] -= bl_dot(n, &G(i, 1), lg, f, 1);
    }
    int mode = ldp(g, lg, mg, n, h, x, xnorm, w, jw);
    if (mode != 1) return mode;
    /* solution of the original problem */
    for (int i = 0; i < n; ++i) x[i] += f[i];
    for (int i = n; i >= 1; --i) {
        int j = (i + 1 < n) ? i + 1 : n;
        x[i - 1] = (x[i - 1] - bl_dot(n - i, &E(i, j), le, &x[j - 1], 1)) / E(i, i);
    }
    int j = (n + 1 < me) ? n + 1 : me;
    t = bl_nrm2(me - n, &f[j - 1], 1);
    *xnorm = sqrt(*xnorm * *xnorm + t * t);
    return 1;
#undef E
#undef G
}

/* Kraft LSQ specialised to m = meq = 0 with finite bounds:
 *   min ||E s - f||,  E = D^1/2 L',  f = -D^-1/2 L^-1 g,   xl <= s <= xu
 * via LSEI(mc = 0) -> LSI -> LDP -> NNLS.  l: packed LDL' (columnwise, D on
 * the diagonal slots).  Returns the LSQ mode (1 = success). */
static int lsq_box(int n, const double *l, const double *g, const double *xl, const double *xu,
                   double *s) {
    double E[SQ_N * SQ_N], f[SQ_N], G[2 * SQ_N * SQ_N], h[2 * SQ_N];
    double w[(SQ_N + 1) * (2 * SQ_N + 2) + 4 * SQ_N + 8];
    int jw[2 * SQ_N];
    int m1 = 2 * n;
    memset(E, 0, sizeof E);
    /* recover matrix E and vector f from L and g */
    int i2 = 0;
    for (int i = 0; i < n; ++i) {
        int i1 = n - i;
        double diag = sqrt(l[i2]);
        for (int k = 0; k < i1; ++k) E[(i + k) * n + i] = l[i2 + k] * diag; /* row i of E */
        E[i * n + i] = diag;
        f[i] = (g[i] - bl_dot(i, &E[i * n], 1, f, 1)) / diag;
        i2 += i1;
    }
    for (int i = 0; i < n; ++i) f[i] = -f[i];
    /* G = [+I; -I], h = [xl; -xu] */
    memset(G, 0, sizeof(double) * (size_t)(m1 * n));
    for (int i = 0; i < n; ++i) {
        G[i * m1 + i] = 1.0;
        G[i * m1 + n + i] = -1.0;
        h[i] = xl[i];
        h[n + i] = -xu[i];
    }
    double xnorm;
    /* LSEI with mc = 0 only copies E, f, G and calls LSI. */
    int mode = lsi(E, f, G, h, n, n, m1, m1, n, s, &xnorm, w, jw);
    if (mode == 1) {
        /* NLopt (SGJ 2010): enforce the bounds against roundoff. */
        for (int i = 0; i < n; ++i) {
            if (s[i] < xl[i]) s[i] = xl[i];
            else if (s[i] > xu[i]) s[i] = xu[i];
        }
    }
    return mode;
}

int ok_lsq_direction(int n, const double *l, const double *g, const double *lo, const double *hi,
                     double *s) {
    return lsq_box(n, l, g, lo, hi, s);
}

/* Fletcher-Powell composite-t rank-one update  LDL' := LDL' + sigma z z'.
 * a: packed LDL' (columnwise); z is destroyed; w workspace (sigma < 0 only). */
static void ldl_update(int n, double *a, double *z, double sigma, double *w) {
    if (sigma == 0.0) return;
    int ij = 0;
    double t = 1.0 / sigma;
    if (sigma < 0.0) {
        /* prepare negative update */
        for (int i = 0; i < n; ++i) w[i] = z[i];
        for (int i = 0; i < n; ++i) {
            double v = w[i];
            t += v * v / a[ij];
            for (int j = i + 1; j < n; ++j) { ++ij; w[j] -= v * a[ij]; }
            ++ij;
        }
        if (t >= 0.0) t = EPMACH / sigma;
        for (int i = 0; i < n; ++i) {
            int j = n - 1 - i;
            ij -= i + 1;
            double u = w[j];
            w[j] = t;
            t -= u * u / a[ij];
        }
    }
    /* here updating begins */
    for (int i = 0; i < n; ++i) {
        double v = z[i];
        double delta = v / a[ij];
        double tp = (sigma < 0.0) ? w[i] : t + delta * v;
        double alpha = tp / t;
        a[ij] = alpha * a[ij];
        if (i == n - 1) return;
        double beta = delta / tp;
        if (alpha > 4.0) {
            double gamma = t / tp;
            for (int j = i + 1; j < n; ++j) {
                ++ij;
                double u = a[ij];
                a[ij] = gamma * u + beta * z[j];
                z[j] -= v * u;
            }
        } else {
            for (int j = i + 1; j < n; ++j) {
                ++ij;
                z[j] -= v * a[ij];
                a[ij] += beta * z[j];
            }
        }
        ++ij;
        t = tp;
    }
}

/* Reverse-communication state of Kraft's SLSQPB body for m = 0. */
typedef struct {
    int n;
    double x[SQ_N], x0[SQ_N], g[SQ_N], s[SQ_N], u[SQ_N], v[SQ_N];
    double l[SQ_N * (SQ_N + 1) / 2 + 1];
    double f, f0, t0, gs, h1, h2, h3, h4, t, alpha;
    int iter, ireset, line;
} slsqp_state;

enum { SQ_MODE_INIT = 0, SQ_MODE_FEVAL = 1, SQ_MODE_FGEVAL = -2, SQ_MODE_GRAD = -1 };

/* One call of SLSQPB.  In: mode 0 (first call; f, g at x set), 1 / -2
 * (function [and gradient] evaluated at x), -1 (gradient evaluated).
 * Out: mode 1 / -2 (evaluate at x), -1 (line search done, gradient wanted),
 * or a terminal mode (3..9).  acc = 0 (NLopt does its own convergence tests),
 * so the mode-0 exits of the original never fire. */
static int slsqpb(slsqp_state *st, const double *xl, const double *xu, int mode) {
    const double alfmin = 0.1;
    const int n = st->n, n1 = n + 1, n2 = n1 * n / 2;
    double w[SQ_N];
    int start_line_trial = 0;

    if (mode == SQ_MODE_GRAD) goto L260;
    if (mode != SQ_MODE_INIT) goto L220;

    /* L100: initialisation */
    st->iter = 0;
    st->ireset = 0;
    for (int i = 0; i < n; ++i) st->s[i] = 0.0;
L110: /* reset BFGS matrix */
    ++st->ireset;
    if (st->ireset > 5) goto L255;
    for (int i = 0; i < n2; ++i) st->l[i] = 0.0;
    {
        int j = 0;
        for (int i = 0; i < n; ++i) { st->l[j] = 1.0; j += n1 - (i + 1); }
    }
L130: /* main iteration: search direction, steplength, LDL'-update */
    ++st->iter;
    /* (iteration limit itermx is disabled by NLopt: iter = 0 passed in) */
    for (int i = 0; i < n; ++i) { st->u[i] = xl[i] - st->x[i]; st->v[i] = xu[i] - st->x[i]; }
    st->h4 = 1.0;
    {
        int lmode = lsq_box(n, st->l, st->g, st->u, st->v, st->s);
        if (lmode != 1) return lmode; /* modes 3,4,5: LSQ sub-problem failed */
    }
    /* update multipliers for L1-test (m = 0: v = g) */
    for (int i = 0; i < n; ++i) st->v[i] = st->g[i];
    st->f0 = st->f;
    for (int i = 0; i < n; ++i) st->x0[i] = st->x[i];
    st->gs = bl_dot(n, st->g, 1, st->s, 1);
    st->h1 = fabs(st->gs);
    st->h2 = 0.0;
    /* acc == 0: "h1 < acc && h2 < acc" never holds */
    st->h1 = 0.0;
    st->t0 = st->f;
    /* check descent direction */
    st->h3 = st->gs - st->h1 * st->h4;
    if (st->h3 >= 0.0) goto L110;
    /* line search with an L1 test function (inexact) */
    st->line = 0;
    st->alpha = 1.0;
    start_line_trial = 1;
L190:
    (void)start_line_trial;
    ++st->line;
    st->h3 = st->alpha * st->h3;
    for (int i = 0; i < n; ++i) st->s[i] *= st->alpha;
    for (int i = 0; i < n; ++i) st->x[i] = st->x0[i];
    for (int i = 0; i < n; ++i) st->x[i] += st->s[i];
    /* NLopt (SGJ 2010): keep roundoff from pushing x past the bounds */
    for (int i = 0; i < n; ++i) {
        if (st->x[i] < xl[i]) st->x[i] = xl[i];
        else if (st->x[i] > xu[i]) st->x[i] = xu[i];
    }
    /* NLopt (SGJ 2010): the first trial is evaluated with its gradient */
    return (st->line == 1) ? SQ_MODE_FGEVAL : SQ_MODE_FEVAL;

L220: /* function evaluated: L1 merit (m = 0: t = f) */
    st->t = st->f;
    st->h1 = st->t - st->t0;
    if (isfinite(st->h1)) {
        if (st->h1 <= st->h3 / 10.0 || st->line > 10) goto L240;
        {
            double a = st->h3 / ((st->h3 - st->h1) * 2.0);
            st->alpha = (a > alfmin) ? a : alfmin;
        }
    } else {
        double a = st->alpha * 0.5;
        st->alpha = (a > alfmin) ? a : alfmin;
    }
    goto L190;
L240: /* check convergence: acc == 0 -> never converged here */
    st->h3 = 0.0;
    return SQ_MODE_GRAD;
L255: /* relaxed convergence after 5 resets: tol = 10*acc = 0 -> mode 8 */
    return 8;

L260: /* gradient evaluated: BFGS update of the LDL' factors */
    for (int i = 0; i < n; ++i) st->u[i] = st->g[i] - st->v[i];
    { /* v = L D L' s */
        int k = -1;
        for (int i = 0; i < n; ++i) {
            double h = 0.0;
            ++k;
            for (int j = i + 1; j < n; ++j) { ++k; h += st->l[k] * st->s[j]; }
            st->v[i] = st->s[i] + h;
        }
        k = 0;
        for (int i = 0; i < n; ++i) { st->v[i] = st->l[k] * st->v[i]; k += n1 - (i + 1); }
        for (int i = n - 1; i >= 0; --i) {
            double h = 0.0;
            k = i;
            for (int j = 0; j < i; ++j) { h += st->l[k] * st->v[j]; k += n - (j + 1); }
            st->v[i] += h;
        }
    }
    st->h1 = bl_dot(n, st->s, 1, st->u, 1);
    st->h2 = bl_dot(n, st->s, 1, st->v, 1);
    st->h3 = st->h2 * 0.2;
    if (st->h1 < st->h3) {
        st->h4 = (st->h2 - st->h3) / (st->h2 - st->h1);
        st->h1 = st->h3;
        for (int i = 0; i < n; ++i) st->u[i] *= st->h4;
        for (int i = 0; i < n; ++i) st->u[i] += (1.0 - st->h4) * st->v[i];
    }
    ldl_update(n, st->l, st->u, 1.0 / st->h1, w);
    ldl_update(n, st->l, st->v, -1.0 / st->h2, w);
    goto L130;
}

/* NLopt stopping helpers (nlopt/src/util/stop.c). */
static int relstop(double vold, double vnew, double reltol, double abstol) {
    if (isinf(vold)) return 0;
    return fabs(vnew - vold) < abstol
           || fabs(vnew - vold) < reltol * (fabs(vnew) + fabs(vold)) * 0.5
           || (reltol > 0 && vnew == vold);
}

/* nlopt_stop_x with xtol_rel = 0 and xtol_abs[i] = tol_dx.  NLopt >= 2.6.2 (rust-nlopt 0.8
 * bundles 2.7.1) first returns 1 when ||x - oldx|| <= xtol_rel * ||x||, i.e. here when the step
 * is exactly zero; NLopt 2.5 has only the per-coordinate test (ok_set_stop_x_zero(0)).  The
 * rule can only change a result when ftol_abs <= 0: a zero step leaves f unchanged and the
 * ftol test, which comes first in slsqp.c, fires on 0 < ftol_abs. */
static int g_stop_x_zero = 1;
void ok_set_stop_x_zero(int on) { g_stop_x_zero = on; }
static int stop_x(int n, const double *x, const double *oldx, double xtol_abs) {
    if (g_stop_x_zero) {
        int zero = 1;
        for (int i = 0; i < n; ++i) zero = zero && (x[i] == oldx[i]);
        if (zero) return 1;
    }
    for (int i = 0; i < n; ++i)
        if (fabs(x[i] - oldx[i]) >= xtol_abs) return 0;
    return 1;
}

#define OK_MAX_EVALS_CAP 100000

/* nlopt_slsqp() driver + the closure of lib.rs:301-391 for one restart. */
/* find_any runs (ok_ik with early_exit == 2, the baseline leg of bench.py): the word the objective callback
 * looks at before every evaluation -- lib.rs:308 `if should_exit.load() { force_stop }`.  NULL: never stop. */
static __thread const int *ok_tls_should_exit = NULL;

void ok_solve_restart(const ok_chain *c, const ok_config *cfg, const ok_pose *target,
                      const ok_pose *ee_offset, const double *x0, uint64_t restart_index,
                      ok_restart_result *out, double *trace, int trace_cap, int *trace_len) {
    const int n = c->n_pos;
    /* lib.rs:283-293 */
    const double tol_df = (cfg->tol_df > 0.0) ? cfg->tol_df : 1e-3 * cfg->tol_f;
    const double stopval = cfg->tol_f;   /* set_stopval, lib.rs:345 */
    const double ftol_abs = tol_df;      /* set_ftol_abs, lib.rs:346 */
    const double xtol_abs = cfg->tol_dx; /* set_xtol_abs1, lib.rs:347 */
    slsqp_state st;
    memset(&st, 0, sizeof st);
    st.n = n;
    int tl = 0;

    /* lib.rs:366-370 */
    if (restart_index == 0) memcpy(st.x, x0, (size_t)n * sizeof(double));
    else ok_restart_seed(c, restart_index, st.x);

    double minf = HUGE_VAL, fprev = HUGE_VAL, xprev[SQ_N], xbest[SQ_N];
    memcpy(xbest, st.x, (size_t)n * sizeof(double));
    memcpy(xprev, st.x, (size_t)n * sizeof(double));
    int ret = 0, mode = 0, prev_mode = 0, nevals = 0;

    /* NLopt: "eval once before calling slsqp the first time" */
    int do_eval = 1, want_grad = 1;
    for (;;) {
        if (do_eval && ok_tls_should_exit && __atomic_load_n(ok_tls_should_exit, __ATOMIC_RELAXED)) {
            ret = OK_RES_FORCED_STOP; /* lib.rs:308-311 */
            break;
        }
        if (do_eval) {
            st.f = ok_eval(c, target, ee_offset, cfg->linear_weight, cfg->angular_weight, st.x,
                           want_grad ? st.g : NULL);
            ++nevals;
            if (trace && tl < trace_cap) {
                memcpy(&trace[(size_t)tl * (size_t)(n + 1)], st.x, (size_t)n * sizeof(double));
                trace[(size_t)tl * (size_t)(n + 1) + (size_t)n] = st.f;
                ++tl;
            }
        }
        prev_mode = mode;
        /* update best point so far */
        if (st.f < minf) {
            minf = st.f;
            memcpy(xbest, st.x, (size_t)n * sizeof(double));
        }
        /* mode == -1: a line search completed; only then test ftol / xtol */
        if (mode == SQ_MODE_GRAD) {
            if (!isinf(fprev)) {
                if (relstop(fprev, st.f, 0.0, ftol_abs)) ret = OK_RES_FTOL_REACHED;
                else if (stop_x(n, st.x, xprev, xtol_abs)) ret = OK_RES_XTOL_REACHED;
            }
            fprev = st.f;
            memcpy(xprev, st.x, (size_t)n * sizeof(double));
        }
        /* additional termination tests (maxeval / maxtime unset) */
        if (minf < stopval) ret = OK_RES_STOPVAL_REACHED;
        if (ret != 0) break;
        if (nevals >= OK_MAX_EVALS_CAP) { ret = OK_RES_ITER_CAP; break; }

        mode = slsqpb(&st, c->lb, c->ub, mode);
        switch (mode) {
        case SQ_MODE_GRAD:
            /* NLopt: "if (prev_mode == -2 && !want_grad) break;  just evaluated this point" */
            do_eval = (prev_mode != SQ_MODE_FGEVAL);
            want_grad = 1;
            break;
        case SQ_MODE_FGEVAL: do_eval = 1; want_grad = 1; break;
        case SQ_MODE_FEVAL: do_eval = 1; want_grad = 0; break;
        case 8: /* positive directional derivative: relaxed test against (f0, x0) */
            ret = OK_RES_ROUNDOFF_LIMITED;
            if (relstop(st.f0, st.f, 0.0, ftol_abs)) ret = OK_RES_FTOL_REACHED;
            else if (stop_x(n, st.x, st.x0, xtol_abs)) ret = OK_RES_XTOL_REACHED;
            break;
        case 5: case 6: case 7: ret = OK_RES_ROUNDOFF_LIMITED; break;
        case 3: case 4: case 9: ret = OK_RES_FAILURE; break;
        default: ret = OK_RES_INVALID_ARGS; break;
        }
        if (ret != 0) break;
    }

    out->result = ret;
    out->n_evals = nevals;
    out->n_iters = st.iter;
    out->f = minf;
    memset(out->x, 0, sizeof out->x);
    memcpy(out->x, xbest, (size_t)n * sizeof(double));
    /* lib.rs:376-379 */
    out->success = (cfg->tol_f >= 0. && ret == OK_RES_STOPVAL_REACHED)
                   || (cfg->tol_df >= 0. && ret == OK_RES_FTOL_REACHED)
                   || (cfg->tol_dx >= 0. && ret == OK_RES_XTOL_REACHED);
    if (trace_len) *trace_len = tl;
}

/* ======================================================================= */
/* lib.rs:241-415 -- restart fan-out and selection (max_time == 0)          */
/* ======================================================================= */

typedef struct {
    const ok_chain *c; const ok_config *cfg; const ok_pose *target; const ok_pose *ee_offset;
    const double *x0;
    uint64_t begin, end;
    uint64_t next;          /* shared restart counter (rayon analogue) */
    int early_exit;         /* 1: deterministic reading (lowest index wins); 2: find_any (lib.rs:409-412) */
    int any_found;          /* find_any: the reference's should_exit word (lib.rs:269, 382-384) */
    uint64_t first_success; /* lowest successful index seen so far (Speed) */
    ok_restart_result *per_restart;
    /* winner state */
    int have; uint64_t winner; double key; double x[SQ_N]; double f;
    uint64_t n_run;
    pthread_mutex_t mu;
} ik_job;

static double dist_to_seed(int n, const double *x, const double *x0) {
    /* nalgebra metric_distance = (x - x0).norm(), sequential sum for n < 8 */
    double acc = 0.0;
    for (int i = 0; i < n; ++i) { double d = x[i] - x0[i]; acc += d * d; }
    return sqrt(acc);
}

static void *ik_worker(void *arg) {
    ik_job *job = (ik_job *)arg;
    const int n = job->c->n_pos;
    const int speed = (job->cfg->solution_mode == 2);
    for (;;) {
        uint64_t i = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
        if (i >= job->end) break;
        /* Speed + early exit: restarts above a known success cannot win (the
         * reference's should_exit flag, lib.rs:308,382-384, in its 1-thread
         * deterministic reading: lowest index wins). */
        if (speed && job->early_exit == 1
            && i > __atomic_load_n(&job->first_success, __ATOMIC_RELAXED))
            break;
        /* find_any (the reference's own multi-thread rule): ANY success ends every other restart at its next
         * objective call -- the answer is whichever restart got there first */
        if (speed && job->early_exit == 2) {
            if (__atomic_load_n(&job->any_found, __ATOMIC_RELAXED)) break;
            ok_tls_should_exit = &job->any_found;
        }
        ok_restart_result r;
        ok_solve_restart(job->c, job->cfg, job->target, job->ee_offset, job->x0, i, &r, NULL, 0,
                         NULL);
        ok_tls_should_exit = NULL;
        if (job->per_restart) job->per_restart[i - job->begin] = r;
        pthread_mutex_lock(&job->mu);
        job->n_run += 1;
        if (r.success) {
            double key = speed ? (double)i : dist_to_seed(n, r.x, job->x0);
            int better = !job->have || (job->early_exit != 2 && (key < job->key || (key == job->key && i < job->winner)));
            if (better) {
                job->have = 1; job->winner = i; job->key = key; job->f = r.f;
                memcpy(job->x, r.x, sizeof job->x);
            }
            if (speed && i < job->first_success)
                __atomic_store_n(&job->first_success, i, __ATOMIC_RELAXED);
            if (speed && job->early_exit == 2) __atomic_store_n(&job->any_found, 1, __ATOMIC_RELAXED);
        }
        pthread_mutex_unlock(&job->mu);
    }
    return NULL;
}

/* A persistent pool of worker threads for ok_ik (the baseline leg of bench.py: rayon keeps its workers between calls,
 * lib.rs:297-300; creating sixteen threads per call costs more than a Panda ik() takes).  ok_pool_start(n): n workers
 * parked on a condition variable; ok_ik(..., n_threads == n) then hands them the job instead of creating threads. */
#define OK_POOL_MAX 256
static struct {
    pthread_t th[OK_POOL_MAX];
    int n, quit, pending;
    unsigned long gen;
    ik_job *job;
    pthread_mutex_t mu;
    pthread_mutex_t user;  /* one ok_ik call at a time hands a job to the pool */
    pthread_cond_t go, done;
} ok_pool = {.n = 0, .mu = PTHREAD_MUTEX_INITIALIZER, .user = PTHREAD_MUTEX_INITIALIZER, .go = PTHREAD_COND_INITIALIZER,
             .done = PTHREAD_COND_INITIALIZER};

static void *ok_pool_worker(void *arg) {
    (void)arg;
    unsigned long seen = 0;
    for (;;) {
        pthread_mutex_lock(&ok_pool.mu);
        while (ok_pool.gen == seen && !ok_pool.quit) pthread_cond_wait(&ok_pool.go, &ok_pool.mu);
        if (ok_pool.quit) { pthread_mutex_unlock(&ok_pool.mu); return NULL; }
        seen = ok_pool.gen;
        ik_job *job = ok_pool.job;
        pthread_mutex_unlock(&ok_pool.mu);
        ik_worker(job);
        pthread_mutex_lock(&ok_pool.mu);
        if (--ok_pool.pending == 0) pthread_cond_signal(&ok_pool.done);
        pthread_mutex_unlock(&ok_pool.mu);
    }
}

void ok_pool_stop(void) {
    if (!ok_pool.n) return;
    pthread_mutex_lock(&ok_pool.mu);
    ok_pool.quit = 1;
    pthread_cond_broadcast(&ok_pool.go);
    pthread_mutex_unlock(&ok_pool.mu);
    for (int t = 0; t < ok_pool.n; ++t) pthread_join(ok_pool.th[t], NULL);
    ok_pool.n = 0; ok_pool.quit = 0;
}

int ok_pool_start(int n_threads) {
    ok_pool_stop();
    if (n_threads < 2) return 0;
    if (n_threads > OK_POOL_MAX) n_threads = OK_POOL_MAX;
    for (int t = 0; t < n_threads; ++t)
        if (pthread_create(&ok_pool.th[t], NULL, ok_pool_worker, NULL) != 0) { ok_pool.n = t; ok_pool_stop(); return -1; }
    ok_pool.n = n_threads;
    return n_threads;
}

int ok_ik(const ok_chain *c, const ok_config *cfg, const ok_pose *target,
          const ok_pose *ee_offset, const double *x0, uint64_t restart_begin,
          uint64_t restart_end, int n_threads, int early_exit, uint64_t *winner, double *x_out,
          double *f_out, ok_restart_result *per_restart, uint64_t *n_restarts_run) {
    ik_job job;
    memset(&job, 0, sizeof job);
    job.c = c; job.cfg = cfg; job.target = target; job.ee_offset = ee_offset; job.x0 = x0;
    job.begin = restart_begin; job.end = restart_end; job.next = restart_begin;
    job.early_exit = per_restart ? 0 : early_exit;
    job.first_success = UINT64_MAX;
    job.per_restart = per_restart;
    pthread_mutex_init(&job.mu, NULL);
    if (n_threads <= 1) {
        ik_worker(&job);
    } else if (ok_pool.n == n_threads) {
        pthread_mutex_lock(&ok_pool.user);
        pthread_mutex_lock(&ok_pool.mu);
        ok_pool.job = &job;
        ok_pool.pending = ok_pool.n;
        ++ok_pool.gen;
        pthread_cond_broadcast(&ok_pool.go);
        while (ok_pool.pending) pthread_cond_wait(&ok_pool.done, &ok_pool.mu);
        pthread_mutex_unlock(&ok_pool.mu);
        pthread_mutex_unlock(&ok_pool.user);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
        for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, ik_worker, &job);
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
        free(th);
    }
    pthread_mutex_destroy(&job.mu);
    if (n_restarts_run) *n_restarts_run = job.n_run;
    if (!job.have) return 0;
    if (winner) *winner = job.winner;
    if (x_out) memcpy(x_out, job.x, (size_t)c->n_pos * sizeof(double));
    if (f_out) *f_out = job.f;
    return 1;
}

/* Many independent ik() calls (BASELINE config 5 on the CPU, bench.py's baseline leg): target t of T gets one
 * single-threaded ok_ik over restart indices [0, n_restarts) with early exit; the targets are handed out to
 * n_threads threads from a shared counter (what a user of the reference would write with rayon over targets). */
typedef struct {
    const ok_chain *c; const ok_config *cfg; const ok_pose *targets; const ok_pose *ee_offset; const double *x0s;
    uint64_t n_restarts; int T; int next; int32_t *found; double *xs;
} many_job;

static void *many_worker(void *arg) {
    many_job *job = (many_job *)arg;
    const int n = job->c->n_pos;
    for (;;) {
        int t = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
        if (t >= job->T) break;
        uint64_t w = 0; double f = 0.0;
        double x[SQ_N];
        int ok = ok_ik(job->c, job->cfg, &job->targets[t], job->ee_offset, job->x0s + (size_t)t * (size_t)n, 0,
                       job->n_restarts, 1, 1, &w, x, &f, NULL, NULL);
        job->found[t] = ok;
        if (ok && job->xs) memcpy(job->xs + (size_t)t * (size_t)n, x, (size_t)n * sizeof(double));
    }
    return NULL;
}

int ok_ik_many(const ok_chain *c, const ok_config *cfg, const ok_pose *targets, const ok_pose *ee_offset,
               const double *x0s, int T, uint64_t n_restarts, int n_threads, int32_t *found, double *xs) {
    many_job job = {c, cfg, targets, ee_offset, x0s, n_restarts, T, 0, found, xs};
    if (n_threads <= 1) { many_worker(&job); return 0; }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, many_worker, &job);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th);
    return 0;
}
