// ik_wide.hpp -- chains with 9 .. 16 joint positions: one restart per lane, joint count at run time.
//
// The reference accepts any chain length (KinematicChain::num_positions,
// /root/reference/crates/optik/src/kinematics.rs:107-110); the tuned solvers of this library stop at
// n = 8 (a quad lane owns two joints, the NNLS permutation is sixteen nibbles, one ChaCha block per
// restart).  This header is the general form for the chains beyond that: the closure of
// lib.rs:301-391 and the SLSQP it calls (NLopt's: Kraft's SLSQPB / LSQ / LSI / LDP, Lawson-Hanson
// NNLS / H12, the Fletcher-Powell LDL' update, NLopt's stopping rules) written as plain loops over a
// run-time n.  No register tiling: ~1.9 K doubles of state per restart live either in a lane-strided
// HBM workspace ([slot][64 lanes]: the 64 lanes of a wave touch one 512-byte line per slot; the
// throughput form, HBM-bound) or, one restart per wave, in the wave's LDS (the latency form: a single
// ik() call's rounds) -- the solver is a template over that choice (WPG / WPL).
//
// Results are bit-identical to the CPU oracle's ok_solve_restart() (tests/test_gpu_wide.py) and, on
// chains of at most 8 joints (OPTIK_SOLVE_KERNEL=general), to the tuned solvers': the loops below
// perform the arithmetic of the textbook routines in the textbook order -- loads are grouped four
// terms at a time, sums are not reassociated, and LSI / LDP skip only what the box problem's
// structure makes an exact zero (w_lsq_box) -- and the objective is ik_eval.hpp's operation sequence
// with the joint loop rolled.
#pragma once

#include "ik_wide_launch.hpp"

namespace optik {

// Where a restart's arrays live.  WPG: the lane-strided HBM workspace of a wave -- element k of an
// array lives 64 doubles after element k - 1, so the 64 lanes of a wave touch one 512-byte line per
// slot.  WPL: LDS, contiguous -- the form of launches with one restart per wave (a single ik() call's
// rounds: the restart's 15 KB fit the wave's LDS, and a dependent access costs an LDS round trip
// instead of an L2 / HBM one).
struct WPG {
    double *p;
    __device__ __forceinline__ double &operator[](int k) const { return p[(size_t)k * 64]; }
    __device__ __forceinline__ WPG operator+(int k) const { return WPG{p + (size_t)k * 64}; }
};
typedef __attribute__((address_space(3))) double lds_double;
struct WPL {
    lds_double *p;
    __device__ __forceinline__ lds_double &operator[](int k) const { return p[k]; }
    __device__ __forceinline__ WPL operator+(int k) const { return WPL{p + k}; }
};
// WPC: the LDS form with the wave's 64 lanes working on its ONE restart together.  Every lane runs the
// solver on the shared arrays -- the scalar recurrences redundantly (same loads, same values, same
// stores) -- and the loops whose iterations are independent (the columns a reflection or a rotation is
// applied to, the rows of the transformed G, the columns of the dual problem, the Jacobian's columns)
// are dealt out lane by lane, each iteration still the sequential sum of the textbook loop: the same
// bits, a several times shorter dependent chain.
struct WPC {
    lds_double *p;
    __device__ __forceinline__ lds_double &operator[](int k) const { return p[k]; }
    __device__ __forceinline__ WPC operator+(int k) const { return WPC{p + k}; }
};
template <class WP> struct wp_coop { static constexpr bool value = false; };
template <> struct wp_coop<WPC> { static constexpr bool value = true; };
// first index offset and stride of a dealt-out loop (0 and 1 in the one-lane forms), and the wave-wide
// barrier that separates it from what reads its results
template <class WP> __device__ __forceinline__ int w_first() { return wp_coop<WP>::value ? (int)(threadIdx.x & 63u) : 0; }
template <class WP> __device__ __forceinline__ constexpr int w_step() { return wp_coop<WP>::value ? 64 : 1; }
template <class WP> __device__ __forceinline__ void w_sync() { if constexpr (wp_coop<WP>::value) __syncthreads(); }
// lane `src`'s value of v in every lane, `src` wave-uniform (two v_readlane_b32: a few cycles, where a
// shuffle by a vector index is an LDS-crossbar round trip)
__device__ __forceinline__ double w_lane_read(double v, int src) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// an element-wise loop: every element by the one lane in the one-lane forms, an element per lane in WPC
#define W_EACH(i, count) for (int i = w_first<WP>(); i < (count); i += w_step<WP>())

// workspace slots of one restart (doubles per lane)
namespace wide_ws {
constexpr int N = WIDE_MAX_DOF;
constexpr int X = 0, X0 = X + N, G = X0 + N, S = G + N, U = S + N, V = U + N, XBEST = V + N, XPREV = XBEST + N;
constexpr int L = XPREV + N;                        // packed LDL' (n (n + 1) / 2, + 1)
constexpr int E = L + N * (N + 1) / 2 + 1 + 7;      // LSQ: E (n x n)
constexpr int F = E + N * N;                        //      f (n)
constexpr int GG = F + N;                           //      G (2n x n)
constexpr int H = GG + 2 * N * N;                   //      h (2n)
constexpr int W = H + 2 * N;                        // LDP / NNLS workspace: (n + 1)(2n + 2) + 4n
constexpr int LW = W + (N + 1) * (2 * N + 2) + 4 * N + 8;  // LDL' update: w (n)
constexpr int TF = LW + N;                          // joint frames of the evaluation (7 per joint)
constexpr int GN = TF + 7 * N;                      // gradient of an evaluation that is not wanted
constexpr int SLOTS = GN + N;
}  // namespace wide_ws

// ---- objective and gradient (ik_eval.hpp's sequence, joint loop rolled) -------------------------

// Forward pass: joint frames into tf (7 doubles per joint), returns the end-effector pose
// (kinematics.rs:123-164).
template <class PQ, class PT>
__device__ Pose wide_forward(const WideChainDev &ch, const EvalParams &ep, int n, PQ q, PT tf) {
    Pose state;
    state.t = V3{0.0, 0.0, 0.0};
    state.q = Q4{0.0, 0.0, 0.0, 1.0};
    if constexpr (wp_coop<PT>::value) {
        // joint.origin * local_transform(q_j) of every joint at once (a joint per lane: kinematics.rs:142-158 forms
        // that product before it multiplies it onto the chain), then the chain product
        for (int j = w_first<PT>(); j < n; j += w_step<PT>()) {
            double s, c;
            sincos_dev(q[j] / 2.0, s, c);
            const Q4 local{ch.axis[j][0] * s, ch.axis[j][1] * s, ch.axis[j][2] * s, c};
            const Q4 jq = qmul(Q4{ch.origin[j][3], ch.origin[j][4], ch.origin[j][5], ch.origin[j][6]}, local);
            tf[7 * j + 3] = jq.i; tf[7 * j + 4] = jq.j; tf[7 * j + 5] = jq.k; tf[7 * j + 6] = jq.w;
        }
        w_sync<PT>();
#pragma unroll 1
        for (int j = 0; j < n; ++j) {
            Pose jt;
            jt.t = V3{ch.origin[j][0], ch.origin[j][1], ch.origin[j][2]};
            jt.q = Q4{tf[7 * j + 3], tf[7 * j + 4], tf[7 * j + 5], tf[7 * j + 6]};
            state = (j == 0) ? jt : pose_mul(state, jt);
            tf[7 * j + 0] = state.t.x; tf[7 * j + 1] = state.t.y; tf[7 * j + 2] = state.t.z;
            tf[7 * j + 3] = state.q.i; tf[7 * j + 4] = state.q.j; tf[7 * j + 5] = state.q.k; tf[7 * j + 6] = state.q.w;
        }
    } else
#pragma unroll 1
    for (int j = 0; j < n; ++j) {
        double s, c;
        sincos_dev(q[j] / 2.0, s, c);  // UnitQuaternion::from_axis_angle
        const Q4 local{ch.axis[j][0] * s, ch.axis[j][1] * s, ch.axis[j][2] * s, c};
        Pose jt;  // joint.origin * local_transform(q): the translation part is exact
        jt.t = V3{ch.origin[j][0], ch.origin[j][1], ch.origin[j][2]};
        jt.q = qmul(Q4{ch.origin[j][3], ch.origin[j][4], ch.origin[j][5], ch.origin[j][6]}, local);
        state = (j == 0) ? jt : pose_mul(state, jt);  // identity * jt is exact
        tf[7 * j + 0] = state.t.x; tf[7 * j + 1] = state.t.y; tf[7 * j + 2] = state.t.z;
        tf[7 * j + 3] = state.q.i; tf[7 * j + 4] = state.q.j; tf[7 * j + 5] = state.q.k; tf[7 * j + 6] = state.q.w;
    }
    if (ch.has_tip) state = pose_mul(state, load_pose(ch.origin[n]));
    return ep.has_ee_offset ? pose_mul(state, load_pose(ep.ee_offset)) : state;
}

// f at q; the gradient goes to g (objective.rs:40-110, kinematics.rs:166-196).
template <class PQ, class PT, class PG>
__device__ double wide_eval_fg(const WideChainDev &ch, const EvalParams &ep, const Pose target, int n, PQ q, PT tf,
                               PG g) {
    const Pose ee = wide_forward(ch, ep, n, q, tf);
    // X = T_target^-1 T_ee  (objective.rs:69-70)
    const Pose X = pose_inv_mul(target, ee);
    const V3 w = so3_log(X.q);
    const RotTerms rt = rot_terms(w);
    const M3 Jr = so3_right_jacobian(rt);      // math.rs:195
    const M3 Qm = se3_q_matrix(rt, X.t, Jr);   // math.rs:196
    const V3 elin = se3_log_linear(rt, X.t);   // math.rs:120-122
    V3 fl = elin, fa = w;
    if (!ep.skip_lin) fl = weight_block(target.q, elin, ep.w_lin);
    if (!ep.skip_ang) fa = weight_block(target.q, w, ep.w_ang);
    V3 gl = fl, ga = fa;
    if (!ep.grad_same_as_value) {
        gl = elin; ga = w;
        if (!ep.skip_lin2) gl = weight_block(target.q, elin, ep.w_lin2);
        if (!ep.skip_ang2) ga = weight_block(target.q, w, ep.w_ang2);
    }
    const double e2[6] = {2.0 * gl.x, 2.0 * gl.y, 2.0 * gl.z, 2.0 * ga.x, 2.0 * ga.y, 2.0 * ga.z};
    const double ef[6] = {fl.x, fl.y, fl.z, fa.x, fa.y, fa.z};
    double f = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) f += ef[i] * ef[i];

    const Q4 eeqc = qconj(ee.q);
    w_sync<PT>();
#pragma unroll 1
    for (int k = w_first<PT>(); k < n; k += w_step<PT>()) {  // (coop: a Jacobian column per lane)
        const V3 tk{tf[7 * k + 0], tf[7 * k + 1], tf[7 * k + 2]};
        const Q4 tq{tf[7 * k + 3], tf[7 * k + 4], tf[7 * k + 5], tf[7 * k + 6]};
        const V3 ax{ch.axis[k][0], ch.axis[k][1], ch.axis[k][2]};
        const V3 angular = qrot(tq, ax);
        const V3 d{ee.t.x - tk.x, ee.t.y - tk.y, ee.t.z - tk.z};
        const V3 linear = cross(angular, d);
        const V3 al = qrot(eeqc, angular);
        const V3 ll = qrot(eeqc, linear);
        const double lin[3] = {ll.x, ll.y, ll.z};
        const double ang[3] = {al.x, al.y, al.z};
        double jt[6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m) acc += Jr.m[r][m] * lin[m];
#pragma unroll
            for (int m = 0; m < 3; ++m) acc += Qm.m[r][m] * ang[m];
            jt[r] = acc;
            double acc2 = 0.0;  // lower-left block of Jlog6 is zero
#pragma unroll
            for (int m = 0; m < 3; ++m) acc2 += Jr.m[r][m] * ang[m];
            jt[r + 3] = acc2;
        }
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) acc += e2[r] * jt[r];
        g[k] = acc;
    }
    w_sync<PT>();
    return f;
}

// ---- RNG: one next_u64 per joint, as many ChaCha8 blocks as the chain needs (lib.rs:86-91, 358-370)

template <class PQ>
__device__ void wide_restart_seed(const uint32_t (&key)[8], const double *lb, const double *scale, uint64_t index,
                                  int n, PQ q) {
    uint32_t blk[16];
#pragma unroll 1
    for (int k0 = 0; k0 < n; k0 += 8) {
        chacha8_block(key, (uint64_t)(k0 / 8), index, blk);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = k0 + kk;
            if (k < n) {
                const uint64_t bits = (uint64_t)blk[2 * kk] | ((uint64_t)blk[2 * kk + 1] << 32);
                q[k] = uniform_inclusive(lb[k], scale[k], bits);
            }
        }
    }
}

// ---- SLSQP pieces over strided arrays --------------------------------------------------------------

// (the loads of four terms are issued together -- a dependent HBM round trip per term is what this
// path is bound by --, the sum is formed in the textbook order)
template <class WP>
__device__ inline double w_dot(int n, WP x, int incx, WP y, int incy) {
    double s = 0.0;
    int i = 0;
    for (; i + 4 <= n; i += 4) {
        const double x0 = x[i * incx], x1 = x[(i + 1) * incx], x2 = x[(i + 2) * incx], x3 = x[(i + 3) * incx];
        const double y0 = y[i * incy], y1 = y[(i + 1) * incy], y2 = y[(i + 2) * incy], y3 = y[(i + 3) * incy];
        s += x0 * y0; s += x1 * y1; s += x2 * y2; s += x3 * y3;
    }
    for (; i < n; ++i) s += x[i * incx] * y[i * incy];
    return s;
}

// The same dot product where the whole wave wants it (WPC, outside the dealt-out loops): lane i forms term i
// -- one LDS round trip for all of them --, the sum is then formed in the textbook order from lane reads.
// (n <= 64; the one-lane forms fall through to w_dot)
template <class WP>
__device__ inline double w_dot_all(int n, WP x, int incx, WP y, int incy) {
    if constexpr (wp_coop<WP>::value) {
        const int lane = (int)(threadIdx.x & 63u);
        double term = 0.0;
        if (lane < n) term = x[lane * incx] * y[lane * incy];
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += w_lane_read(term, i);
        return s;
    } else {
        return w_dot(n, x, incx, y, incy);
    }
}

// NLopt's dnrm2: scaled by the largest magnitude.
template <class WP>
__device__ inline double w_nrm2(int n, WP x, int incx) {
    double xmax = 0.0;
    for (int i = 0; i < n; ++i) { const double a = __builtin_fabs(x[i * incx]); if (a > xmax) xmax = a; }
    if (xmax == 0.0) return 0.0;
    const double scale = 1.0 / xmax;
    double sum = 0.0;
    for (int i = 0; i < n; ++i) { const double xs = scale * x[i * incx]; sum += xs * xs; }
    return xmax * __builtin_sqrt(sum);
}

// w_nrm2 where the whole wave wants it (WPC): an element per lane, maximum and sum in index order from lane reads.
template <class WP>
__device__ inline double w_nrm2_all(int n, WP x, int incx) {
    if constexpr (wp_coop<WP>::value) {
        const int lane = (int)(threadIdx.x & 63u);
        double xi = 0.0;
        if (lane < n) xi = x[lane * incx];
        double xmax = 0.0;
        for (int i = 0; i < n; ++i) { const double a = __builtin_fabs(w_lane_read(xi, i)); if (a > xmax) xmax = a; }
        if (xmax == 0.0) return 0.0;
        const double scale = 1.0 / xmax;
        const double xs = scale * xi;
        const double sq = xs * xs;
        double sum = 0.0;
        for (int i = 0; i < n; ++i) sum += w_lane_read(sq, i);
        return xmax * __builtin_sqrt(sum);
    } else {
        return w_nrm2(n, x, incx);
    }
}

// H12 where the whole wave wants it (WPC; NNLS's step five: the construction on the chosen column and its
// application to ONE vector): the elements l1 .. m of u (and of c) one per lane, maxima and sums in index order
// from lane reads, the update of c an element per lane.  ncv <= 1.
template <class WP>
__device__ inline void w_h12_all(int mode, int lpivot, int l1, int m, WP u, double &up, WP c, int ncv) {
    if (0 >= lpivot || lpivot >= l1 || l1 > m) return;
    const int lane = (int)(threadIdx.x & 63u);
    const int cnt = m - l1 + 1;
    const double up0 = u[lpivot - 1];
    double uj = 0.0;
    if (lane < cnt) uj = u[l1 - 1 + lane];
    double cl = __builtin_fabs(up0);
    if (mode != 2) {
        for (int t = 0; t < cnt; ++t) { const double sm = __builtin_fabs(w_lane_read(uj, t)); if (sm > cl) cl = sm; }
        if (cl <= 0.0) return;
        const double clinv = 1.0 / cl;
        double d = up0 * clinv;
        double sm = d * d;
        const double dj = uj * clinv;
        const double dj2 = dj * dj;
        for (int t = 0; t < cnt; ++t) sm += w_lane_read(dj2, t);
        cl *= __builtin_sqrt(sm);
        if (up0 > 0.0) cl = -cl;
        up = up0 - cl;
        u[lpivot - 1] = cl;
    } else if (cl <= 0.0) {
        return;
    }
    if (ncv <= 0) return;
    const double pivot = (mode != 2) ? cl : up0;  // u(lpivot) as it stands now
    double b = up * pivot;
    if (b >= 0.0) return;
    b = 1.0 / b;
    const double c0 = c[lpivot - 1];
    double cj = 0.0;
    if (lane < cnt) cj = c[l1 - 1 + lane];
    const double term = cj * uj;
    double sm = c0 * up;
    for (int t = 0; t < cnt; ++t) sm += w_lane_read(term, t);
    if (sm == 0.0) return;
    sm *= b;
    c[lpivot - 1] = c0 + sm * up;
    if (lane < cnt) c[l1 - 1 + lane] = cj + sm * uj;
    w_sync<WP>();
}

// Lawson-Hanson H12: construct (mode 1) / apply (mode 2) a Householder transformation.  u: pivot
// vector with stride iue; the ncv vectors of c have element stride ice, vector stride icv.
// lpivot, l1, m are 1-based.
template <class WP>
__device__ inline void w_h12(int mode, int lpivot, int l1, int m, WP u, int iue, double &up, WP c, int ice, int icv,
                             int ncv) {
    if (0 >= lpivot || lpivot >= l1 || l1 > m) return;
    double cl = __builtin_fabs(u[(lpivot - 1) * iue]);
    if (mode != 2) {
        {
            int j = l1;
            for (; j + 3 <= m; j += 4) {
                const double u0 = u[(j - 1) * iue], u1 = u[j * iue], u2 = u[(j + 1) * iue], u3 = u[(j + 2) * iue];
                double sm = __builtin_fabs(u0); if (sm > cl) cl = sm;
                sm = __builtin_fabs(u1); if (sm > cl) cl = sm;
                sm = __builtin_fabs(u2); if (sm > cl) cl = sm;
                sm = __builtin_fabs(u3); if (sm > cl) cl = sm;
            }
            for (; j <= m; ++j) { const double sm = __builtin_fabs(u[(j - 1) * iue]); if (sm > cl) cl = sm; }
        }
        if (cl <= 0.0) return;
        const double clinv = 1.0 / cl;
        double d = u[(lpivot - 1) * iue] * clinv;
        double sm = d * d;
        {
            int j = l1;
            for (; j + 3 <= m; j += 4) {
                const double u0 = u[(j - 1) * iue], u1 = u[j * iue], u2 = u[(j + 1) * iue], u3 = u[(j + 2) * iue];
                d = u0 * clinv; sm += d * d;
                d = u1 * clinv; sm += d * d;
                d = u2 * clinv; sm += d * d;
                d = u3 * clinv; sm += d * d;
            }
            for (; j <= m; ++j) { d = u[(j - 1) * iue] * clinv; sm += d * d; }
        }
        cl *= __builtin_sqrt(sm);
        if (u[(lpivot - 1) * iue] > 0.0) cl = -cl;
        up = u[(lpivot - 1) * iue] - cl;
        u[(lpivot - 1) * iue] = cl;
    } else if (cl <= 0.0) {
        return;
    }
    if (ncv <= 0) return;
    double b = up * u[(lpivot - 1) * iue];
    if (b >= 0.0) return;
    b = 1.0 / b;
    int i2 = 1 - icv + ice * (lpivot - 1);
    const int incr = ice * (l1 - lpivot);
    for (int j = 1; j <= ncv; ++j) {
        i2 += icv;
        int i3 = i2 + incr, i4 = i3;
        double sm = c[i2 - 1] * up;
        {
            int i = l1;
            for (; i + 3 <= m; i += 4) {  // four terms' loads together, summed in order
                const double c0 = c[i3 - 1], c1 = c[i3 - 1 + ice], c2 = c[i3 - 1 + 2 * ice], c3 = c[i3 - 1 + 3 * ice];
                const double u0 = u[(i - 1) * iue], u1 = u[i * iue], u2 = u[(i + 1) * iue], u3 = u[(i + 2) * iue];
                sm += c0 * u0; sm += c1 * u1; sm += c2 * u2; sm += c3 * u3;
                i3 += 4 * ice;
            }
            for (; i <= m; ++i) { sm += c[i3 - 1] * u[(i - 1) * iue]; i3 += ice; }
        }
        if (sm == 0.0) continue;
        sm *= b;
        c[i2 - 1] += sm * up;
        {
            // (c is a column other than the pivot vector u: the elements are independent)
            int i = l1;
            for (; i + 3 <= m; i += 4) {
                const double c0 = c[i4 - 1], c1 = c[i4 - 1 + ice], c2 = c[i4 - 1 + 2 * ice], c3 = c[i4 - 1 + 3 * ice];
                const double u0 = u[(i - 1) * iue], u1 = u[i * iue], u2 = u[(i + 1) * iue], u3 = u[(i + 2) * iue];
                c[i4 - 1] = c0 + sm * u0; c[i4 - 1 + ice] = c1 + sm * u1;
                c[i4 - 1 + 2 * ice] = c2 + sm * u2; c[i4 - 1 + 3 * ice] = c3 + sm * u3;
                i4 += 4 * ice;
            }
            for (; i <= m; ++i) { c[i4 - 1] += sm * u[(i - 1) * iue]; i4 += ice; }
        }
    }
}

// Lawson-Hanson NNLS: min ||A x - b|| s.t. x >= 0.  A is m x n column-major (leading dimension mda).
// Returns mode: 1 ok, 2 bad dimensions, 3 iteration count exceeded.
template <class WP>
__device__ inline int w_nnls(WP a, int mda, int m, int n, WP b, WP x, double &rnorm, WP w, WP z, int *indx) {
#define WA(i, j) a[((j) - 1) * mda + ((i) - 1)]
    const double factor = 0.01;
    if (m <= 0 || n <= 0) return 2;
    int mode = 1, iter = 0;
    const int itmax = 3 * n;
    for (int i = 1; i <= n; ++i) indx[i - 1] = i;
    int iz1 = 1, nsetp = 0, npp1 = 1;
    const int iz2 = n;
    int izmax = 0, j, jj = 0;
    double up = 0.0;
    bool finished = false;
    W_EACH(i, n) x[i] = 0.0;
    w_sync<WP>();

    while (!finished) {  // step two
        if (iz1 > iz2 || nsetp >= m) break;
        w_sync<WP>();
        for (int iz = iz1 + w_first<WP>(); iz <= iz2; iz += w_step<WP>()) {
            const int jc = indx[iz - 1];
            w[jc - 1] = w_dot(m - nsetp, a + ((jc - 1) * mda + (npp1 - 1)), 1, b + (npp1 - 1), 1);
        }
        w_sync<WP>();
        bool found = false;
        for (;;) {  // step three
            double wmax = 0.0;
            if constexpr (wp_coop<WP>::value) {  // (a candidate per lane, the comparisons in index order from lane reads)
                const int cnt = iz2 - iz1 + 1;
                const int lane = (int)(threadIdx.x & 63u);
                double cand = 0.0;
                if (lane < cnt) cand = w[indx[iz1 - 1 + lane] - 1];
                for (int t = 0; t < cnt; ++t) {
                    const double wv = w_lane_read(cand, t);
                    if (wv <= wmax) continue;
                    wmax = wv;
                    izmax = iz1 + t;
                }
            } else
            for (int iz = iz1; iz <= iz2; ++iz) {
                j = indx[iz - 1];
                if (w[j - 1] <= wmax) continue;
                wmax = w[j - 1];
                izmax = iz;
            }
            if (wmax <= 0.0) break;  // step four: KKT satisfied
            const int iz = izmax;
            j = indx[iz - 1];
            // step five
            const double asave = WA(npp1, j);
            if constexpr (wp_coop<WP>::value) w_h12_all(1, npp1, npp1 + 1, m, a + (j - 1) * mda, up, z, 0);
            else w_h12(1, npp1, npp1 + 1, m, a + (j - 1) * mda, 1, up, z, 1, 1, 0);
            const double unorm = w_nrm2_all(nsetp, a + (j - 1) * mda, 1);
            const double t = factor * __builtin_fabs(WA(npp1, j));
            const double d1 = unorm + t;
            if (d1 - unorm > 0.0) {
                W_EACH(i, m) z[i] = b[i];
                w_sync<WP>();
                if constexpr (wp_coop<WP>::value) w_h12_all(2, npp1, npp1 + 1, m, a + (j - 1) * mda, up, z, 1);
                else w_h12(2, npp1, npp1 + 1, m, a + (j - 1) * mda, 1, up, z, 1, 1, 1);
                if (z[npp1 - 1] / WA(npp1, j) > 0.0) found = true;
            }
            if (found) {
                W_EACH(i, m) b[i] = z[i];
                indx[iz - 1] = indx[iz1 - 1];
                indx[iz1 - 1] = j;
                ++iz1;
                nsetp = npp1;
                ++npp1;
                w_sync<WP>();
                for (int jz = iz1 + w_first<WP>(); jz <= iz2; jz += w_step<WP>()) {
                    const int jc = indx[jz - 1];
                    w_h12(2, nsetp, npp1, m, a + (j - 1) * mda, 1, up, a + (jc - 1) * mda, 1, mda, 1);
                }
                w_sync<WP>();
                w[j - 1] = 0.0;
                for (int i = npp1; i <= m; ++i) WA(i, j) = 0.0;
                break;
            }
            WA(npp1, j) = asave;
            w[j - 1] = 0.0;
        }
        if (!found) break;  // wmax <= 0: done

        // step six: solve the triangular system; then steps seven .. eleven
        for (;;) {
            for (int ip = nsetp; ip >= 1; --ip) {
                if (ip != nsetp) {
                    const double zp = z[ip];
                    int i = 0;
                    for (; i + 4 <= ip; i += 4) {
                        double zi[4], ai[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { zi[q] = z[i + q]; ai[q] = WA(i + q + 1, jj); }
#pragma unroll
                        for (int q = 0; q < 4; ++q) z[i + q] = zi[q] - zp * ai[q];
                    }
                    for (; i < ip; ++i) z[i] -= zp * WA(i + 1, jj);
                }
                jj = indx[ip - 1];
                z[ip - 1] /= WA(ip, jj);
            }
            ++iter;
            if (iter > itmax) { mode = 3; finished = true; break; }
            double alpha = 1.0;
            jj = 0;
            if constexpr (wp_coop<WP>::value) {  // (the quotients a position per lane, the comparisons in order from lane reads)
                const int lane = (int)(threadIdx.x & 63u);
                double zq = 1.0, tq = 0.0;
                if (lane < nsetp) {
                    zq = z[lane];
                    const double xl = x[indx[lane] - 1];
                    tq = -xl / (zq - xl);
                }
                for (int ip = 1; ip <= nsetp; ++ip) {
                    if (w_lane_read(zq, ip - 1) > 0.0) continue;
                    const double t = w_lane_read(tq, ip - 1);
                    if (alpha < t) continue;
                    alpha = t;
                    jj = ip;
                }
            } else
            for (int ip = 1; ip <= nsetp; ++ip) {
                if (z[ip - 1] > 0.0) continue;
                const int l = indx[ip - 1];
                const double t = -x[l - 1] / (z[ip - 1] - x[l - 1]);
                if (alpha < t) continue;
                alpha = t;
                jj = ip;
            }
            W_EACH(ipz, nsetp) {
                const int l = indx[ipz];
                x[l - 1] = (1.0 - alpha) * x[l - 1] + alpha * z[ipz];
            }
            w_sync<WP>();
            if (jj == 0) break;  // back to step two
            // step eleven: move coefficient i from set P to set Z
            int i = indx[jj - 1];
            bool failed = false;
            for (;;) {
                x[i - 1] = 0.0;
                ++jj;
                for (j = jj; j <= nsetp; ++j) {
                    const int ii = indx[j - 1];
                    indx[j - 2] = ii;
                    double c, s;
                    double ra = WA(j - 1, ii), rb = WA(j, ii);
                    rotg(ra, rb, c, s);
                    WA(j - 1, ii) = ra;
                    WA(j, ii) = rb;
                    const double t = ra;
                    w_sync<WP>();
                    if constexpr (wp_coop<WP>::value) {
                        for (int col = 1 + w_first<WP>(); col <= n; col += w_step<WP>()) {
                            const double xi = WA(j - 1, col), yi = WA(j, col);
                            WA(j - 1, col) = c * xi + s * yi;
                            WA(j, col) = c * yi - s * xi;
                        }
                    } else {  // rot over the two rows, every column (four columns' loads together)
                        int col = 1;
                        for (; col + 3 <= n; col += 4) {
                            double xi[4], yi[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) { xi[q] = WA(j - 1, col + q); yi[q] = WA(j, col + q); }
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                WA(j - 1, col + q) = c * xi[q] + s * yi[q];
                                WA(j, col + q) = c * yi[q] - s * xi[q];
                            }
                        }
                        for (; col <= n; ++col) {
                            const double xi = WA(j - 1, col), yi = WA(j, col);
                            WA(j - 1, col) = c * xi + s * yi;
                            WA(j, col) = c * yi - s * xi;
                        }
                    }
                    w_sync<WP>();
                    WA(j - 1, ii) = t;
                    WA(j, ii) = 0.0;
                    {
                        const double xi = b[j - 2], yi = b[j - 1];
                        b[j - 2] = c * xi + s * yi;
                        b[j - 1] = c * yi - s * xi;
                    }
                }
                npp1 = nsetp;
                --nsetp;
                --iz1;
                indx[iz1 - 1] = i;
                if (nsetp <= 0) { mode = 3; failed = true; break; }
                bool again = false;
                for (jj = 1; jj <= nsetp; ++jj) {
                    i = indx[jj - 1];
                    if (x[i - 1] <= 0.0) { again = true; break; }
                }
                if (!again) break;
            }
            if (failed) { finished = true; break; }
            W_EACH(k, m) z[k] = b[k];
            w_sync<WP>();
        }
    }
    {
        const int k = (npp1 < m) ? npp1 : m;
        rnorm = w_nrm2_all(m - nsetp, b + (k - 1), 1);
        if (npp1 > m) for (int i = 0; i < n; ++i) w[i] = 0.0;
    }
    return mode;
#undef WA
}

// Lawson-Hanson LDP: min ||x|| s.t. G x >= h.  G is m x n column-major (leading dimension mg).
template <class WP>
__device__ inline int w_ldp(WP g, int mg, int m, int n, WP h, WP x, double &xnorm, WP w, int *indx) {
    if (n <= 0) return 2;
    for (int i = 0; i < n; ++i) x[i] = 0.0;
    xnorm = 0.0;
    if (m == 0) return 1;
    int iw = 0;
    for (int j = 0; j < m; ++j) {
        int i = 0;
        for (; i + 4 <= n; i += 4) {
            const double g0 = g[i * mg + j], g1 = g[(i + 1) * mg + j], g2 = g[(i + 2) * mg + j], g3 = g[(i + 3) * mg + j];
            w[iw] = g0; w[iw + 1] = g1; w[iw + 2] = g2; w[iw + 3] = g3;
            iw += 4;
        }
        for (; i < n; ++i) w[iw++] = g[i * mg + j];
        w[iw++] = h[j];
    }
    const int if_ = iw;
    for (int i = 0; i < n; ++i) w[iw++] = 0.0;
    w[iw] = 1.0;
    const int n1 = n + 1;
    const int iz = iw + 1, iy = iz + n1, iwdual = iy + m;
    double rnorm;
    const int mode = w_nnls(w, n1, n1, m, w + if_, w + iy, rnorm, w + iwdual, w + iz, indx);
    if (mode != 1) return mode;
    if (rnorm <= 0.0) return 4;
    double fac = 1.0 - w_dot(m, h, 1, w + iy, 1);
    const double d1 = 1.0 + fac;
    if (d1 - 1.0 <= 0.0) return 4;
    fac = 1.0 / fac;
    for (int j = 0; j < n; ++j) x[j] = fac * w_dot(m, g + j * mg, 1, w + iy, 1);
    xnorm = w_nrm2(n, x, 1);
    for (int i = 0; i < m; ++i) w[i] = 0.0;
    for (int i = 0; i < m; ++i) w[i] += fac * w[iy + i];
    return 1;
}

// Kraft LSI: min ||E x - f|| s.t. G x >= h.  E is me x n (ld le), G mg x n (ld lg).
template <class WP>
__device__ inline int w_lsi(WP e, WP f, WP g, WP h, int le, int me, int lg, int mg, int n, WP x, double &xnorm, WP w,
                            int *jw) {
#define WE(i, j) e[((j) - 1) * le + ((i) - 1)]
#define WG(i, j) g[((j) - 1) * lg + ((i) - 1)]
    double t = 0.0;
    // QR factors of E and application to f
    for (int i = 1; i <= n; ++i) {
        const int j = (i + 1 < n) ? i + 1 : n;
        w_h12(1, i, i + 1, me, e + (i - 1) * le, 1, t, e + (j - 1) * le, 1, le, n - i);
        w_h12(2, i, i + 1, me, e + (i - 1) * le, 1, t, f, 1, 1, 1);
    }
    // transform G and h to get the least distance problem
    for (int i = 1; i <= mg; ++i) {
        for (int j = 1; j <= n; ++j) {
            if (!(__builtin_fabs(WE(j, j)) >= EPMACH)) return 5;
            WG(i, j) = (WG(i, j) - w_dot(j - 1, g + (i - 1), lg, e + (j - 1) * le, 1)) / WE(j, j);
        }
        h[i - 1] -= w_dot(n, g + (i - 1), lg, f, 1);
    }
    const int mode = w_ldp(g, lg, mg, n, h, x, xnorm, w, jw);
    if (mode != 1) return mode;
    // solution of the original problem
    for (int i = 0; i < n; ++i) x[i] += f[i];
    for (int i = n; i >= 1; --i) {
        const int j = (i + 1 < n) ? i + 1 : n;
        x[i - 1] = (x[i - 1] - w_dot(n - i, e + ((j - 1) * le + (i - 1)), le, x + (j - 1), 1)) / WE(i, i);
    }
    const int j = (n + 1 < me) ? n + 1 : me;
    t = w_nrm2(me - n, f + (j - 1), 1);
    xnorm = __builtin_sqrt(xnorm * xnorm + t * t);
    return 1;
#undef WE
#undef WG
}

// Kraft LSQ for m = meq = 0 with finite bounds: min ||E s - f||, E = D^1/2 L', f = -D^-1/2 L^-1 g,
// xl <= s <= xu, via LSEI (mc = 0) -> LSI -> LDP -> NNLS.  l: packed LDL'.  Returns the LSQ mode.
//
// LSI and LDP are written for what this problem hands them (w_lsi / w_ldp above are the general
// routines, kept for -DOPTIK_WIDE_GENERAL_LSI builds: same bits): E is upper triangular, so the sums
// of a reflection over the zeros below its diagonal are skipped (a sum that only gains +-0 keeps its
// value); G = [I; -I], so row n + i of the transformed G is row i negated -- only the top half is
// formed and stored, its leading zeros without their dot products -- and the solution's norm and the
// multipliers, which SLSQP does not read for m = 0, are not formed.  (A product by an exact zero is
// +-0 and x + (+-0) == x: every non-zero keeps its bits; an exact zero of the top half keeps its sign
// in the bottom half, as it does when the bottom row is computed.)  A third of the path's HBM
// traffic was those zeros.
__device__ __forceinline__ double w_mirror(double v) { return v == 0.0 ? v : -v; }

template <class WP>
__device__ inline int w_lsq_box(int n, WP ws, WP l, WP g, WP xl, WP xu, WP s) {
    WP E = ws + wide_ws::E, f = ws + wide_ws::F, G = ws + wide_ws::GG, h = ws + wide_ws::H, w = ws + wide_ws::W;
    int jw[2 * WIDE_MAX_DOF];
    const int m1 = 2 * n;
#ifdef OPTIK_WIDE_GENERAL_LSI
    for (int i = 0; i < n * n; ++i) E[i] = 0.0;
#endif
    // recover matrix E and vector f from L and g
    if constexpr (wp_coop<WP>::value) {
        // (the rows of E are independent: a row per lane; only f's forward substitution is a chain)
        W_EACH(i, n) {
            const int ii = i * n - (i * (i - 1)) / 2;  // index of l(i, i) in the packed factor
            const double diag = __builtin_sqrt(l[ii]);
            for (int k = 1; k < n - i; ++k) E[(i + k) * n + i] = l[ii + k] * diag;
            E[i * n + i] = diag;
        }
        w_sync<WP>();
        for (int i = 0; i < n; ++i) f[i] = (g[i] - w_dot_all(i, E + i * n, 1, f, 1)) / E[i * n + i];
    } else {
        int i2 = 0;
        for (int i = 0; i < n; ++i) {
            const int i1 = n - i;
            const double diag = __builtin_sqrt(l[i2]);
            for (int k = 0; k < i1; ++k) E[(i + k) * n + i] = l[i2 + k] * diag;  // row i of E
            E[i * n + i] = diag;
            f[i] = (g[i] - w_dot(i, E + i * n, 1, f, 1)) / diag;
            i2 += i1;
        }
    }
    W_EACH(i, n) f[i] = -f[i];
    w_sync<WP>();
#ifdef OPTIK_WIDE_GENERAL_LSI
    // G = [+I; -I], h = [xl; -xu]
    for (int i = 0; i < m1 * n; ++i) G[i] = 0.0;
    for (int i = 0; i < n; ++i) {
        G[i * m1 + i] = 1.0;
        G[i * m1 + n + i] = -1.0;
        h[i] = xl[i];
        h[n + i] = -xu[i];
    }
    double xnorm;
    const int mode = w_lsi(E, f, G, h, n, n, m1, m1, n, s, xnorm, w, jw);
#else
#define WE(i, j) E[((j) - 1) * n + ((i) - 1)]
#define WGT(i, j) G[((j) - 1) * n + ((i) - 1)]  // top half of the transformed G, n x n
    W_EACH(i, n) { h[i] = xl[i]; h[n + i] = -xu[i]; }
    w_sync<WP>();
    // LSI: QR factors of E and application to f (H12 with lpivot = i, l1 = i + 1, m = n; i = n: l1 > m, no-op).
    // (E is upper triangular: reflection i touches row i of E and f(i) only -- in WPC a reflection per lane)
    for (int i = 1 + w_first<WP>(); i < n; i += w_step<WP>()) {
        const double eii = WE(i, i);
        double cl = __builtin_fabs(eii);
        if (cl <= 0.0) continue;  // (mode 1 leaves the column alone; mode 2 finds the same zero pivot)
        const double clinv = 1.0 / cl;
        const double d = eii * clinv;
        const double sm0 = d * d;
        cl *= __builtin_sqrt(sm0);
        if (eii > 0.0) cl = -cl;
        const double up = eii - cl;
        WE(i, i) = cl;
        double b = up * cl;
        if (b >= 0.0) continue;  // (both applications return)
        b = 1.0 / b;
        int col = i + 1;
        for (; col + 3 <= n; col += 4) {  // row i of four columns at a time
            double ci[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) ci[q] = WE(i, col + q);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double sm = ci[q] * up;
                if (sm == 0.0) continue;
                sm *= b;
                WE(i, col + q) = ci[q] + sm * up;
            }
        }
        for (; col <= n; ++col) {
            const double ci = WE(i, col);
            double sm = ci * up;
            if (sm == 0.0) continue;
            sm *= b;
            WE(i, col) = ci + sm * up;
        }
        {
            const double fi = f[i - 1];
            double sm = fi * up;
            if (sm != 0.0) {
                sm *= b;
                f[i - 1] = fi + sm * up;
            }
        }
    }
    w_sync<WP>();
    // transform G and h to get the least distance problem: rows 1 .. n (row n + i is row i negated)
    for (int j = 1; j <= n; ++j)
        if (!(__builtin_fabs(WE(j, j)) >= EPMACH)) return 5;
    w_sync<WP>();
    for (int i = 1 + w_first<WP>(); i <= n; i += w_step<WP>()) {  // (coop: a row per lane)
        for (int j = 1; j < i; ++j) WGT(i, j) = 0.0 / WE(j, j);  // (0 - 0) / E(j, j)
        WGT(i, i) = 1.0 / WE(i, i);
        for (int j = i + 1; j <= n; ++j)
            WGT(i, j) = (0.0 - w_dot(j - i, G + ((i - 1) * n + (i - 1)), n, E + ((j - 1) * n + (i - 1)), 1)) / WE(j, j);
        const double dt = w_dot(n - i + 1, G + ((i - 1) * n + (i - 1)), n, f + (i - 1), 1);
        h[i - 1] -= dt;
        h[n + i - 1] -= w_mirror(dt);
    }
    w_sync<WP>();
    // LDP: the (n + 1) x 2n dual problem [G'; h'], NNLS, the primal solution
    int mode;
    {
        for (int i = 0; i < n; ++i) s[i] = 0.0;
        int iw = 0;
        for (int j = w_first<WP>(); j < m1; j += w_step<WP>()) {  // (coop: a column of the dual problem per lane)
            iw = j * (n + 1);
            const bool bottom = j >= n;
            const int r = bottom ? j - n : j;  // row of the top half
            int i = 0;
            for (; i + 4 <= n; i += 4) {
                double gv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) gv[q] = G[(i + q) * n + r];
#pragma unroll
                for (int q = 0; q < 4; ++q) w[iw + q] = bottom ? w_mirror(gv[q]) : gv[q];
                iw += 4;
            }
            for (; i < n; ++i) { const double gv = G[i * n + r]; w[iw++] = bottom ? w_mirror(gv) : gv; }
            w[iw++] = h[j];
        }
        iw = m1 * (n + 1);
        w_sync<WP>();
        const int if_ = iw;
        for (int i = 0; i < n; ++i) w[iw++] = 0.0;
        w[iw] = 1.0;
        const int n1 = n + 1;
        const int iz = iw + 1, iy = iz + n1, iwdual = iy + m1;
        double rnorm;
        mode = w_nnls(w, n1, n1, m1, w + if_, w + iy, rnorm, w + iwdual, w + iz, jw);
        if (mode == 1) {
            if (rnorm <= 0.0) mode = 4;
            else {
                double fac = 1.0 - w_dot_all(m1, h, 1, w + iy, 1);
                const double d1 = 1.0 + fac;
                if (d1 - 1.0 <= 0.0) mode = 4;
                else {
                    fac = 1.0 / fac;
                    for (int j = w_first<WP>(); j < n; j += w_step<WP>()) {
                        // dot(2n, column j of G, y): the top half, then the bottom half, one running sum
                        double acc = 0.0;
                        int r = 0;
                        for (; r + 4 <= n; r += 4) {
                            double gv[4], yv[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) { gv[q] = G[j * n + r + q]; yv[q] = w[iy + r + q]; }
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc += gv[q] * yv[q];
                        }
                        for (; r < n; ++r) acc += G[j * n + r] * w[iy + r];
                        r = 0;
                        for (; r + 4 <= n; r += 4) {
                            double gv[4], yv[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) { gv[q] = G[j * n + r + q]; yv[q] = w[iy + n + r + q]; }
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc += w_mirror(gv[q]) * yv[q];
                        }
                        for (; r < n; ++r) acc += w_mirror(G[j * n + r]) * w[iy + n + r];
                        s[j] = fac * acc;
                    }
                }
            }
        }
        w_sync<WP>();
    }
    if (mode == 1) {
        // solution of the original problem
        W_EACH(i, n) s[i] += f[i];
        w_sync<WP>();
        for (int i = n; i >= 1; --i) {
            const int j = (i + 1 < n) ? i + 1 : n;
            s[i - 1] = (s[i - 1] - w_dot_all(n - i, E + ((j - 1) * n + (i - 1)), n, s + (j - 1), 1)) / WE(i, i);
        }
    }
#undef WE
#undef WGT
#endif
    if (mode == 1) {
        // NLopt (SGJ 2010): enforce the bounds against roundoff
        W_EACH(i, n) {
            if (s[i] < xl[i]) s[i] = xl[i];
            else if (s[i] > xu[i]) s[i] = xu[i];
        }
        w_sync<WP>();
    }
    return mode;
}

// Fletcher-Powell composite-t rank-one update LDL' := LDL' + sigma z z' (z is destroyed).
template <class WP>
__device__ inline void w_ldl_update(int n, WP a, WP z, double sigma, WP w) {
    if (sigma == 0.0) return;
    if constexpr (wp_coop<WP>::value) {
        // WPC: lane j keeps z(j) (and w(j)) in a register and owns row j's share of every pivot's update; the
        // pivot's v = z(i) and the diagonal l(i, i) -- which only its own pivot changes -- come by lane read.  No
        // LDS round trip and no barrier on the pivot-to-pivot chain; the same operations on the same operands.
        const int lane = (int)(threadIdx.x & 63u);
        const bool mine = lane < n;
        double zr = mine ? (double)z[lane] : 0.0;
        double dr = mine ? (double)a[lane * n - (lane * (lane - 1)) / 2] : 1.0;
        double wr = 0.0;
        double t = 1.0 / sigma;
        if (sigma < 0.0) {
            wr = zr;
            for (int i = 0; i < n; ++i) {
                const double v = w_lane_read(wr, i);
                t += v * v / w_lane_read(dr, i);
                if (mine && lane > i) wr -= v * a[i * n - (i * (i - 1)) / 2 + (lane - i)];
            }
            if (t >= 0.0) t = EPMACH / sigma;
            for (int i = 0; i < n; ++i) {
                const int j = n - 1 - i;
                const double u = w_lane_read(wr, j);
                if (lane == j) wr = t;
                t -= u * u / w_lane_read(dr, j);
            }
        }
        for (int i = 0; i < n; ++i) {
            const int di = i * n - (i * (i - 1)) / 2;
            const double v = w_lane_read(zr, i);
            const double aii = w_lane_read(dr, i);
            const double delta = v / aii;
            const double tp = (sigma < 0.0) ? w_lane_read(wr, i) : t + delta * v;
            const double alpha = tp / t;
            if (lane == i) a[di] = alpha * aii;
            if (i == n - 1) break;
            const double beta = delta / tp;
            if (alpha > 4.0) {
                const double gamma = t / tp;
                if (mine && lane > i) {
                    const double u = a[di + (lane - i)];
                    a[di + (lane - i)] = gamma * u + beta * zr;
                    zr -= v * u;
                }
            } else {
                if (mine && lane > i) {
                    zr -= v * a[di + (lane - i)];
                    a[di + (lane - i)] += beta * zr;
                }
            }
            t = tp;
        }
        w_sync<WP>();
        return;
    }
    int ij = 0;
    double t = 1.0 / sigma;
    if (sigma < 0.0) {
        for (int i = 0; i < n; ++i) w[i] = z[i];
        for (int i = 0; i < n; ++i) {
            const double v = w[i];
            t += v * v / a[ij];
            if constexpr (wp_coop<WP>::value) {  // the row's elements, one per lane
                w_sync<WP>();
                for (int j = i + 1 + w_first<WP>(); j < n; j += w_step<WP>()) w[j] -= v * a[ij + (j - i)];
                w_sync<WP>();
                ij += n - 1 - i;
            } else {
                int j = i + 1;
                for (; j + 4 <= n; j += 4) {
                    double au[4], wj[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { au[q] = a[ij + 1 + q]; wj[q] = w[j + q]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[j + q] = wj[q] - v * au[q];
                    ij += 4;
                }
                for (; j < n; ++j) { ++ij; w[j] -= v * a[ij]; }
            }
            ++ij;
        }
        if (t >= 0.0) t = EPMACH / sigma;
        for (int i = 0; i < n; ++i) {
            const int j = n - 1 - i;
            ij -= i + 1;
            const double u = w[j];
            w[j] = t;
            t -= u * u / a[ij];
        }
    }
    for (int i = 0; i < n; ++i) {
        const double v = z[i];
        const double delta = v / a[ij];
        const double tp = (sigma < 0.0) ? w[i] : t + delta * v;
        const double alpha = tp / t;
        a[ij] = alpha * a[ij];
        if (i == n - 1) return;
        const double beta = delta / tp;
        if (alpha > 4.0) {
            const double gamma = t / tp;
            int j = i + 1;
            if constexpr (wp_coop<WP>::value) {
                w_sync<WP>();
                for (int jj = i + 1 + w_first<WP>(); jj < n; jj += w_step<WP>()) {
                    const double u = a[ij + (jj - i)];
                    a[ij + (jj - i)] = gamma * u + beta * z[jj];
                    z[jj] -= v * u;
                }
                w_sync<WP>();
                ij += n - 1 - i;
                j = n;
            }
            for (; j + 4 <= n; j += 4) {
                double au[4], zj[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { au[q] = a[ij + 1 + q]; zj[q] = z[j + q]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) { a[ij + 1 + q] = gamma * au[q] + beta * zj[q]; z[j + q] = zj[q] - v * au[q]; }
                ij += 4;
            }
            for (; j < n; ++j) {
                ++ij;
                const double u = a[ij];
                a[ij] = gamma * u + beta * z[j];
                z[j] -= v * u;
            }
        } else {
            int j = i + 1;
            if constexpr (wp_coop<WP>::value) {
                w_sync<WP>();
                for (int jj = i + 1 + w_first<WP>(); jj < n; jj += w_step<WP>()) {
                    const double zn = z[jj] - v * a[ij + (jj - i)];
                    z[jj] = zn;
                    a[ij + (jj - i)] += beta * zn;
                }
                w_sync<WP>();
                ij += n - 1 - i;
                j = n;
            }
            for (; j + 4 <= n; j += 4) {
                double au[4], zj[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { au[q] = a[ij + 1 + q]; zj[q] = z[j + q]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) { zj[q] -= v * au[q]; z[j + q] = zj[q]; a[ij + 1 + q] = au[q] + beta * zj[q]; }
                ij += 4;
            }
            for (; j < n; ++j) {
                ++ij;
                z[j] -= v * a[ij];
                a[ij] += beta * z[j];
            }
        }
        ++ij;
        t = tp;
    }
}

// Scalars of Kraft's SLSQPB body (m = 0) that survive between its reverse-communication calls.
struct WideSlsqp {
    double f, f0, t0, h3, alpha;
    int ireset, line;
};
enum : int { WQ_INIT = 0, WQ_FEVAL = 1, WQ_FGEVAL = -2, WQ_GRAD = -1 };

// One call of SLSQPB.  In: mode 0 (first call; f, g at x set), 1 / -2 (function [and gradient]
// evaluated at x), -1 (gradient evaluated).  Out: 1 / -2 (evaluate at x), -1 (line search done,
// gradient wanted) or a terminal mode (3 .. 9).  acc = 0 (NLopt does the convergence tests).
template <class WP>
__device__ inline int w_slsqpb(int n, WideSlsqp &st, WP ws, const double *xl, const double *xu, int mode) {
    const double alfmin = 0.1;
    const int n1 = n + 1, n2 = n1 * n / 2;
    WP x = ws + wide_ws::X, x0 = ws + wide_ws::X0, g = ws + wide_ws::G, s = ws + wide_ws::S, u = ws + wide_ws::U,
       v = ws + wide_ws::V, l = ws + wide_ws::L, lw = ws + wide_ws::LW;
    bool reset = false;
    if (mode == WQ_GRAD) {
        // label 260: BFGS update of the LDL' factors
        W_EACH(i, n) u[i] = g[i] - v[i];
        {  // v = L D L' s
            int k = -1;
            if constexpr (wp_coop<WP>::value) {  // (the rows are independent: one per lane)
                w_sync<WP>();
                for (int i = w_first<WP>(); i < n; i += w_step<WP>()) {
                    const int kd = i * n - (i * (i - 1)) / 2;  // index of l(i, i) in the packed factor
                    v[i] = s[i] + w_dot(n - i - 1, l + (kd + 1), 1, s + (i + 1), 1);
                }
                w_sync<WP>();
            } else
            for (int i = 0; i < n; ++i) {
                ++k;
                const double h = w_dot(n - i - 1, l + (k + 1), 1, s + (i + 1), 1);
                k += n - i - 1;
                v[i] = s[i] + h;
            }
            if constexpr (wp_coop<WP>::value) {
                W_EACH(i, n) v[i] = l[i * n - (i * (i - 1)) / 2] * v[i];
                w_sync<WP>();
                // (row i adds the products of the rows above it, which are still unchanged when it is formed going
                // down from the last row: every lane forms its sum from the old values, then all add)
                double hrow[(WIDE_MAX_DOF + 63) / 64];
                int r = 0;
                W_EACH(i, n) {
                    double h = 0.0;
                    int kk = i;
                    for (int j = 0; j < i; ++j) { h += l[kk] * v[j]; kk += n - (j + 1); }
                    hrow[r++] = h;
                }
                w_sync<WP>();
                r = 0;
                W_EACH(i, n) v[i] += hrow[r++];
                w_sync<WP>();
            } else {
                k = 0;
                for (int i = 0; i < n; ++i) { v[i] = l[k] * v[i]; k += n1 - (i + 1); }
                for (int i = n - 1; i >= 0; --i) {
                    double h = 0.0;
                    k = i;
                    for (int j = 0; j < i; ++j) { h += l[k] * v[j]; k += n - (j + 1); }
                    v[i] += h;
                }
            }
        }
        double h1 = w_dot_all(n, s, 1, u, 1);
        const double h2 = w_dot_all(n, s, 1, v, 1);
        const double h3 = h2 * 0.2;
        if (h1 < h3) {
            const double h4 = (h2 - h3) / (h2 - h1);
            h1 = h3;
            W_EACH(i, n) {
                double ui = u[i];
                ui *= h4;
                ui += (1.0 - h4) * v[i];
                u[i] = ui;
            }
            w_sync<WP>();
        }
        w_ldl_update(n, l, u, 1.0 / h1, lw);
        w_ldl_update(n, l, v, -1.0 / h2, lw);
    } else if (mode == WQ_INIT) {
        // label 100
        st.ireset = 0;
        W_EACH(i, n) s[i] = 0.0;
        w_sync<WP>();
        reset = true;
    } else {
        // label 220: function evaluated, L1 merit (m = 0: t = f)
        const double h1 = st.f - st.t0;
        if (__builtin_isfinite(h1)) {
            if (h1 <= st.h3 / 10.0 || st.line > 10) { st.h3 = 0.0; return WQ_GRAD; }  // label 240
            const double a = st.h3 / ((st.h3 - h1) * 2.0);
            st.alpha = (a > alfmin) ? a : alfmin;
        } else {
            const double a = st.alpha * 0.5;
            st.alpha = (a > alfmin) ? a : alfmin;
        }
        goto trial;
    }
    for (;;) {
        if (reset) {  // label 110: reset the BFGS matrix
            ++st.ireset;
            if (st.ireset > 5) return 8;  // label 255 with acc = 0
            W_EACH(i, n2) l[i] = 0.0;
            w_sync<WP>();
            W_EACH(i, n) l[i * n - (i * (i - 1)) / 2] = 1.0;
            w_sync<WP>();
        }
        // label 130: search direction
        W_EACH(i, n) { u[i] = xl[i] - x[i]; v[i] = xu[i] - x[i]; }
        w_sync<WP>();
        const int lmode = w_lsq_box(n, ws, l, g, u, v, s);
        if (lmode != 1) return lmode;
        W_EACH(i, n) { v[i] = g[i]; x0[i] = x[i]; }
        w_sync<WP>();
        st.f0 = st.f;
        const double gs = w_dot_all(n, g, 1, s, 1);
        st.t0 = st.f;
        st.h3 = gs;  // gs - h1 * h4 with h1 = 0 (acc = 0, no constraints)
        if (st.h3 >= 0.0) { reset = true; continue; }
        st.line = 0;
        st.alpha = 1.0;
        break;
    }
trial:
    // label 190: next trial point
    ++st.line;
    st.h3 = st.alpha * st.h3;
    W_EACH(i, n) {
        const double si = s[i] * st.alpha;
        s[i] = si;
        double xi = x0[i];
        xi += si;
        if (xi < xl[i]) xi = xl[i];  // NLopt (SGJ 2010): roundoff must not push x past the bounds
        else if (xi > xu[i]) xi = xu[i];
        x[i] = xi;
    }
    w_sync<WP>();
    return (st.line == 1) ? WQ_FGEVAL : WQ_FEVAL;  // NLopt: the first trial comes with its gradient
}

// nlopt_stop_x with xtol_rel = 0 and xtol_abs[i] = tol_dx (see stop_x in ik_solve.hpp).
template <class WP>
__device__ inline bool w_stop_x(const SolveParams &sp, int n, WP x, WP oldx) {
    if constexpr (wp_coop<WP>::value) {  // (an element per lane, the verdicts by ballot)
        bool differs = false, large = false;
        W_EACH(i, n) {
            differs = differs || !(x[i] == oldx[i]);
            large = large || (__builtin_fabs(x[i] - oldx[i]) >= sp.xtol_abs);
        }
        if (sp.stop_x_zero && !wave_any(differs)) return true;
        return !wave_any(large);
    }
    if (sp.stop_x_zero) {
        bool zero = true;
        for (int i = 0; i < n; ++i) zero = zero && (x[i] == oldx[i]);
        if (zero) return true;
    }
    for (int i = 0; i < n; ++i)
        if (__builtin_fabs(x[i] - oldx[i]) >= sp.xtol_abs) return false;
    return true;
}
__device__ inline bool w_relstop(double vold, double vnew, double abstol) {
    if (__builtin_isinf(vold)) return false;
    return __builtin_fabs(vnew - vold) < abstol;  // (reltol = 0: the other two terms cannot fire)
}

// One 64-lane wave solving restarts until the queue is empty: the driver of nlopt_slsqp() and the
// closure of lib.rs:301-391, one trip = (one evaluation, its bookkeeping, one SLSQPB call) per lane.
template <class WP>
__device__ inline void wide_solve_wave(const WideChainDev &ch, const EvalParams &ep, const SolveParams &sp,
                                       const uint32_t (&key)[8], const WorkQueue &wq, const WP ws) {
    const int n = ch.n_pos;
    const int lane = (int)(threadIdx.x & 63u);
    const WP x = ws + wide_ws::X, g = ws + wide_ws::G, xbest = ws + wide_ws::XBEST, xprev = ws + wide_ws::XPREV,
             x0 = ws + wide_ws::X0, tf = ws + wide_ws::TF, gn = ws + wide_ws::GN;
    WideSlsqp st{};
    double minf = __builtin_huge_val(), fprev = __builtin_huge_val();
    int mode = 0, nevals = 0;
    bool do_eval = true, want_grad = true;
    Pose target;
    target.t = V3{0, 0, 0};
    target.q = Q4{0, 0, 0, 1};
    unsigned long long item = 0, index = 0;
    unsigned tslot = 0;
    constexpr bool coop = wp_coop<WP>::value;  // the wave's 64 lanes share ONE restart (identical state in all of them)
    bool active = false, want = coop || lane < wq.lanes;

    for (;;) {
        // ---- refill: lanes without a restart pull the next work item ----------------------------
        if (wave_any(want)) {
            unsigned long long it;
            if constexpr (coop) {
                it = 0;
                if (lane == 0) it = atomicAdd(wq.next_item, 1ull);
                it = __shfl(it, 0, 64);
            } else {
                it = fetch_items(wq.next_item, want);
            }
            if (want) {
                want = false;
                if (it < wq.total_items) {
                    unsigned long long r;
                    if (wq.restart_major) { r = it / wq.n_targets; tslot = (unsigned)(it - r * wq.n_targets); }
                    else { tslot = (unsigned)(it / wq.n_restarts); r = it - (unsigned long long)tslot * wq.n_restarts; }
                    item = (unsigned long long)tslot * wq.n_restarts + r;  // output column
                    index = wq.restart_begin + r;
                    target = load_pose(wq.targets + (size_t)tslot * 7);
                    // lib.rs:366-370: restart 0 starts from the caller's seed
                    if (index == 0) {
                        const double *x0p = wq.x0 + (size_t)tslot * n;
                        for (int i = 0; i < n; ++i) x[i] = x0p[i];
                    } else {
                        wide_restart_seed(key, ch.lb, ch.scale, index, n, x);
                    }
                    for (int i = 0; i < n; ++i) { xbest[i] = x[i]; xprev[i] = x[i]; }
                    st = WideSlsqp{};
                    minf = __builtin_huge_val(); fprev = __builtin_huge_val();
                    mode = 0; nevals = 0;
                    do_eval = true; want_grad = true;  // NLopt: "eval once before calling slsqp the first time"
                    active = true;
                }
            }
        }
        if (!wave_any(active)) break;

        int32_t ret = 0;
        if (active) {
            // lib.rs:308: abandon when timed out or a lower-index restart succeeded
            bool stop = false;
            if (wq.first_success) {
                const unsigned long long fs = __hip_atomic_load(wq.first_success + tslot, __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT);
                stop = wq.find_any ? (fs != ~0ull) : (fs < index);
            }
            if (wq.deadline && (unsigned long long)wall_clock64() > wq.deadline) stop = true;
            // (cooperative form: the 64 lanes share ONE restart in LDS, but each read first_success / the clock
            // with its own vector load -- another wave's atomicMin can land between two lane groups of that
            // load.  Lane 0 decides for the wave, as the quad solver's leader does (ik_quad.hpp: quad_get(ret, 0)))
            if constexpr (wp_coop<WP>::value) stop = __builtin_amdgcn_readfirstlane((int)stop) != 0;
            if (stop) ret = RES_FORCED_STOP;
        }
        if (active && ret == 0) {
            if (do_eval) {
                st.f = want_grad ? wide_eval_fg(ch, ep, target, n, x, tf, g) : wide_eval_fg(ch, ep, target, n, x, tf, gn);
                ++nevals;
            }
            const int prev_mode = mode;
            if (st.f < minf) {  // NLopt: best point so far
                minf = st.f;
                W_EACH(i, n) xbest[i] = x[i];
                w_sync<WP>();
            }
            if (mode == WQ_GRAD) {  // a line search completed: only then are ftol / xtol tested
                if (!__builtin_isinf(fprev)) {
                    if (w_relstop(fprev, st.f, sp.ftol_abs)) ret = RES_FTOL_REACHED;
                    else if (w_stop_x(sp, n, x, xprev)) ret = RES_XTOL_REACHED;
                }
                fprev = st.f;
                W_EACH(i, n) xprev[i] = x[i];
                w_sync<WP>();
            }
            if (minf < sp.stopval) ret = RES_STOPVAL_REACHED;
            if (ret == 0 && nevals >= MAX_EVALS_CAP) ret = RES_ITER_CAP;
            if (ret == 0) {
                mode = w_slsqpb(n, st, ws, ch.lb, ch.ub, mode);
                switch (mode) {
                case WQ_GRAD:
                    do_eval = (prev_mode != WQ_FGEVAL);  // NLopt: that point was just evaluated with its gradient
                    want_grad = true;
                    break;
                case WQ_FGEVAL: do_eval = true; want_grad = true; break;
                case WQ_FEVAL: do_eval = true; want_grad = false; break;
                case 8:  // positive directional derivative: the relaxed test against (f0, x0)
                    ret = RES_ROUNDOFF_LIMITED;
                    if (w_relstop(st.f0, st.f, sp.ftol_abs)) ret = RES_FTOL_REACHED;
                    else if (w_stop_x(sp, n, x, x0)) ret = RES_XTOL_REACHED;
                    break;
                case 5: case 6: case 7: ret = RES_ROUNDOFF_LIMITED; break;
                default: ret = RES_FAILURE; break;  // 3, 4, 9
                }
            }
        }
        // ---- a restart ended: classify (lib.rs:376-379), publish, free the lane -------------------
        if (ret != 0) {
            const bool success = (sp.ok_stopval && ret == RES_STOPVAL_REACHED) || (sp.ok_ftol && ret == RES_FTOL_REACHED)
                                 || (sp.ok_xtol && ret == RES_XTOL_REACHED);
            const bool writer = !coop || lane == 0;
            if (wq.out_x && writer)
                for (int i = 0; i < n; ++i) wq.out_x[(size_t)i * wq.total_items + item] = xbest[i];
            if (wq.out_f && writer) wq.out_f[item] = minf;
            if (wq.out_status && writer) wq.out_status[item] = ret;
            if (wq.out_evals && writer) wq.out_evals[item] = nevals;
            // selection key (lib.rs:402-407): Quality = ||x - x0||_2, Speed = index
            double k = __builtin_huge_val();
            if (success) {
                if (wq.quality) {
                    const double *x0p = wq.x0 + (size_t)tslot * n;
                    double acc = 0.0;
                    for (int i = 0; i < n; ++i) { const double d = xbest[i] - x0p[i]; acc += d * d; }
                    k = __builtin_sqrt(acc);
                } else {
                    k = (double)index;
                    if (wq.first_success && writer) atomicMin(wq.first_success + tslot, index);
                }
            }
            if (wq.out_key && writer) wq.out_key[item] = k;
            active = false;
            want = true;
        }
    }
}

}  // namespace optik
