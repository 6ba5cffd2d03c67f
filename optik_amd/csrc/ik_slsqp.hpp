// ik_slsqp.hpp -- per-lane building blocks of SLSQP (box bounds only).
//
// What is restated: the inner loop the reference delegates to NLopt
// (/root/reference/crates/optik/src/lib.rs:302-356, 372; NLopt SLSQP = Kraft's
// SLSQPB/LSQ/LSEI/LSI/LDP/NNLS/H12/LDL with NLopt's stopping rules), for
// m = meq = 0 and finite bounds.  Here: the pieces one lane computes on one restart's state in
// registers with fully unrolled, statically indexed loops -- LSQ's set-up (E = D^1/2 L',
// f = -E^-T g), Kraft's transformation of the 2n bound rows to a least-distance problem
// (G E^-1 = +-E^-1 because G = [I; -I]; zeros are skipped, which leaves every non-zero
// bit-identical), the back-substitution, the Fletcher-Powell LDL' update and the damped BFGS
// update.  Their user: the lane-per-restart form (ik_lane64.hpp); the quad solver (ik_quad.hpp) spreads the
// same arithmetic over four lanes.
// The NNLS of the dual problem -- Lawson-Hanson's active-set iteration, whose column choices are
// data dependent -- is ik_nnls_quad.hpp (matrix in LDS).
//
// Operation order equals oracle/optik_oracle.c everywhere a non-zero flows, so
// kernel results are bit-identical to the CPU oracle (-ffp-contract=off).
#pragma once

#include "ik_eval.hpp"

namespace optik {

constexpr double EPMACH = 2.220446049250313e-16;

OPTIK_DEV bool wave_any(bool p) { return __ballot(p) != 0ull; }

// 4-bit packed permutation (columns 1..16), positions 1-based.
struct PackedIndex {
    uint64_t v;
    OPTIK_DEV int get(int pos) const { return (int)((v >> (4 * (pos - 1))) & 15ull) + 1; }
    OPTIK_DEV void set(int pos, int col) {
        const int sh = 4 * (pos - 1);
        v = (v & ~(15ull << sh)) | ((uint64_t)(col - 1) << sh);
    }
};

// BLAS drotg as NLopt's slsqp.c restates it.
OPTIK_DEV void rotg(double &da, double &db, double &c, double &s) {
    const double roe = (__builtin_fabs(da) > __builtin_fabs(db)) ? da : db;
    const double scale = __builtin_fabs(da) + __builtin_fabs(db);
    double r, z;
    if (scale == 0.0) {
        c = 1.0; s = 0.0; r = 0.0;
    } else {
        const double a = da / scale, b = db / scale;
        r = scale * __builtin_sqrt(a * a + b * b);
        if (roe < 0.0) r = -r;
        c = da / r;
        s = db / r;
    }
    z = s;
    if (__builtin_fabs(c) > 0.0 && __builtin_fabs(c) <= s) z = 1.0 / c;
    da = r;
    db = z;
}

// Element `idx` (1-based, per-lane) of a register array: a select chain, no scratch.
template <int M>
OPTIK_DEV double pick(const double (&a)[M], int idx) {
    double v = a[0];
#pragma unroll
    for (int i = 1; i < M; ++i) v = (idx == i + 1) ? a[i] : v;
    return v;
}

// index of element (row j, column i), j >= i, in the column-packed LDL' array
template <int N>
OPTIK_DEV constexpr int lidx(int i, int j) { return i * N - (i * (i - 1)) / 2 + (j - i); }

#ifndef OPTIK_ROWS_GROUP
#define OPTIK_ROWS_GROUP 1
#endif

// ---- Kraft LSQ for m = 0 and finite bounds:  min ||E s - f||, lo <= s <= hi ------
// In pieces, so that their callers run the LDS-free parts per lane and send only the restarts whose step hits a
// bound through the NNLS:
//   lsq_factor        E = D^1/2 L', f, Kraft's Householder pass
//   lsq_bound_rows    G E^-1 = +-E^-1 and h: the rows of the dual problem (handed to a sink: LDS record, HBM slot)
//   (NNLS, then the LDP tail: ik_lane64.hpp)
//   lsq_finish        s = E^-1 (y + f), clipped

// E = D^1/2 L', f = -E^-T g, then Kraft's LSI Householder pass.  Returns 1, or 5 when E is
// numerically singular (Kraft LSI mode 5).
template <int N>
OPTIK_DEV int lsq_factor(const double (&l)[N * (N + 1) / 2], const double (&g)[N], double (&E)[N][N],
                         double (&f)[N]) {
    // recover E and f from L and g
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double diag = __builtin_sqrt(l[lidx<N>(i, i)]);
#pragma unroll
        for (int j = i + 1; j < N; ++j) E[i][j] = l[lidx<N>(i, j)] * diag;
        E[i][i] = diag;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) acc += E[k][i] * f[k];
        f[i] = (g[i] - acc) / diag;
        OPTIK_SCHED_FENCE_SLSQP();
    }
#pragma unroll
    for (int i = 0; i < N; ++i) f[i] = -f[i];

    // LSI: "QR" of the already-triangular E (a Householder reflection on a column
    // whose sub-diagonal is zero: flips the sign of row i up to roundoff)
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
        const double p = E[i][i];
        double cl = __builtin_fabs(p);
        if (!(cl <= 0.0)) {
            const double clinv = 1.0 / cl;
            const double d = p * clinv;
            const double sm0 = d * d;
            cl *= __builtin_sqrt(sm0);
            if (p > 0.0) cl = -cl;
            const double up = p - cl;
            E[i][i] = cl;
            double b = up * cl;
            if (!(b >= 0.0)) {
                b = 1.0 / b;
#pragma unroll
                for (int j = i + 1; j < N; ++j) {
                    double sm = E[i][j] * up;
                    if (sm != 0.0) { sm *= b; E[i][j] += sm * up; }
                }
                double sm = f[i] * up;
                if (sm != 0.0) { sm *= b; f[i] += sm * up; }
            }
        }
        OPTIK_SCHED_FENCE_SLSQP();
    }
    bool singular = false;
#pragma unroll
    for (int j = 0; j < N; ++j) singular = singular || !(__builtin_fabs(E[j][j]) >= EPMACH);
    return singular ? 5 : 1;
}

// Transforms G = [I; -I] and h = [lo; -hi]: row i of E^-1 (entries j >= i) and the two
// bound rows it gives, handed to sink(i, row, h_lo, h_hi) one row at a time -- a caller
// that streams the rows out never holds E^-1.  Returns whether some h_j > 0, i.e. the
// unconstrained step leaves the box.  (NNLS's first dual check is w_j = h_j (b = e_{n+1});
// when no h_j is positive it returns at once with zero multipliers (y = 0, fac = 1, step
// 0): skipping it then is bit-identical to running it, every product being an exact zero.)
template <int N, class RowSink>
OPTIK_DEV bool lsq_bound_rows(const double (&E)[N][N], const double (&f)[N], const double (&lo)[N],
                              const double (&hi)[N], RowSink &&sink) {
    bool need = false;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double row[N];
#pragma unroll
        for (int j = 0; j < N; ++j) row[j] = 0.0;
#pragma unroll
        for (int j = i; j < N; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int k = i; k < j; ++k) acc += row[k] * E[k][j];
            row[j] = (((j == i) ? 1.0 : 0.0) - acc) / E[j][j];
        }
        double acc = 0.0;
#pragma unroll
        for (int j = i; j < N; ++j) acc += row[j] * f[j];
        const double h_lo = lo[i] - acc;
        const double h_hi = (-hi[i]) - (-acc);
        need = need || (h_lo > 0.0) || (h_hi > 0.0);
        sink(i, row, h_lo, h_hi);
        // rows are independent recurrences (each a chain of divisions): letting the scheduler
        // interleave OPTIK_ROWS_GROUP of them hides the division latency at 2 waves per SIMD
        if (i % OPTIK_ROWS_GROUP == OPTIK_ROWS_GROUP - 1) OPTIK_SCHED_FENCE_SLSQP();
    }
    return need;
}

// s (transformed space, zero when NNLS was skipped) -> solution of the original
// problem s = E^-1 (s + f), clipped into [lo, hi] (NLopt).
template <int N>
OPTIK_DEV void lsq_finish(const double (&E)[N][N], const double (&f)[N], const double (&lo)[N],
                          const double (&hi)[N], double (&s)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] += f[i];
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double acc = 0.0;
#pragma unroll
        for (int j = i + 1; j < N; ++j) acc += E[i][j] * s[j];
        s[i] = (s[i] - acc) / E[i][i];
        OPTIK_SCHED_FENCE_SLSQP();
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (s[i] < lo[i]) s[i] = lo[i];
        else if (s[i] > hi[i]) s[i] = hi[i];
    }
}

// Fletcher-Powell composite-t update  LDL' += sigma z z'  on the packed factor.
template <int N>
OPTIK_DEV void ldl_update(double (&a)[N * (N + 1) / 2], double (&z)[N], double sigma) {
    if (sigma == 0.0) return;
    double w[N];
    double t = 1.0 / sigma;
    if (sigma < 0.0) {
#pragma unroll
        for (int i = 0; i < N; ++i) w[i] = z[i];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double v = w[i];
            t += v * v / a[lidx<N>(i, i)];
#pragma unroll
            for (int j = i + 1; j < N; ++j) w[j] -= v * a[lidx<N>(i, j)];
            OPTIK_SCHED_FENCE_SLSQP();
        }
        if (t >= 0.0) t = EPMACH / sigma;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = N - 1 - i;
            const double u = w[j];
            w[j] = t;
            t -= u * u / a[lidx<N>(j, j)];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double v = z[i];
        const double delta = v / a[lidx<N>(i, i)];
        const double tp = (sigma < 0.0) ? w[i] : t + delta * v;
        const double alpha = tp / t;
        a[lidx<N>(i, i)] = alpha * a[lidx<N>(i, i)];
        if (i < N - 1) {
            const double beta = delta / tp;
            if (alpha > 4.0) {
                const double gamma = t / tp;
#pragma unroll
                for (int j = i + 1; j < N; ++j) {
                    const double u = a[lidx<N>(i, j)];
                    a[lidx<N>(i, j)] = gamma * u + beta * z[j];
                    z[j] -= v * u;
                }
            } else {
#pragma unroll
                for (int j = i + 1; j < N; ++j) {
                    z[j] -= v * a[lidx<N>(i, j)];
                    a[lidx<N>(i, j)] += beta * z[j];
                }
            }
            t = tp;
        }
        OPTIK_SCHED_FENCE_SLSQP();
    }
}

// BFGS update of the LDL' factors (Kraft SLSQPB label 260) with Powell damping.
// u_in = g_new - g_old on entry (destroyed); s = accepted step.
template <int N>
OPTIK_DEV void bfgs_update(double (&l)[N * (N + 1) / 2], const double (&s)[N], double (&u)[N]) {
    double v[N];
    // v = L D L' s
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double h = 0.0;
#pragma unroll
        for (int j = i + 1; j < N; ++j) h += l[lidx<N>(i, j)] * s[j];
        v[i] = s[i] + h;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = l[lidx<N>(i, i)] * v[i];
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double h = 0.0;
#pragma unroll
        for (int j = 0; j < i; ++j) h += l[lidx<N>(j, i)] * v[j];
        v[i] += h;
    }
    double h1 = 0.0, h2 = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) h1 += s[i] * u[i];
#pragma unroll
    for (int i = 0; i < N; ++i) h2 += s[i] * v[i];
    const double h3 = h2 * 0.2;
    if (h1 < h3) {
        const double h4 = (h2 - h3) / (h2 - h1);
        h1 = h3;
#pragma unroll
        for (int i = 0; i < N; ++i) u[i] *= h4;
#pragma unroll
        for (int i = 0; i < N; ++i) u[i] += (1.0 - h4) * v[i];
    }
    OPTIK_SCHED_FENCE_SLSQP();
    ldl_update<N>(l, u, 1.0 / h1);
    OPTIK_SCHED_FENCE_SLSQP();
    ldl_update<N>(l, v, -1.0 / h2);
    OPTIK_SCHED_FENCE_SLSQP();
}

}  // namespace optik
