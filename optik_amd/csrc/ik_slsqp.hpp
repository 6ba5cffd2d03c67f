// ik_slsqp.hpp -- per-lane SLSQP (box bounds only) for the batched restart kernel.
//
// What is restated: the inner loop the reference delegates to NLopt
// (/root/reference/crates/optik/src/lib.rs:302-356, 372; NLopt SLSQP = Kraft's
// SLSQPB/LSQ/LSEI/LSI/LDP/NNLS/H12/LDL with NLopt's stopping rules), for
// m = meq = 0 and finite bounds.  How it is laid out for CDNA4:
//
//   * one restart per lane; x, x0, g, s, the packed LDL' factor (n(n+1)/2) and
//     all scalars of the reverse-communication state stay in VGPRs with fully
//     unrolled, statically indexed loops;
//   * LSQ's set-up (E = D^1/2 L', f = -E^-T g), Kraft's transformation of the
//     2n bound rows to a least-distance problem (G E^-1 = +-E^-1 because
//     G = [I; -I]; zeros are skipped, which leaves every non-zero bit-identical)
//     and the back-substitution are O(n^3/6) register code;
//   * only NNLS -- Lawson-Hanson's active-set iteration, whose column choices
//     are data dependent per lane -- needs indexable storage: its (n+1) x 2n
//     matrix, b, z, x and w live in LDS as [slot][64 lanes] doubles, so lane l
//     always touches banks {2l, 2l+1} whatever slot it indexes (conflict-free
//     under divergent indices).  The permutation vector is 4-bit packed in one
//     64-bit VGPR pair.
//
// Operation order equals oracle/optik_oracle.c everywhere a non-zero flows, so
// kernel results are bit-identical to the CPU oracle (-ffp-contract=off).
#pragma once

#include "ik_eval.hpp"

namespace optik {

constexpr double EPMACH = 2.220446049250313e-16;

OPTIK_DEV bool wave_any(bool p) { return __ballot(p) != 0ull; }

// LDS slots per lane for the NNLS workspace of an n-DoF problem.
template <int N>
struct NnlsLayout {
    static constexpr int R = N + 1;   // rows of the dual problem
    static constexpr int M = 2 * N;   // columns = bound rows
    static constexpr int A0 = 0;
    static constexpr int B0 = A0 + R * M;
    static constexpr int Z0 = B0 + R;
    static constexpr int X0 = Z0 + R;
    static constexpr int W0 = X0 + M;
    static constexpr int SLOTS = W0 + M;
};

// View of one lane's NNLS workspace: slot k of this lane is base[k * 64].
template <int N>
struct NnlsWs {
    using L = NnlsLayout<N>;
    double *base;
    // 1-based accessors, as in Lawson-Hanson
    OPTIK_DEV double &A(int i, int j) const { return base[(L::A0 + (j - 1) * L::R + (i - 1)) * 64]; }
    OPTIK_DEV double &b(int i) const { return base[(L::B0 + i - 1) * 64]; }
    OPTIK_DEV double &z(int i) const { return base[(L::Z0 + i - 1) * 64]; }
    OPTIK_DEV double &x(int j) const { return base[(L::X0 + j - 1) * 64]; }
    OPTIK_DEV double &w(int j) const { return base[(L::W0 + j - 1) * 64]; }
};

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

// Lawson-Hanson NNLS on the (N+1) x 2N dual problem held in LDS.
// Returns mode (1 ok, 3 iteration count exceeded); multipliers in ws.x().
//
// Same decisions and the same arithmetic, in the same order, as the textbook loop
// nest (oracle/optik_oracle.c:nnls); what differs is the shape given to the GPU:
// row loops are unrolled over the m = n+1 rows with per-lane predicates, so a
// column is fetched with one address and m immediate offsets (m LDS reads in flight
// instead of a dependent read per element), the Householder vector is held in
// registers while it is applied, and loops over "the columns still in set Z" run
// over all 2n columns under a bit mask (their order does not matter).
template <int N>
OPTIK_DEV int nnls(const NnlsWs<N> &ws, double &rnorm) {
    using L = NnlsLayout<N>;
    constexpr int m = N + 1, n = 2 * N;
    static_assert(m <= 16 && n <= 16, "row vectors are dvec8 / dvec16, the permutation is 16 nibbles");
    typedef typename RowVecOf<(m <= 8)>::type rowvec;
    const double factor = 0.01;
    int mode = 1, iter = 0;
    const int itmax = 3 * n;
    PackedIndex indx;
    indx.v = 0xFEDCBA9876543210ull;  // indx[pos] = pos
    unsigned zmask = (1u << n) - 1u;  // bit (col-1) set: column is in set Z
    int iz1 = 1, nsetp = 0, npp1 = 1;
    const int iz2 = n;
    int izmax = 0, j = 0, jj = 0;
    double up = 0.0;
    double *const Abase = ws.base + L::A0 * 64;
    auto col = [&](int c) -> double * { return Abase + (c - 1) * (L::R * 64); };  // row r at [(r-1)*64]
#pragma unroll
    for (int i = 1; i <= n; ++i) ws.x(i) = 0.0;

    for (;;) {  // step two: dual variables w = A'(b - Ax) of the columns in Z
        if (iz1 > iz2 || nsetp >= m) break;
        {
            double bv[m];
#pragma unroll
            for (int r = 0; r < m; ++r) bv[r] = ws.b(r + 1);
#pragma unroll
            for (int c = 1; c <= n; ++c) {
                const double *cp = col(c);
                double sdot = 0.0;
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    const double a = cp[(r - 1) * 64];
                    if (r >= npp1) sdot += a * bv[r - 1];
                }
                if (zmask & (1u << (c - 1))) ws.w(c) = sdot;
            }
        }
        bool found = false;
        for (;;) {  // step three / four: most positive dual, in position order
            double wmax = 0.0;
            for (int iz = iz1; iz <= iz2; ++iz) {
                j = indx.get(iz);
                const double wj = ws.w(j);
                if (wj <= wmax) continue;
                wmax = wj;
                izmax = iz;
            }
            if (wmax <= 0.0) break;
            const int iz = izmax;
            j = indx.get(iz);
            // step five: does column j enter the positive set?  (H12 construction on
            // column j, pivot row npp1, rows npp1+1..m)
            double *const cj = col(j);
            rowvec u = 0.0;
#pragma unroll
            for (int r = 0; r < m; ++r) u[r] = cj[r * 64];
            const double asave = vpick(u, npp1);
            const bool h12_live = npp1 < m;  // "lpivot >= l1 || l1 > m" returns early
            double ulp = asave;              // U(lpivot) after the construction
            bool constructed = false;
            if (h12_live) {
                double cl = __builtin_fabs(asave);
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    const double sm = __builtin_fabs(u[r - 1]);
                    if (r > npp1 && sm > cl) cl = sm;
                }
                if (!(cl <= 0.0)) {
                    const double clinv = 1.0 / cl;
                    double d = asave * clinv;
                    double sm = d * d;
#pragma unroll
                    for (int r = 1; r <= m; ++r) {
                        d = u[r - 1] * clinv;
                        if (r > npp1) sm += d * d;
                    }
                    cl *= __builtin_sqrt(sm);
                    if (asave > 0.0) cl = -cl;
                    up = asave - cl;
                    ulp = cl;
                    constructed = true;
                }
            }
            // unorm = ||A(1..nsetp, j)|| (NLopt's scaled dnrm2)
            double unorm = 0.0;
            {
                double xmax = 0.0;
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    const double a = __builtin_fabs(u[r - 1]);
                    if (r <= nsetp && a > xmax) xmax = a;
                }
                if (xmax != 0.0) {
                    const double scale = 1.0 / xmax;
                    double sum = 0.0;
#pragma unroll
                    for (int r = 1; r <= m; ++r) {
                        const double xs = scale * u[r - 1];
                        if (r <= nsetp) sum += xs * xs;
                    }
                    unorm = xmax * __builtin_sqrt(sum);
                }
            }
            const double t = factor * __builtin_fabs(ulp);
            const double d1 = unorm + t;
            rowvec zz = 0.0;
            // b factor of the H12 application (same for every vector it is applied to)
            double hb = 0.0;
            bool apply_live = false;
            if (h12_live && !(__builtin_fabs(ulp) <= 0.0)) {
                hb = up * ulp;
                if (!(hb >= 0.0)) { hb = 1.0 / hb; apply_live = true; }
            }
            if (d1 - unorm > 0.0) {
#pragma unroll
                for (int r = 0; r < m; ++r) zz[r] = ws.b(r + 1);
                if (apply_live) {
                    double sm = vpick(zz, npp1) * up;
#pragma unroll
                    for (int r = 1; r <= m; ++r)
                        if (r > npp1) sm += zz[r - 1] * u[r - 1];
                    if (sm != 0.0) {
                        sm *= hb;
#pragma unroll
                        for (int r = 1; r <= m; ++r) {
                            if (r == npp1) zz[r - 1] += sm * up;
                            else if (r > npp1) zz[r - 1] += sm * u[r - 1];
                        }
                    }
                }
                if (vpick(zz, npp1) / ulp > 0.0) found = true;
            }
            if (found) {
                // b := Q b; column j joins set P at position iz1
#pragma unroll
                for (int r = 0; r < m; ++r) ws.b(r + 1) = zz[r];
                indx.set(iz, indx.get(iz1));
                indx.set(iz1, j);
                ++iz1;
                nsetp = npp1;
                ++npp1;
                zmask &= ~(1u << (j - 1));
                // apply the transformation to the columns left in Z (pivot nsetp, rows npp1..m)
                if (apply_live) {
#pragma unroll
                    for (int c = 1; c <= n; ++c) {
                        if (!(zmask & (1u << (c - 1)))) continue;
                        double *cp = col(c);
                        rowvec cv = 0.0;
#pragma unroll
                        for (int r = 0; r < m; ++r) cv[r] = cp[r * 64];
                        double sm = vpick(cv, nsetp) * up;
#pragma unroll
                        for (int r = 1; r <= m; ++r)
                            if (r >= npp1) sm += cv[r - 1] * u[r - 1];
                        if (sm != 0.0) {
                            sm *= hb;
#pragma unroll
                            for (int r = 1; r <= m; ++r) {
                                if (r == nsetp) cp[(r - 1) * 64] = cv[r - 1] + sm * up;
                                else if (r >= npp1) cp[(r - 1) * 64] = cv[r - 1] + sm * u[r - 1];
                            }
                        }
                    }
                }
                // column j itself: pivot value, zeros below
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    if (r == nsetp) cj[(r - 1) * 64] = ulp;
                    else if (r >= npp1) cj[(r - 1) * 64] = 0.0;
                }
                ws.w(j) = 0.0;
                break;
            }
            // rejected: A(npp1, j) keeps its value (the construction is discarded)
            (void)constructed;
            ws.w(j) = 0.0;
        }
        if (!found) break;

        for (;;) {  // step six: solve the triangular system R z = Q'b on set P
            rowvec zz = 0.0;
#pragma unroll
            for (int r = 0; r < m; ++r) zz[r] = ws.b(r + 1);
            for (int ip = nsetp; ip >= 1; --ip) {
                jj = indx.get(ip);
                const double *cp = col(jj);
                rowvec cv = 0.0;
#pragma unroll
                for (int r = 0; r < m; ++r) cv[r] = cp[r * 64];
                const double zi = vpick(zz, ip) / vpick(cv, ip);
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    if (r == ip) zz[r - 1] = zi;
                    else if (r < ip) zz[r - 1] -= zi * cv[r - 1];
                }
            }
            ++iter;
            if (iter > itmax) { mode = 3; goto done; }
            // steps seven..ten: step length towards z that keeps x >= 0
            double alpha = 1.0;
            jj = 0;
#pragma unroll
            for (int ip = 1; ip <= m; ++ip) {
                if (ip > nsetp) continue;
                const double zi = zz[ip - 1];
                if (zi > 0.0) continue;
                const int l = indx.get(ip);
                const double xl = ws.x(l);
                const double t = -xl / (zi - xl);
                if (alpha < t) continue;
                alpha = t;
                jj = ip;
            }
#pragma unroll
            for (int ip = 1; ip <= m; ++ip) {
                if (ip > nsetp) continue;
                const int l = indx.get(ip);
                ws.x(l) = (1.0 - alpha) * ws.x(l) + alpha * zz[ip - 1];
            }
            if (jj == 0) break;  // back to step two
            // step eleven: move coefficient i from set P to set Z
            int i = indx.get(jj);
            for (;;) {
                ws.x(i) = 0.0;
                zmask |= 1u << (i - 1);
                ++jj;
                for (j = jj; j <= nsetp; ++j) {
                    const int ii = indx.get(j);
                    indx.set(j - 1, ii);
                    double c, s;
                    double *const r0 = Abase + (j - 2) * 64;  // row j-1 of column 1
                    double *const r1 = Abase + (j - 1) * 64;  // row j
                    double a0 = r0[(ii - 1) * (L::R * 64)], a1 = r1[(ii - 1) * (L::R * 64)];
                    rotg(a0, a1, c, s);
                    const double t = a0;
                    // rows j-1, j of every column (column ii takes the (r, z) pair first)
                    double xa[n], ya[n];
#pragma unroll
                    for (int cc = 0; cc < n; ++cc) {
                        xa[cc] = r0[cc * (L::R * 64)];
                        ya[cc] = r1[cc * (L::R * 64)];
                    }
#pragma unroll
                    for (int cc = 0; cc < n; ++cc) {
                        const bool is_ii = (cc == ii - 1);
                        const double xi = is_ii ? a0 : xa[cc], yi = is_ii ? a1 : ya[cc];
                        const double nx = c * xi + s * yi;
                        const double ny = c * yi - s * xi;
                        r0[cc * (L::R * 64)] = is_ii ? t : nx;
                        r1[cc * (L::R * 64)] = is_ii ? 0.0 : ny;
                    }
                    const double bx = ws.b(j - 1), by = ws.b(j);
                    ws.b(j - 1) = c * bx + s * by;
                    ws.b(j) = c * by - s * bx;
                }
                npp1 = nsetp;
                --nsetp;
                --iz1;
                indx.set(iz1, i);
                if (nsetp <= 0) { mode = 3; goto done; }
                bool again = false;
                for (jj = 1; jj <= nsetp; ++jj) {
                    i = indx.get(jj);
                    if (ws.x(i) <= 0.0) { again = true; break; }
                }
                if (!again) break;
            }
        }
    }
done: {
        // rnorm = ||b(npp1..m)||
        const int k = (npp1 < m) ? npp1 : m;
        const int cnt = m - nsetp;
        double xmax = 0.0;
#pragma unroll
        for (int r = 1; r <= m; ++r) {
            const double a = __builtin_fabs(ws.b(r));
            if (r >= k && r < k + cnt && a > xmax) xmax = a;
        }
        rnorm = 0.0;
        if (xmax != 0.0) {
            const double scale = 1.0 / xmax;
            double sum = 0.0;
#pragma unroll
            for (int r = 1; r <= m; ++r) {
                const double xs = scale * ws.b(r);
                if (r >= k && r < k + cnt) sum += xs * xs;
            }
            rnorm = xmax * __builtin_sqrt(sum);
        }
    }
    return mode;
}

// index of element (row j, column i), j >= i, in the column-packed LDL' array
template <int N>
OPTIK_DEV constexpr int lidx(int i, int j) { return i * N - (i * (i - 1)) / 2 + (j - i); }

#ifndef OPTIK_ROWS_GROUP
#define OPTIK_ROWS_GROUP 1
#endif

// ---- Kraft LSQ for m = 0 and finite bounds:  min ||E s - f||, lo <= s <= hi ------
// Split in three so the streaming engine can run the LDS-free parts at high
// occupancy and only the lanes whose step hits a bound through NNLS:
//   lsq_prepare   E = D^1/2 L', f, Kraft's Householder pass, G E^-1 = +-E^-1, h
//   lsq_dual      LDP/NNLS on the dual (LDS)  -> the step in the transformed space
//   lsq_finish    s = E^-1 (y + f), clipped
// lsq_box chains them (the single-kernel solver).

template <int N>
struct LsqPrep {
    double E[N][N];   // upper triangular; [i][j] used for j >= i
    double f[N];
    double Gi[N][N];  // row i of E^-1, entries j >= i
    double h[2 * N];  // transformed bound rows
    bool need_nnls;   // some h_j > 0: the unconstrained step leaves the box
};

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
        OPTIK_SCHED_FENCE();
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
        OPTIK_SCHED_FENCE();
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
        if (i % OPTIK_ROWS_GROUP == OPTIK_ROWS_GROUP - 1) OPTIK_SCHED_FENCE();
    }
    return need;
}

// Returns 1, or 5 when E is numerically singular (Kraft LSI mode 5).
template <int N>
OPTIK_DEV int lsq_prepare(const double (&l)[N * (N + 1) / 2], const double (&g)[N], const double (&lo)[N],
                          const double (&hi)[N], LsqPrep<N> &P) {
    P.need_nnls = false;
    if (lsq_factor<N>(l, g, P.E, P.f) != 1) return 5;
    P.need_nnls = lsq_bound_rows<N>(P.E, P.f, lo, hi, [&](int i, const double (&row)[N], double h_lo, double h_hi) {
#pragma unroll
        for (int j = 0; j < N; ++j) P.Gi[i][j] = row[j];
        P.h[i] = h_lo;
        P.h[N + i] = h_hi;
    });
    return 1;
}

// LDP on the dual problem (LDS).  Writes the transformed-space step into s; mode 1 ok.
template <int N>
OPTIK_DEV int lsq_dual(const NnlsWs<N> &ws, const LsqPrep<N> &P, double (&s)[N],
                       unsigned long long &nnls_cycles) {
    constexpr int M = 2 * N;
#pragma unroll
    for (int c = 0; c < N; ++c) {
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const double v = (r >= c) ? P.Gi[c][r] : 0.0;
            ws.A(r + 1, c + 1) = v;
            ws.A(r + 1, N + c + 1) = (r >= c) ? -v : 0.0;
        }
        ws.A(N + 1, c + 1) = P.h[c];
        ws.A(N + 1, N + c + 1) = P.h[N + c];
        OPTIK_SCHED_FENCE();
    }
#pragma unroll
    for (int r = 1; r <= N; ++r) ws.b(r) = 0.0;
    ws.b(N + 1) = 1.0;
    double rnorm;
    OPTIK_SCHED_FENCE();
#ifdef OPTIK_PROFILE
    const unsigned long long t_nnls = __builtin_readcyclecounter();
#endif
    int mode = nnls<N>(ws, rnorm);
#ifdef OPTIK_PROFILE
    nnls_cycles += __builtin_readcyclecounter() - t_nnls;
#else
    (void)nnls_cycles;
#endif
    OPTIK_SCHED_FENCE();
    if (mode == 1 && rnorm <= 0.0) mode = 4;
    if (mode != 1) return mode;
    double y[M];
#pragma unroll
    for (int r = 0; r < M; ++r) y[r] = ws.x(r + 1);
    double hy = 0.0;
#pragma unroll
    for (int r = 0; r < M; ++r) hy += P.h[r] * y[r];
    double fac = 1.0 - hy;
    const double d1 = 1.0 + fac;
    if (d1 - 1.0 <= 0.0) return 4;
    fac = 1.0 / fac;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r <= j; ++r) acc += P.Gi[r][j] * y[r];
#pragma unroll
        for (int r = 0; r <= j; ++r) acc += (-P.Gi[r][j]) * y[N + r];
        s[j] = fac * acc;
        OPTIK_SCHED_FENCE();
    }
    return 1;
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
        OPTIK_SCHED_FENCE();
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (s[i] < lo[i]) s[i] = lo[i];
        else if (s[i] > hi[i]) s[i] = hi[i];
    }
}

template <int N>
OPTIK_DEV void lsq_finish(const LsqPrep<N> &P, const double (&lo)[N], const double (&hi)[N], double (&s)[N]) {
    lsq_finish<N>(P.E, P.f, lo, hi, s);
}

// The whole direction sub-problem.  Returns the LSQ mode (1 ok).
template <int N>
OPTIK_DEV int lsq_box(const NnlsWs<N> &ws, const double (&l)[N * (N + 1) / 2], const double (&g)[N],
                      const double (&lo)[N], const double (&hi)[N], double (&s)[N],
                      unsigned long long &nnls_cycles) {
    LsqPrep<N> P;
    int mode = lsq_prepare<N>(l, g, lo, hi, P);
    if (mode != 1) return mode;
    if (P.need_nnls) {
        mode = lsq_dual<N>(ws, P, s, nnls_cycles);
        if (mode != 1) return mode;
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) s[j] = 0.0;
    }
    lsq_finish<N>(P, lo, hi, s);
    return 1;
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
            OPTIK_SCHED_FENCE();
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
        OPTIK_SCHED_FENCE();
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
    OPTIK_SCHED_FENCE();
    ldl_update<N>(l, u, 1.0 / h1);
    OPTIK_SCHED_FENCE();
    ldl_update<N>(l, v, -1.0 / h2);
    OPTIK_SCHED_FENCE();
}

}  // namespace optik
