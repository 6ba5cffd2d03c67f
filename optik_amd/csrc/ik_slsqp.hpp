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

// NLopt's dnrm2 on LDS rows i0..i0+cnt-1 of a column (stride one slot).
template <typename F>
OPTIK_DEV double nrm2_by(int cnt, F at) {
    double xmax = 0.0;
    for (int i = 0; i < cnt; ++i) { const double a = __builtin_fabs(at(i)); if (a > xmax) xmax = a; }
    if (xmax == 0.0) return 0.0;
    const double scale = 1.0 / xmax;
    double sum = 0.0;
    for (int i = 0; i < cnt; ++i) { const double xs = scale * at(i); sum += xs * xs; }
    return xmax * __builtin_sqrt(sum);
}

// Lawson-Hanson H12, construction phase, on column j of A (pivot lp, rows l1..m).
template <int N>
OPTIK_DEV bool h12_construct(const NnlsWs<N> &ws, int j, int lp, int l1, int m, double &up) {
    if (0 >= lp || lp >= l1 || l1 > m) return false;
    double cl = __builtin_fabs(ws.A(lp, j));
    for (int r = l1; r <= m; ++r) { const double sm = __builtin_fabs(ws.A(r, j)); if (sm > cl) cl = sm; }
    if (cl <= 0.0) return false;
    const double clinv = 1.0 / cl;
    double d = ws.A(lp, j) * clinv;
    double sm = d * d;
    for (int r = l1; r <= m; ++r) { d = ws.A(r, j) * clinv; sm += d * d; }
    cl *= __builtin_sqrt(sm);
    if (ws.A(lp, j) > 0.0) cl = -cl;
    up = ws.A(lp, j) - cl;
    ws.A(lp, j) = cl;
    return true;
}

// H12 application phase of the transformation stored in column j to a vector
// accessed through `c(r)` (r = row, 1-based).
template <int N, typename C>
OPTIK_DEV void h12_apply(const NnlsWs<N> &ws, int j, int lp, int l1, int m, double up, C c) {
    if (0 >= lp || lp >= l1 || l1 > m) return;
    const double cl = __builtin_fabs(ws.A(lp, j));
    if (cl <= 0.0) return;
    double b = up * ws.A(lp, j);
    if (b >= 0.0) return;
    b = 1.0 / b;
    double sm = c(lp) * up;
    for (int r = l1; r <= m; ++r) sm += c(r) * ws.A(r, j);
    if (sm == 0.0) return;
    sm *= b;
    c(lp) += sm * up;
    for (int r = l1; r <= m; ++r) c(r) += sm * ws.A(r, j);
}

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

// Lawson-Hanson NNLS on the (N+1) x 2N dual problem held in LDS.
// Returns mode (1 ok, 3 iteration count exceeded); multipliers in ws.x().
template <int N>
OPTIK_DEV int nnls(const NnlsWs<N> &ws, double &rnorm) {
    constexpr int m = N + 1, n = 2 * N;
    const double factor = 0.01;
    int mode = 1, iter = 0;
    const int itmax = 3 * n;
    PackedIndex indx;
    indx.v = 0xFEDCBA9876543210ull;  // indx[pos] = pos
    int iz1 = 1, nsetp = 0, npp1 = 1;
    const int iz2 = n;
    int izmax = 0, j = 0, jj = 0;
    double up = 0.0;
    for (int i = 1; i <= n; ++i) ws.x(i) = 0.0;

    for (;;) {  // step two: dual variables of the columns still at their bound
        if (iz1 > iz2 || nsetp >= m) break;
        for (int iz = iz1; iz <= iz2; ++iz) {
            j = indx.get(iz);
            double sdot = 0.0;
            for (int r = npp1; r <= m; ++r) sdot += ws.A(r, j) * ws.b(r);
            ws.w(j) = sdot;
        }
        bool found = false;
        for (;;) {  // step three / four
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
            // step five: does column j enter the positive set?
            const double asave = ws.A(npp1, j);
            h12_construct<N>(ws, j, npp1, npp1 + 1, m, up);
            const double unorm = nrm2_by(nsetp, [&](int i) { return ws.A(i + 1, j); });
            const double t = factor * __builtin_fabs(ws.A(npp1, j));
            const double d1 = unorm + t;
            if (d1 - unorm > 0.0) {
                for (int r = 1; r <= m; ++r) ws.z(r) = ws.b(r);
                h12_apply<N>(ws, j, npp1, npp1 + 1, m, up, [&](int r) -> double & { return ws.z(r); });
                if (ws.z(npp1) / ws.A(npp1, j) > 0.0) found = true;
            }
            if (found) {
                for (int r = 1; r <= m; ++r) ws.b(r) = ws.z(r);
                indx.set(iz, indx.get(iz1));
                indx.set(iz1, j);
                ++iz1;
                nsetp = npp1;
                ++npp1;
                for (int jz = iz1; jz <= iz2; ++jz) {
                    jj = indx.get(jz);
                    const int cj = jj;
                    h12_apply<N>(ws, j, nsetp, npp1, m, up,
                                 [&](int r) -> double & { return ws.A(r, cj); });
                }
                ws.w(j) = 0.0;
                for (int r = npp1; r <= m; ++r) ws.A(r, j) = 0.0;
                break;
            }
            ws.A(npp1, j) = asave;
            ws.w(j) = 0.0;
        }
        if (!found) break;

        for (;;) {  // step six: solve the triangular system for z
            for (int ip = nsetp; ip >= 1; --ip) {
                if (ip != nsetp) {
                    const double zip1 = ws.z(ip + 1);
                    for (int i = 1; i <= ip; ++i) ws.z(i) -= zip1 * ws.A(i, jj);
                }
                jj = indx.get(ip);
                ws.z(ip) /= ws.A(ip, jj);
            }
            ++iter;
            if (iter > itmax) { mode = 3; goto done; }
            // steps seven..ten: step length
            double alpha = 1.0;
            jj = 0;
            for (int ip = 1; ip <= nsetp; ++ip) {
                const double zi = ws.z(ip);
                if (zi > 0.0) continue;
                const int l = indx.get(ip);
                const double xl = ws.x(l);
                const double t = -xl / (zi - xl);
                if (alpha < t) continue;
                alpha = t;
                jj = ip;
            }
            for (int ip = 1; ip <= nsetp; ++ip) {
                const int l = indx.get(ip);
                ws.x(l) = (1.0 - alpha) * ws.x(l) + alpha * ws.z(ip);
            }
            if (jj == 0) break;  // back to step two
            // step eleven: move coefficient i from set P to set Z
            int i = indx.get(jj);
            for (;;) {
                ws.x(i) = 0.0;
                ++jj;
                for (j = jj; j <= nsetp; ++j) {
                    const int ii = indx.get(j);
                    indx.set(j - 1, ii);
                    double c, s;
                    double a0 = ws.A(j - 1, ii), a1 = ws.A(j, ii);
                    rotg(a0, a1, c, s);
                    ws.A(j - 1, ii) = a0;
                    ws.A(j, ii) = a1;
                    const double t = a0;
                    for (int col = 1; col <= n; ++col) {
                        const double xi = ws.A(j - 1, col), yi = ws.A(j, col);
                        ws.A(j - 1, col) = c * xi + s * yi;
                        ws.A(j, col) = c * yi - s * xi;
                    }
                    ws.A(j - 1, ii) = t;
                    ws.A(j, ii) = 0.0;
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
            for (int r = 1; r <= m; ++r) ws.z(r) = ws.b(r);
        }
    }
done: {
        const int k = (npp1 < m) ? npp1 : m;
        rnorm = nrm2_by(m - nsetp, [&](int i) { return ws.b(k + i); });
    }
    return mode;
}

// index of element (row j, column i), j >= i, in the column-packed LDL' array
template <int N>
OPTIK_DEV constexpr int lidx(int i, int j) { return i * N - (i * (i - 1)) / 2 + (j - i); }

// Kraft LSQ for m = 0 and finite bounds:  min ||E s - f||, lo <= s <= hi.
// Returns the LSQ mode (1 ok).
template <int N>
OPTIK_DEV int lsq_box(const NnlsWs<N> &ws, const double (&l)[N * (N + 1) / 2], const double (&g)[N],
                      const double (&lo)[N], const double (&hi)[N], double (&s)[N]) {
    double E[N][N];  // upper triangular; [i][j] used for j >= i
    double f[N];
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
    // transform G = [I; -I] and h = [lo; -hi]: rows of +-E^-1
    bool singular = false;
#pragma unroll
    for (int j = 0; j < N; ++j) singular = singular || !(__builtin_fabs(E[j][j]) >= EPMACH);
    if (singular) return 5;
    double Gi[N][N];  // row i of E^-1, entries j >= i
    double h[2 * N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = i; j < N; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int k = i; k < j; ++k) acc += Gi[i][k] * E[k][j];
            Gi[i][j] = (((j == i) ? 1.0 : 0.0) - acc) / E[j][j];
        }
        double acc = 0.0;
#pragma unroll
        for (int j = i; j < N; ++j) acc += Gi[i][j] * f[j];
        h[i] = lo[i] - acc;
        h[N + i] = (-hi[i]) - (-acc);
        OPTIK_SCHED_FENCE();
    }
    // LDP.  NNLS's first dual check is w_j = h_j (b = e_{n+1}); when no h_j is positive
    // it returns at once with zero multipliers (y = 0, fac = 1, step 0): that case --
    // the unconstrained step is feasible -- never touches LDS.  Bit-identical to
    // running NNLS, whose every product is then an exact zero.
    constexpr int M = 2 * N;
    bool need_nnls = false;
#pragma unroll
    for (int r = 0; r < M; ++r) need_nnls = need_nnls || (h[r] > 0.0);
    int mode = 1;
    if (need_nnls) {
#pragma unroll
        for (int c = 0; c < N; ++c) {
#pragma unroll
            for (int r = 0; r < N; ++r) {
                const double v = (r >= c) ? Gi[c][r] : 0.0;
                ws.A(r + 1, c + 1) = v;
                ws.A(r + 1, N + c + 1) = (r >= c) ? -v : 0.0;
            }
            ws.A(N + 1, c + 1) = h[c];
            ws.A(N + 1, N + c + 1) = h[N + c];
            OPTIK_SCHED_FENCE();
        }
#pragma unroll
        for (int r = 1; r <= N; ++r) ws.b(r) = 0.0;
        ws.b(N + 1) = 1.0;
        double rnorm;
        OPTIK_SCHED_FENCE();
        mode = nnls<N>(ws, rnorm);
        OPTIK_SCHED_FENCE();
        if (mode == 1 && rnorm <= 0.0) mode = 4;
        if (mode == 1) {
            double y[M];
#pragma unroll
            for (int r = 0; r < M; ++r) y[r] = ws.x(r + 1);
            double hy = 0.0;
#pragma unroll
            for (int r = 0; r < M; ++r) hy += h[r] * y[r];
            double fac = 1.0 - hy;
            const double d1 = 1.0 + fac;
            if (d1 - 1.0 <= 0.0) {
                mode = 4;
            } else {
                fac = 1.0 / fac;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r <= j; ++r) acc += Gi[r][j] * y[r];
#pragma unroll
                    for (int r = 0; r <= j; ++r) acc += (-Gi[r][j]) * y[N + r];
                    s[j] = fac * acc;
                    OPTIK_SCHED_FENCE();
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) s[j] = 0.0;
    }
    if (mode != 1) return mode;
    // solution of the original problem: s = E^-1 (y + f)
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
    // NLopt: enforce the bounds against roundoff
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (s[i] < lo[i]) s[i] = lo[i];
        else if (s[i] > hi[i]) s[i] = hi[i];
    }
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
