// ik_nnls_coop.hpp -- Lawson-Hanson NNLS of the LSQ dual, cooperative register form.
//
// One problem per group of G = 16 / CPL lanes (64 / G problems per wave): each lane owns
// CPL columns of the (n+1) x 2n matrix in REGISTERS (LLVM vector values), plus their duals
// w, multipliers x and positions in the permutation.  Everything the textbook loop does
// "for each column in Z" is a few instructions across the group; everything scalar (the
// Householder construction, the triangular solve, the step length) is computed
// redundantly by the lanes of the group from values broadcast with ds_bpermute, so a
// group never diverges internally.  No LDS storage.  CPL trades scalar redundancy
// (CPL = 1: 4 problems per wave) against the spread of iteration counts inside a wave
// (CPL = 16: one lane per problem, 64 problems per wave).
//
// Arithmetic per matrix element, its order, and every decision (argmax ties by
// position, step-length scan order, Givens updates) are those of the per-lane LDS
// implementation in ik_slsqp.hpp and of oracle/optik_oracle.c:nnls -- results are
// bit-identical.
#pragma once

#include "ik_slsqp.hpp"

namespace optik {

constexpr int COOP_COLS = 16;  // columns a group covers (2n <= 16)

template <int G>
struct Group {
    static OPTIK_DEV int base() { return (int)(threadIdx.x & 63u) & ~(G - 1); }
    static OPTIK_DEV int lane() { return (int)(threadIdx.x & (unsigned)(G - 1)); }
    // value held by lane `src` of the caller's group
    static OPTIK_DEV double bcast(double v, int src) {
        if (G == 1) return v;
        return __shfl(v, base() + src, 64);
    }
    static OPTIK_DEV int bcast(int v, int src) {
        if (G == 1) return v;
        return __shfl(v, base() + src, 64);
    }
    static OPTIK_DEV dvec8 bcast(const dvec8 v, int src) {
        if (G == 1) return v;
        dvec8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = __shfl(v[i], base() + src, 64);
        return o;
    }
    // lane of the caller's group for which p holds, or -1
    static OPTIK_DEV int find(bool p) {
        if (G == 1) return p ? 0 : -1;
        const unsigned long long m = __ballot(p);
        const unsigned g = (unsigned)((m >> base()) & ((1ull << G) - 1ull));
        return g ? (__ffs((int)g) - 1) : -1;
    }
};

// Solves the problem whose columns cid0+1 .. cid0+CPL (1-based ids; ids > 2N are padding)
// this lane holds in col[].  `live` = the group has a problem.  On return xv[] are the
// lane's multipliers; group-uniform results: mode (1 ok, 3 iteration cap), rnorm, and
// the number of solve passes (for scheduling statistics).
template <int N, int CPL>
OPTIK_DEV void nnls_coop(bool live, int cid0, dvec8 (&col)[CPL], double (&xv)[CPL], int &mode_out,
                         double &rnorm_out, int &iters_out) {
    constexpr int m = N + 1, n = 2 * N;
    constexpr int G = COOP_COLS / CPL;
    static_assert(m <= 8, "a column is one dvec8");
    static_assert(COOP_COLS % CPL == 0 && n <= COOP_COLS, "column tiling");
    using Gr = Group<G>;
    const double factor = 0.01;
    const int itmax = 3 * n;
    // group-uniform state (replicated in every lane of the group)
    dvec8 b = 0.0;
    b[m - 1] = 1.0;
    int nsetp = 0, npp1 = 1, iter = 0, mode = 1;
    double up = 0.0;
    // per-column state
    int pos[CPL];      // position of the column in the permutation (indx[pos] = id)
    bool inZ[CPL], isc[CPL];
    double wv[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        pos[k] = cid0 + k + 1;
        isc[k] = pos[k] <= n;
        inZ[k] = isc[k];
        wv[k] = 0.0;
        xv[k] = 0.0;
    }
    dvec8 zz = 0.0;
    int rem_jj = 0;  // step eleven: position being removed
    // phases: 0 = step two (recompute duals, then choose), 1 = step three (choose again),
    // 2 = step six (solve), 3 = step eleven (remove), 4 = done
    int phase = live ? 0 : 4;

    while (wave_any(phase != 4)) {
        // ---------------- steps two .. five --------------------------------------------
        if (wave_any(phase == 0 || phase == 1)) {
            const bool inA = (phase == 0 || phase == 1);
            if (inA && (nsetp + 1 > n || nsetp >= m)) phase = 4;  // iz1 > iz2 || nsetp >= m
            const bool run = (phase == 0 || phase == 1);
            if (phase == 0) {
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    if (!inZ[k]) continue;
                    double sdot = 0.0;
#pragma unroll
                    for (int r = 1; r <= m; ++r)
                        if (r >= npp1) sdot += col[k][r - 1] * b[r - 1];
                    wv[k] = sdot;
                }
            }
            // step three: largest positive dual among Z, ties to the smallest position
            double bw = 0.0;
            int bp = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const bool c = run && inZ[k] && wv[k] > 0.0;
                const double w = c ? wv[k] : 0.0;
                const int p = c ? pos[k] : 0x7fffffff;
                if ((w > bw) || (w == bw && p < bp)) { bw = w; bp = p; }
            }
#pragma unroll
            for (int off = G / 2; off >= 1; off >>= 1) {
                const double ow = __shfl_xor(bw, off, 64);
                const int op = __shfl_xor(bp, off, 64);
                if ((ow > bw) || (ow == bw && op < bp)) { bw = ow; bp = op; }
            }
            const bool none = !(bw > 0.0);
            if (run && none) phase = 4;  // step four: every dual <= 0 -> done
            const bool cand = run && !none;
            // step five: Householder construction on the chosen column j (position bp)
            int myk = -1;
            dvec8 mine = 0.0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const bool hit = cand && inZ[k] && pos[k] == bp;
                myk = hit ? k : myk;
                mine = vsel(hit, col[k], mine);
            }
            const int jl = Gr::find(myk >= 0);
            const dvec8 u = Gr::bcast(mine, jl < 0 ? 0 : jl);
            if (cand) {
                const double asave = vpick(u, npp1);
                const bool h12_live = npp1 < m;
                double ulp = asave;
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
                    }
                }
                double unorm = 0.0;
                {
                    double xmax = 0.0;
#pragma unroll
                    for (int r = 1; r <= m; ++r) {
                        const double av = __builtin_fabs(u[r - 1]);
                        if (r <= nsetp && av > xmax) xmax = av;
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
                double hb = 0.0;
                bool apply_live = false;
                if (h12_live && !(__builtin_fabs(ulp) <= 0.0)) {
                    hb = up * ulp;
                    if (!(hb >= 0.0)) { hb = 1.0 / hb; apply_live = true; }
                }
                bool found = false;
                dvec8 zt = b;
                if (d1 - unorm > 0.0) {
                    if (apply_live) {
                        double sm = vpick(zt, npp1) * up;
#pragma unroll
                        for (int r = 1; r <= m; ++r)
                            if (r > npp1) sm += zt[r - 1] * u[r - 1];
                        if (sm != 0.0) {
                            sm *= hb;
#pragma unroll
                            for (int r = 1; r <= m; ++r) {
                                if (r == npp1) zt[r - 1] += sm * up;
                                else if (r > npp1) zt[r - 1] += sm * u[r - 1];
                            }
                        }
                    }
                    if (vpick(zt, npp1) / ulp > 0.0) found = true;
                }
                if (found) {
                    // b := Q b; column j takes position iz1 = nsetp + 1, the column there takes j's
                    b = zt;
                    const int iz1 = nsetp + 1;
#pragma unroll
                    for (int k = 0; k < CPL; ++k) {
                        const bool me = (k == myk);
                        if (isc[k] && pos[k] == iz1 && !me) pos[k] = bp;
                        if (me) { pos[k] = iz1; inZ[k] = false; }
                    }
                    nsetp = npp1;
                    ++npp1;
#pragma unroll
                    for (int k = 0; k < CPL; ++k) {
                        if (apply_live && inZ[k]) {
                            double sm = vpick(col[k], nsetp) * up;
#pragma unroll
                            for (int r = 1; r <= m; ++r)
                                if (r >= npp1) sm += col[k][r - 1] * u[r - 1];
                            if (sm != 0.0) {
                                sm *= hb;
#pragma unroll
                                for (int r = 1; r <= m; ++r) {
                                    if (r == nsetp) col[k][r - 1] += sm * up;
                                    else if (r >= npp1) col[k][r - 1] += sm * u[r - 1];
                                }
                            }
                        }
                        if (k == myk) {
#pragma unroll
                            for (int r = 1; r <= m; ++r) {
                                if (r == nsetp) col[k][r - 1] = ulp;
                                else if (r >= npp1) col[k][r - 1] = 0.0;
                            }
                            wv[k] = 0.0;
                        }
                    }
                    zz = b;
                    phase = 2;
                } else {
#pragma unroll
                    for (int k = 0; k < CPL; ++k)
                        if (k == myk) wv[k] = 0.0;
                    phase = 1;  // choose again without recomputing the duals
                }
            }
        }
        // ---------------- steps six .. ten ---------------------------------------------
        if (wave_any(phase == 2)) {
            const bool run = phase == 2;
            // step six: solve the triangular system on set P (positions nsetp .. 1)
            int nmax = run ? nsetp : 0;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(nmax, off, 64); nmax = o > nmax ? o : nmax; }
            for (int ip = nmax; ip >= 1; --ip) {
                const bool step = run && ip <= nsetp;
                bool hit_any = false;
                dvec8 mine = 0.0;
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    const bool hit = step && isc[k] && !inZ[k] && pos[k] == ip;
                    hit_any = hit_any || hit;
                    mine = vsel(hit, col[k], mine);
                }
                const int ol = Gr::find(hit_any);
                const dvec8 cv = Gr::bcast(mine, ol < 0 ? 0 : ol);
                if (step) {
                    const double zi = vpick(zz, ip) / vpick(cv, ip);
#pragma unroll
                    for (int r = 1; r <= m; ++r) {
                        if (r == ip) zz[r - 1] = zi;
                        else if (r < ip) zz[r - 1] -= zi * cv[r - 1];
                    }
                }
            }
            if (run) {
                ++iter;
                if (iter > itmax) { mode = 3; phase = 4; }
            }
            const bool go = phase == 2;
            // steps seven..ten: step length; scan the positions in order, as the serial code does
            double zown[CPL], tcand[CPL];
            bool neg[CPL];
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const bool inP = isc[k] && !inZ[k];
                zown[k] = vpick(zz, pos[k] > m ? m : (pos[k] < 1 ? 1 : pos[k]));
                neg[k] = go && inP && !(zown[k] > 0.0);
                tcand[k] = neg[k] ? (-xv[k] / (zown[k] - xv[k])) : 0.0;
            }
            double alpha = 1.0;
            int jj = 0;
            for (int ip = 1; ip <= nmax; ++ip) {
                const bool step = go && ip <= nsetp;
                bool hit_any = false, hneg = false;
                double ht = 0.0;
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    const bool hit = step && isc[k] && !inZ[k] && pos[k] == ip;
                    hit_any = hit_any || hit;
                    hneg = hit ? neg[k] : hneg;
                    ht = hit ? tcand[k] : ht;
                }
                const int ol = Gr::find(hit_any);
                const int src = ol < 0 ? 0 : ol;
                const double t = Gr::bcast(ht, src);
                const int isneg = Gr::bcast(hneg ? 1 : 0, src);
                if (step && ol >= 0 && isneg) {
                    if (!(alpha < t)) { alpha = t; jj = ip; }
                }
            }
#pragma unroll
            for (int k = 0; k < CPL; ++k)
                if (go && isc[k] && !inZ[k]) xv[k] = (1.0 - alpha) * xv[k] + alpha * zown[k];
            if (go) {
                if (jj == 0) phase = 0;  // back to step two
                else { rem_jj = jj; phase = 3; }
            }
        }
        // ---------------- step eleven ----------------------------------------------------
        if (wave_any(phase == 3)) {
            const bool run = phase == 3;
            // move the coefficient at position rem_jj from set P to set Z
            bool leaving[CPL];
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                leaving[k] = run && isc[k] && !inZ[k] && pos[k] == rem_jj;
                if (leaving[k]) xv[k] = 0.0;
            }
            const int jlo = run ? rem_jj + 1 : 0x7fffffff, jhi = run ? nsetp : 0;
            int wlo = jlo, whi = jhi;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const int a0 = __shfl_xor(wlo, off, 64), a1 = __shfl_xor(whi, off, 64);
                wlo = a0 < wlo ? a0 : wlo;
                whi = a1 > whi ? a1 : whi;
            }
            for (int j = wlo; j <= whi; ++j) {
                const bool step = run && j >= jlo && j <= jhi;
                const int jm1 = j - 1 < 1 ? 1 : (j - 1 > m ? m : j - 1), jc = j > m ? m : (j < 1 ? 1 : j);
                // the column at position j moves to position j-1; Givens on its rows j-1, j
                int myk = -1;
                double m0 = 0.0, m1 = 0.0;
                double a0s[CPL], a1s[CPL];
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    a0s[k] = vpick(col[k], jm1);
                    a1s[k] = vpick(col[k], jc);
                    const bool hit = step && isc[k] && !inZ[k] && pos[k] == j && !leaving[k];
                    myk = hit ? k : myk;
                    m0 = hit ? a0s[k] : m0;
                    m1 = hit ? a1s[k] : m1;
                }
                const int il = Gr::find(myk >= 0);
                const int src = il < 0 ? 0 : il;
                double a0 = Gr::bcast(m0, src), a1 = Gr::bcast(m1, src);
                if (step) {
                    double c, s;
                    rotg(a0, a1, c, s);
                    const double t = a0;
#pragma unroll
                    for (int k = 0; k < CPL; ++k) {
                        if (!isc[k]) continue;
                        const bool ii = (k == myk);
                        const double xi = ii ? a0 : a0s[k], yi = ii ? a1 : a1s[k];
                        const double nx = c * xi + s * yi;
                        const double ny = c * yi - s * xi;
                        vput(col[k], j - 1, ii ? t : nx);
                        vput(col[k], j, ii ? 0.0 : ny);
                        if (ii) pos[k] = j - 1;
                    }
                    const double bx = vpick(b, j - 1), by = vpick(b, j);
                    vput(b, j - 1, c * bx + s * by);
                    vput(b, j, c * by - s * bx);
                }
            }
            if (run) {
                npp1 = nsetp;
                --nsetp;
#pragma unroll
                for (int k = 0; k < CPL; ++k)
                    if (leaving[k]) { pos[k] = nsetp + 1; inZ[k] = true; }  // --iz1; indx[iz1] = i
                if (nsetp <= 0) { mode = 3; phase = 4; }
            }
            if (phase == 3) {
                // is every coefficient left in P feasible?  first offending position, in order
                int bad = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    const int p = (isc[k] && !inZ[k] && xv[k] <= 0.0) ? pos[k] : 0x7fffffff;
                    bad = p < bad ? p : bad;
                }
#pragma unroll
                for (int off = G / 2; off >= 1; off >>= 1) { const int o = __shfl_xor(bad, off, 64); bad = o < bad ? o : bad; }
                if (bad != 0x7fffffff) {
                    rem_jj = bad;  // again
                } else {
                    zz = b;
                    phase = 2;
                }
            }
        }
    }
    // rnorm = ||b(npp1..m)||
    {
        const int k0 = (npp1 < m) ? npp1 : m;
        const int cnt = m - nsetp;
        double xmax = 0.0;
#pragma unroll
        for (int r = 1; r <= m; ++r) {
            const double av = __builtin_fabs(b[r - 1]);
            if (r >= k0 && r < k0 + cnt && av > xmax) xmax = av;
        }
        double rn = 0.0;
        if (xmax != 0.0) {
            const double scale = 1.0 / xmax;
            double sum = 0.0;
#pragma unroll
            for (int r = 1; r <= m; ++r) {
                const double xs = scale * b[r - 1];
                if (r >= k0 && r < k0 + cnt) sum += xs * xs;
            }
            rn = xmax * __builtin_sqrt(sum);
        }
        rnorm_out = rn;
    }
    mode_out = mode;
    iters_out = iter;
}

}  // namespace optik
