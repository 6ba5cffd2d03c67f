// ik_nnls_coop.hpp -- Lawson-Hanson NNLS of the LSQ dual, cooperative form.
//
// One problem per 16-lane group (4 problems per wave): lane c of a group owns column
// c+1 of the (n+1) x 2n matrix in REGISTERS, plus its dual w_c, its multiplier x_c and
// its position in the permutation.  Everything that the textbook loop does "for each
// column in Z" is one instruction across the group; everything scalar (the Householder
// construction, the triangular solve, the step length) is computed redundantly by all
// 16 lanes from values broadcast with ds_bpermute, so the group never diverges
// internally.  No LDS storage, ~100 VGPRs -> several waves per SIMD, and the latency
// of one NNLS iteration is a few hundred instructions instead of a few thousand.
//
// Arithmetic per matrix element, its order, and every decision (argmax ties by
// position, step-length scan order, Givens updates) are those of the per-lane
// implementation in ik_slsqp.hpp and of oracle/optik_oracle.c:nnls -- results are
// bit-identical.
#pragma once

#include "ik_slsqp.hpp"

namespace optik {

constexpr int COOP_GROUP = 16;

// value of `v` held by lane `src` (0..15) of the caller's group
OPTIK_DEV double group_bcast(double v, int src) {
    const int base = (int)(threadIdx.x & 63u) & ~(COOP_GROUP - 1);
    return __shfl(v, base + src, 64);
}

OPTIK_DEV int group_bcast_i(int v, int src) {
    const int base = (int)(threadIdx.x & 63u) & ~(COOP_GROUP - 1);
    return __shfl(v, base + src, 64);
}

// lane (0..15) of the caller's group for which `p` holds, or -1
OPTIK_DEV int group_find(bool p) {
    const unsigned long long m = __ballot(p);
    const int base = (int)(threadIdx.x & 63u) & ~(COOP_GROUP - 1);
    const unsigned g = (unsigned)((m >> base) & 0xffffull);
    return g ? (__ffs((int)g) - 1) : -1;
}

OPTIK_DEV bool group_any(bool p) { return group_find(p) >= 0; }

// writes a[idx-1] = v (1-based per-lane index) without dynamic register indexing
template <int M>
OPTIK_DEV void put(double (&a)[M], int idx, double v) {
#pragma unroll
    for (int i = 0; i < M; ++i) a[i] = (idx == i + 1) ? v : a[i];
}

// Solves the problem whose column `cid` (1-based; lanes with cid > 2N idle) this lane
// holds in col[0..N].  `live` = the group has a problem.  On return xv is the lane's
// multiplier; the group-uniform results are mode (1 ok, 3 iteration cap) and rnorm.
template <int N>
OPTIK_DEV void nnls_coop(bool live, int cid, double (&col)[N + 1], double &xv, int &mode_out, double &rnorm_out) {
    constexpr int m = N + 1, n = 2 * N;
    const double factor = 0.01;
    const int itmax = 3 * n;
    const bool is_col = cid <= n;
    // group-uniform state (replicated in every lane of the group)
    double b[m];
#pragma unroll
    for (int r = 0; r < m; ++r) b[r] = (r == m - 1) ? 1.0 : 0.0;
    int nsetp = 0, npp1 = 1, iter = 0, mode = 1;
    double up = 0.0;
    // per-lane state
    int pos = cid;        // position of this column in the permutation (indx[pos] = cid)
    bool inZ = is_col;
    double wv = 0.0;
    xv = 0.0;
    double zz[m];
#pragma unroll
    for (int r = 0; r < m; ++r) zz[r] = 0.0;
    int rem_jj = 0;       // step eleven: position being removed
    // phases: 0 = step two (recompute duals, then choose), 1 = step three (choose again),
    // 2 = step six (solve), 3 = step eleven (remove), 4 = done
    int phase = live ? 0 : 4;

    while (wave_any(phase != 4)) {
        // ---------------- steps two .. five --------------------------------------------
        if (wave_any(phase == 0 || phase == 1)) {
            const bool inA = (phase == 0 || phase == 1);
            if (inA && (nsetp + 1 > n || nsetp >= m)) phase = 4;  // iz1 > iz2 || nsetp >= m
            const bool run = (phase == 0 || phase == 1);
            if (phase == 0 && inZ) {
                double sdot = 0.0;
#pragma unroll
                for (int r = 1; r <= m; ++r)
                    if (r >= npp1) sdot += col[r - 1] * b[r - 1];
                wv = sdot;
            }
            // step three: largest positive dual among Z, ties to the smallest position
            double bw = (run && inZ && wv > 0.0) ? wv : 0.0;
            int bp = (run && inZ && wv > 0.0) ? pos : 0x7fffffff;
#pragma unroll
            for (int off = COOP_GROUP / 2; off >= 1; off >>= 1) {
                const double ow = __shfl_xor(bw, off, 64);
                const int op = __shfl_xor(bp, off, 64);
                const bool take = (ow > bw) || (ow == bw && op < bp);
                if (take) { bw = ow; bp = op; }
            }
            const bool none = !(bw > 0.0);
            if (run && none) phase = 4;  // step four: every dual <= 0 -> done
            const bool cand = run && !none;
            // step five: Householder construction on the chosen column j (position bp)
            const int jl = group_find(cand && inZ && pos == bp);
            const int src = jl < 0 ? 0 : jl;
            double u[m];
#pragma unroll
            for (int r = 0; r < m; ++r) u[r] = group_bcast(col[r], src);
            if (cand) {
                const double asave = pick<m>(u, npp1);
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
                double zt[m];
#pragma unroll
                for (int r = 0; r < m; ++r) zt[r] = b[r];
                if (d1 - unorm > 0.0) {
                    if (apply_live) {
                        double sm = pick<m>(zt, npp1) * up;
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
                    if (pick<m>(zt, npp1) / ulp > 0.0) found = true;
                }
                const bool me = inZ && pos == bp;  // this lane owns column j
                if (found) {
                    // b := Q b; column j takes position iz1 = nsetp + 1, the column there takes j's
#pragma unroll
                    for (int r = 0; r < m; ++r) b[r] = zt[r];
                    const int iz1 = nsetp + 1;
                    if (is_col && pos == iz1 && !me) pos = bp;
                    if (me) { pos = iz1; inZ = false; }
                    nsetp = npp1;
                    ++npp1;
                    if (apply_live && inZ) {
                        double sm = pick<m>(col, nsetp) * up;
#pragma unroll
                        for (int r = 1; r <= m; ++r)
                            if (r >= npp1) sm += col[r - 1] * u[r - 1];
                        if (sm != 0.0) {
                            sm *= hb;
#pragma unroll
                            for (int r = 1; r <= m; ++r) {
                                if (r == nsetp) col[r - 1] += sm * up;
                                else if (r >= npp1) col[r - 1] += sm * u[r - 1];
                            }
                        }
                    }
                    if (me) {
#pragma unroll
                        for (int r = 1; r <= m; ++r) {
                            if (r == nsetp) col[r - 1] = ulp;
                            else if (r >= npp1) col[r - 1] = 0.0;
                        }
                        wv = 0.0;
                    }
#pragma unroll
                    for (int r = 0; r < m; ++r) zz[r] = b[r];
                    phase = 2;
                } else {
                    if (me) wv = 0.0;
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
                const int ol = group_find(step && is_col && !inZ && pos == ip);
                const int src = ol < 0 ? 0 : ol;
                double cv[m];
#pragma unroll
                for (int r = 0; r < m; ++r) cv[r] = group_bcast(col[r], src);
                if (step) {
                    const double zi = pick<m>(zz, ip) / pick<m>(cv, ip);
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
            const bool inP = is_col && !inZ;
            const double zown = pick<m>(zz, pos > m ? m : (pos < 1 ? 1 : pos));
            const bool neg = go && inP && !(zown > 0.0);
            const double tcand = neg ? (-xv / (zown - xv)) : 0.0;
            double alpha = 1.0;
            int jj = 0;
            for (int ip = 1; ip <= nmax; ++ip) {
                const bool step = go && ip <= nsetp;
                const int ol = group_find(step && inP && pos == ip);
                const int src = ol < 0 ? 0 : ol;
                const double t = group_bcast(tcand, src);
                const int isneg = group_bcast_i(neg ? 1 : 0, src);
                if (step && ol >= 0 && isneg) {
                    if (!(alpha < t)) { alpha = t; jj = ip; }
                }
            }
            if (go && inP) xv = (1.0 - alpha) * xv + alpha * zown;
            if (go) {
                if (jj == 0) phase = 0;  // back to step two
                else { rem_jj = jj; phase = 3; }
            }
        }
        // ---------------- step eleven ----------------------------------------------------
        if (wave_any(phase == 3)) {
            const bool run = phase == 3;
            // move the coefficient at position rem_jj from set P to set Z
            const bool leaving = run && is_col && !inZ && pos == rem_jj;
            if (leaving) xv = 0.0;
            int jlo = run ? rem_jj + 1 : 0x7fffffff, jhi = run ? nsetp : 0;
            int wlo = jlo, whi = jhi;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const int a0 = __shfl_xor(wlo, off, 64), a1 = __shfl_xor(whi, off, 64);
                wlo = a0 < wlo ? a0 : wlo;
                whi = a1 > whi ? a1 : whi;
            }
            for (int j = wlo; j <= whi; ++j) {
                const bool step = run && j >= jlo && j <= jhi;
                // the column at position j moves to position j-1; Givens on its rows j-1, j
                const int il = group_find(step && is_col && !inZ && pos == j && !leaving);
                const int src = il < 0 ? 0 : il;
                const double a0s = pick<m>(col, j - 1 < 1 ? 1 : j - 1), a1s = pick<m>(col, j > m ? m : j);
                double a0 = group_bcast(a0s, src), a1 = group_bcast(a1s, src);
                if (step) {
                    double c, s;
                    rotg(a0, a1, c, s);
                    const double t = a0;
                    const bool ii = (il >= 0) && ((int)(threadIdx.x & (COOP_GROUP - 1)) == il);
                    if (is_col) {
                        const double xi = ii ? a0 : a0s, yi = ii ? a1 : a1s;
                        const double nx = c * xi + s * yi;
                        const double ny = c * yi - s * xi;
                        put<m>(col, j - 1, ii ? t : nx);
                        put<m>(col, j, ii ? 0.0 : ny);
                    }
                    const double bx = pick<m>(b, j - 1), by = pick<m>(b, j);
                    put<m>(b, j - 1, c * bx + s * by);
                    put<m>(b, j, c * by - s * bx);
                    if (ii) pos = j - 1;
                }
            }
            if (run) {
                npp1 = nsetp;
                --nsetp;
                if (leaving) { pos = nsetp + 1; inZ = true; }  // --iz1; indx[iz1] = i
                if (nsetp <= 0) { mode = 3; phase = 4; }
            }
            if (phase == 3) {
                // is every coefficient left in P feasible?  first offending position, in order
                int bad = (is_col && !inZ && xv <= 0.0) ? pos : 0x7fffffff;
#pragma unroll
                for (int off = COOP_GROUP / 2; off >= 1; off >>= 1) { const int o = __shfl_xor(bad, off, 64); bad = o < bad ? o : bad; }
                if (bad != 0x7fffffff) {
                    rem_jj = bad;  // again
                } else {
#pragma unroll
                    for (int r = 0; r < m; ++r) zz[r] = b[r];
                    phase = 2;
                }
            }
        }
    }
    // rnorm = ||b(npp1..m)||
    {
        const int k = (npp1 < m) ? npp1 : m;
        const int cnt = m - nsetp;
        double xmax = 0.0;
#pragma unroll
        for (int r = 1; r <= m; ++r) {
            const double av = __builtin_fabs(b[r - 1]);
            if (r >= k && r < k + cnt && av > xmax) xmax = av;
        }
        double rn = 0.0;
        if (xmax != 0.0) {
            const double scale = 1.0 / xmax;
            double sum = 0.0;
#pragma unroll
            for (int r = 1; r <= m; ++r) {
                const double xs = scale * b[r - 1];
                if (r >= k && r < k + cnt) sum += xs * xs;
            }
            rn = xmax * __builtin_sqrt(sum);
        }
        rnorm_out = rn;
    }
    mode_out = mode;
}

}  // namespace optik
