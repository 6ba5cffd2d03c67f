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

// LDS window of one group: the columns of set P by position (column p at doubles
// [8 (p-1), 8 p): the triangular factor the solve of step six walks), then the solution z.
// The registers stay the master copy; the window only serves reads whose index differs
// from problem to problem (a register file cannot be indexed per lane, and selecting among
// eight registers costs 14 instructions per value).  Stride 74: 16-byte aligned, and the 16
// groups of a wave fall on disjoint banks for 16-byte reads.
constexpr int COOP_WIN_Z = 64;
constexpr int COOP_WIN = 74;
// doubles of LDS per wave: the windows of its groups, then one column of zeros
template <int CPL>
constexpr int coop_wave_lds() { return (64 / (COOP_COLS / CPL)) * COOP_WIN + 8; }

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

// What a suspended problem carries to its next launch besides the (transformed) matrix:
// everything else the loop holds at step two.  Group-uniform fields are replicated.
template <int CPL>
struct CoopCarry {
    dvec8 b;
    double up;
    int nsetp, iter;
    int pos[CPL];
    double xv[CPL];
};

constexpr int NNLS_SUSPENDED = 7;  // mode: pass budget of this launch used up, state in `cs`

// -DOPTIK_PROFILE_NNLS: wave cycles per step of the loop below, accumulated into g_nnls_prof[8]
// (tools/nnls_step_profile.py): 0 steps two-three, 1 step five construction, 2 step five applied to
// the columns, 3 step six, 4 steps seven-ten, 5 step eleven, 6 loop trips, 7 calls
#ifdef OPTIK_PROFILE_NNLS
__device__ unsigned long long g_nnls_prof[8];
#define NNLS_PROBE(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); np_[slot] += now_ - nt_; nt_ = now_; } while (0)
#else
#define NNLS_PROBE(slot)
#endif

// Solves the problem whose columns cid0+1 .. cid0+CPL (1-based ids; ids > 2N are padding)
// this lane holds in col[].  `live` = the group has a problem; `resume` = cs holds the
// state of a suspended problem (else it is initialised here).  At most `budget` solve
// passes are started from step two in this call; a problem that needs more returns mode
// NNLS_SUSPENDED with col[] / cs ready to be resumed -- the arithmetic sequence of a
// problem does not depend on where it is cut.  On return cs.xv[] are the lane's
// multipliers; group-uniform results: mode (1 ok, 3 iteration cap), rnorm, and the total
// number of solve passes.
//
// `park(col, cs)` is called by the lanes of a group at the moment it is suspended, with cs
// filled in (the caller stores matrix and state there and then: keeping the matrix live
// past the loop just to store it costs ~280 B of scratch per lane).
//
// `win` = the group's LDS window, `zeros` = eight doubles of +0.0 in LDS (read only).
template <int N, int CPL, class Park>
OPTIK_DEV void nnls_coop(bool live, bool resume, int budget, int cid0, dvec8 (&col)[CPL], CoopCarry<CPL> &cs,
                         int &mode_out, double &rnorm_out, int &iters_out, double *win, const double *zeros,
                         Park &&park) {
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
    b = vsel(resume, cs.b, b);
    int nsetp = resume ? cs.nsetp : 0, iter = resume ? cs.iter : 0, mode = 1;
    int npp1 = nsetp + 1;
    double up = resume ? cs.up : 0.0;
    const int iter_stop = iter + budget;
    double (&xv)[CPL] = cs.xv;
    // per-column state
    int pos[CPL];      // position of the column in the permutation (indx[pos] = id)
    bool inZ[CPL], isc[CPL];
    double wv[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        isc[k] = cid0 + k + 1 <= n;
        pos[k] = resume ? cs.pos[k] : cid0 + k + 1;
        inZ[k] = isc[k] && pos[k] > nsetp;  // set P holds positions 1 .. nsetp
        wv[k] = 0.0;
        xv[k] = resume ? xv[k] : 0.0;
    }
    int rem_jj = 0;  // step eleven: position being removed
    // copies the lane's columns that are in set P to the window, each at its position
    auto mirror = [&](bool on) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            if (on && isc[k] && !inZ[k]) {
                double *d = win + (pos[k] - 1) * 8;
#pragma unroll
                for (int r = 0; r < 8; ++r) d[r] = col[k][r];
            }
        }
    };
    mirror(live && resume);
    // phases: 0 = step two (recompute duals, then choose), 1 = step three (choose again),
    // 2 = step six (solve), 3 = step eleven (remove), 4 = done, 5 = suspended at step two
    int phase = live ? 0 : 4;
#ifdef OPTIK_PROFILE_NNLS
    unsigned long long np_[8] = {0, 0, 0, 0, 0, 0, 0, 1};
    unsigned long long nt_ = __builtin_readcyclecounter();
#endif

    // The loop body is straight-line code: every conditional update is a select, and rows /
    // columns an update must not touch get an exact no-op instead of a branch -- products
    // with a zero weight, and `x + (-0.0)` (which returns x bit-for-bit, signed zeros
    // included) where an addend has to be neutralised.  Divergent branches here cost more
    // in exec-mask handling and register copies than the arithmetic they skip.
    while (wave_any(phase < 4)) {
#ifdef OPTIK_PROFILE_NNLS
        np_[6] += 1;
        nt_ = __builtin_readcyclecounter();
#endif
        // ---------------- steps two .. five --------------------------------------------
        if (wave_any(phase == 0 || phase == 1)) {
            const bool inA = (phase == 0 || phase == 1);
            if (inA && (nsetp + 1 > n || nsetp >= m)) phase = 4;  // iz1 > iz2 || nsetp >= m
            const bool run = (phase == 0 || phase == 1);
            if (wave_any(phase == 0)) {
                // step two: duals of the columns in Z over rows npp1 .. m
                dvec8 bm = 0.0;
#pragma unroll
                for (int r = 1; r <= m; ++r) bm[r - 1] = (r >= npp1) ? b[r - 1] : 0.0;
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    double sdot = 0.0;
#pragma unroll
                    for (int r = 1; r <= m; ++r) sdot += col[k][r - 1] * bm[r - 1];
                    wv[k] = (phase == 0 && inZ[k]) ? sdot : wv[k];
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
                const bool better = (w > bw) || (w == bw && p < bp);
                bw = better ? w : bw;
                bp = better ? p : bp;
            }
#pragma unroll
            for (int off = G / 2; off >= 1; off >>= 1) {
                const double ow = __shfl_xor(bw, off, 64);
                const int op = __shfl_xor(bp, off, 64);
                const bool better = (ow > bw) || (ow == bw && op < bp);
                bw = better ? ow : bw;
                bp = better ? op : bp;
            }
            const bool none = !(bw > 0.0);
            if (run && none) phase = 4;  // step four: every dual <= 0 -> done
            // out of budget with work left: suspend before anything is modified (the duals
            // are recomputed from the same data on resume)
            if (phase == 0 && iter >= iter_stop) {
                phase = 5;
                cs.b = b;
                cs.up = up;
                cs.nsetp = nsetp;
                cs.iter = iter;
#pragma unroll
                for (int k = 0; k < CPL; ++k) cs.pos[k] = pos[k];
                park(col, cs);
            }
            const bool cand = run && !none && phase != 5;
            NNLS_PROBE(0);
            // (the last trip of most waves: every problem left has just finished)
            if (wave_any(cand)) {
            // step five: Householder construction on the chosen column j (position bp)
            bool hitk[CPL];
            bool hit_any = false;
            dvec8 mine = 0.0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                hitk[k] = cand && inZ[k] && pos[k] == bp;
                hit_any = hit_any || hitk[k];
                mine = vsel(hitk[k], col[k], mine);
            }
            const int jl = Gr::find(hit_any);
            const dvec8 u = Gr::bcast(mine, jl < 0 ? 0 : jl);
            // (all lanes run the construction; only `cand` groups keep its results)
            const double asave = vpick(u, npp1);
            const bool h12_live = npp1 < m;
            dvec8 um = 0.0, ul = 0.0;  // u below the pivot row / above it, zero elsewhere
#pragma unroll
            for (int r = 1; r <= m; ++r) {
                um[r - 1] = (r > npp1) ? u[r - 1] : 0.0;
                ul[r - 1] = (r <= nsetp) ? u[r - 1] : 0.0;
            }
            double cl = __builtin_fabs(asave);
#pragma unroll
            for (int r = 1; r <= m; ++r) {
                const double sm = __builtin_fabs(um[r - 1]);
                cl = (sm > cl) ? sm : cl;
            }
            const bool pivot = h12_live && !(cl <= 0.0);
            double ulp = asave;
            {
                const double clinv = 1.0 / cl;
                double d = asave * clinv;
                double sm = d * d;
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    d = um[r - 1] * clinv;
                    sm += d * d;
                }
                double c2 = cl * __builtin_sqrt(sm);
                c2 = (asave > 0.0) ? -c2 : c2;
                up = (cand && pivot) ? asave - c2 : up;
                ulp = pivot ? c2 : ulp;
            }
            // Lawson-Hanson's independence test diff(unorm + factor |ulp|, unorm) > 0 with unorm the
            // norm of the column above the pivot.  unorm <= sqrt(m) xmax < 3 xmax, and an addend
            // above 2^-52 unorm always survives the rounding of the sum: whenever
            // t > 3 * 2^-52 * xmax the test is true without unorm (also for xmax = 0: t > 0 is the
            // test itself).  Only if some problem of the wave is not decided that way (t within
            // 1e-15 of the column norm: a numerically dependent column) is unorm formed.
            double xmax = 0.0;
#pragma unroll
            for (int r = 1; r <= m; ++r) {
                const double av = __builtin_fabs(ul[r - 1]);
                xmax = (av > xmax) ? av : xmax;
            }
            const double t = factor * __builtin_fabs(ulp);
            bool ok1 = t > 6.7e-16 * xmax;
            if (wave_any(cand && !ok1)) {
                const double scale = 1.0 / xmax;
                double sum = 0.0;
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    const double xs = scale * ul[r - 1];
                    sum += xs * xs;
                }
                const double unorm = (xmax != 0.0) ? xmax * __builtin_sqrt(sum) : 0.0;
                const double d1 = unorm + t;
                ok1 = d1 - unorm > 0.0;
            }
            const double hprod = up * ulp;
            const bool apply_live = cand && h12_live && !(__builtin_fabs(ulp) <= 0.0) && !(hprod >= 0.0);
            const double hb = apply_live ? 1.0 / hprod : 0.0;
            // the transformation as a weight per row: up at the pivot row, u below, 0 above
            dvec8 w = 0.0;
#pragma unroll
            for (int r = 1; r <= m; ++r) w[r - 1] = (r == npp1) ? up : um[r - 1];
            // b := Q b on a copy
            dvec8 zt = b;
            {
                double sm = 0.0;
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    const double pr = zt[r - 1] * w[r - 1];
                    sm = (r == 1) ? pr : sm + pr;
                }
                const bool act = apply_live && ok1 && sm != 0.0;
                const double smh = act ? sm * hb : 0.0;
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    const double add = smh * w[r - 1];
                    zt[r - 1] = (act && r >= npp1) ? zt[r - 1] + add : zt[r - 1];
                }
            }
            const bool found = cand && ok1 && (vpick(zt, npp1) / ulp > 0.0);
            // b := Q b; column j takes position iz1 = nsetp + 1, the column there takes j's
            b = vsel(found, zt, b);
            {
                const int iz1 = nsetp + 1;
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    const bool me = found && hitk[k];
                    const bool other = found && isc[k] && pos[k] == iz1 && !hitk[k];
                    pos[k] = other ? bp : pos[k];
                    pos[k] = me ? iz1 : pos[k];
                    inZ[k] = me ? false : inZ[k];
                }
            }
            nsetp = found ? npp1 : nsetp;
            npp1 = nsetp + 1;
            NNLS_PROBE(1);
            // the column that entered P: untouched above the pivot row, ulp on it (the rows below
            // are never read from the window)
            if (found && Gr::lane() == 0) {
                double *d = win + (nsetp - 1) * 8;
#pragma unroll
                for (int r = 0; r < 8; ++r) d[r] = u[r];
                d[nsetp - 1] = ulp;
            }
            // rows the transformation leaves alone get -0.0 added (sign bit forced on a zero)
            unsigned rowkeep[8];
            dvec8 newv = 0.0;  // the chosen column after the transformation
#pragma unroll
            for (int r = 1; r <= m; ++r) {
                rowkeep[r - 1] = (r < nsetp) ? 0x80000000u : 0u;
                newv[r - 1] = (r == nsetp) ? ulp : 0.0;
            }
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                double sm = 0.0;
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    const double pr = col[k][r - 1] * w[r - 1];
                    sm = (r == 1) ? pr : sm + pr;
                }
                const bool act = found && apply_live && inZ[k] && sm != 0.0;
                const double smh = act ? sm * hb : 0.0;
                const unsigned colkeep = act ? 0u : 0x80000000u;
                const bool chosen = found && hitk[k];
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    const double add = smh * w[r - 1];  // +-0 wherever the row or column is kept
                    const double addz = __hiloint2double((int)((unsigned)__double2hiint(add) | rowkeep[r - 1] | colkeep),
                                                         __double2loint(add));
                    const double v = col[k][r - 1] + addz;
                    col[k][r - 1] = (chosen && r >= nsetp) ? newv[r - 1] : v;
                }
                wv[k] = (cand && hitk[k]) ? 0.0 : wv[k];
            }
            // found: solve (step six); else choose again without recomputing the duals
            phase = cand ? (found ? 2 : 1) : phase;
            NNLS_PROBE(2);
            }
        }
        // ---------------- steps six .. ten ---------------------------------------------
        if (wave_any(phase == 2)) {
            const bool run = phase == 2;
            dvec8 zz = b;  // (steps five / eleven leave z := b)
            int nmax = 0;
#pragma unroll
            for (int v = 1; v <= m; ++v)
                if (wave_any(run && nsetp >= v)) nmax = v;
            // step six: solve the triangular system on set P (positions nsetp .. 1); column ip of the
            // factor comes from the window.  A group whose set P ends below ip reads the zeros
            // instead and subtracts 0 * 0 = +0, which changes nothing (x - (+0) == x bit for bit).
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int ppos[CPL];  // position of the lane's columns that are in P (1 .. nsetp), else 0
#pragma unroll
            for (int k = 0; k < CPL; ++k) ppos[k] = (run && isc[k] && !inZ[k]) ? pos[k] : 0;
#pragma unroll
            for (int ip = m; ip >= 1; --ip) {
                if (ip > nmax) continue;
                const bool step = run && ip <= nsetp;
                const double *cp = step ? win + (ip - 1) * 8 : zeros;
                dvec8 cv = 0.0;
#pragma unroll
                for (int r = 1; r <= ip; ++r) cv[r - 1] = cp[r - 1];
                const double zi = zz[ip - 1] / cv[ip - 1];
                const double zie = step ? zi : 0.0;
                zz[ip - 1] = step ? zi : zz[ip - 1];
#pragma unroll
                for (int r = 1; r < ip; ++r) zz[r - 1] = zz[r - 1] - zie * cv[r - 1];
            }
            // z of the lane's own columns, by position
            if (run && Gr::lane() == 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r) win[COOP_WIN_Z + r] = zz[r];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double zown[CPL];
#pragma unroll
            for (int k = 0; k < CPL; ++k) zown[k] = win[COOP_WIN_Z + (ppos[k] > 1 ? ppos[k] : 1) - 1];
            if (run) {
                ++iter;
                if (iter > itmax) { mode = 3; phase = 4; }
            }
            NNLS_PROBE(3);
            const bool go = phase == 2;
            // steps seven..ten: step length; scan the positions in order, as the serial code does
            double alpha = 1.0;
            int jj = 0;
#pragma unroll
            for (int ip = 1; ip <= m; ++ip) {
                if (ip > nmax) continue;
                const bool step = go && ip <= nsetp;
                // only positions whose z is not positive limit the step: most have none in the whole wave
                if (!wave_any(step && !(zz[ip - 1] > 0.0))) continue;
                bool hit_any = false;
                double hx = 0.0;
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    const bool hit = go && ppos[k] == ip;
                    hit_any = hit_any || hit;
                    hx = hit ? xv[k] : hx;
                }
                const int ol = Gr::find(hit_any);
                const double xp = Gr::bcast(hx, ol < 0 ? 0 : ol);
                const double zp = zz[ip - 1];
                const double tq = -xp / (zp - xp);
                const bool take = step && ol >= 0 && !(zp > 0.0) && !(alpha < tq);
                alpha = take ? tq : alpha;
                jj = take ? ip : jj;
            }
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const double nx = (1.0 - alpha) * xv[k] + alpha * zown[k];
                xv[k] = (go && isc[k] && !inZ[k]) ? nx : xv[k];
            }
            rem_jj = (go && jj != 0) ? jj : rem_jj;
            phase = go ? (jj == 0 ? 0 : 3) : phase;  // back to step two, or remove position jj
            NNLS_PROBE(4);
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
            mirror(run);  // the columns left in P moved and were rotated
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
                    phase = 2;
                }
            }
            NNLS_PROBE(5);
        }
    }
#ifdef OPTIK_PROFILE_NNLS
    if ((threadIdx.x & 63u) == 0 && live)
        for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_nnls_prof[i_], np_[i_]);
#endif
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
    mode_out = phase == 5 ? NNLS_SUSPENDED : mode;
    iters_out = iter;
}

}  // namespace optik
