// ik_nnls_quad.hpp -- Lawson-Hanson NNLS of the LSQ dual for the quad solver: matrix in LDS.
//
// Rounds 1-2 kept the (n+1) x 2n matrix in REGISTERS, four columns per lane (64 VGPRs) plus ~120 more for
// the masked copies of the Householder vector: 242 VGPRs on its own, which held every kernel that
// contained it at two waves per SIMD at best -- and the quad solver, whose own state lives next to it,
// at one.  A register file cannot be indexed per lane; LDS can.  Here
// the quad's matrix lives in a 1 KB block of LDS, column-major (column id c at doubles [8 (c-1),
// 8 c)), next to the multipliers x by column id:
//
//   * a lane still OWNS four columns (any assignment; the quad solver gives it the rows of E^-1 it
//     computed) and does everything "for each column" on them: it loads one column at a time
//     (4 x ds_read_b128), works on it, stores the rows that changed;
//   * reads whose column differs from problem to problem -- the chosen column of step five, the
//     columns of set P by position in the triangular solve, the pivot pair of a Givens step --
//     are plain LDS reads at a per-quad address, all four lanes the same one (broadcast);
//   * the permutation (position -> column id) is a 64-bit nibble vector replicated in every lane.
//
// Layout: quad g at doubles [g * NNLS_QUAD_STRIDE, ...).  The stride (130 doubles = 65 16-byte
// granules, odd) puts the 16 quads of a wave on distinct granules, and consecutive columns are 4
// granules apart, so a wave-wide ds_read_b128 of "my k-th column" (lanes of a quad on four
// consecutive columns) touches 64 distinct granules of every 256-byte row: conflict-free.
//
// Arithmetic per matrix element, its order, and every decision are those of
// oracle/optik_oracle.c:nnls: same bits.  Control flow around every cross-lane move and LDS
// hand-over is wave-uniform (tests/emu runs this file on the host).
#pragma once

#include <type_traits>

#include "ik_lane.hpp"
#include "ik_slsqp.hpp"

namespace optik {

// Geometry of a quad's block for an N-joint chain: 2N columns of CS doubles (m = N + 1 rows: eight
// doubles up to N = 7, ten -- 16-byte aligned -- for the nine rows of N = 8, whose row vectors are
// dvec16), then the 2N multipliers.  STRIDE: doubles per quad, an odd number of 16-byte granules.
template <int N>
struct NnlsQuadGeom {
    static constexpr int CS = (N + 1 <= 8) ? 8 : 10;
    static constexpr int XS = (N <= 7 ? 14 : 2 * N) * CS;                    // (one block geometry for every N <= 7)
    static constexpr int STRIDE = (N <= 7) ? 130 : ((XS + 2 * N + 1) / 2 * 2 + 2);  // N = 8: 160 + 16 -> 178 = 89 granules
    typedef typename RowVecOf<(N + 1 <= 8)>::type rowvec;
};
static_assert(NnlsQuadGeom<8>::STRIDE == 178, "N = 8: 89 granules per quad");
// doubles of LDS per wave: the blocks of its 16 quads, then sixteen doubles of +0.0 (read only)
template <int N>
constexpr int nnls_quad_wave_lds() { return 16 * NnlsQuadGeom<N>::STRIDE + 16; }

// -DOPTIK_PROFILE: wave cycles per part of the loop below, summed over all waves into g_quad_nnls_prof
// (phase_profile.py (a rounds 3-5 tool: git history)): 0 steps two-four, 1 step five, 2 steps six-ten, 3 step eleven, 4 loop trips,
// 5 calls, 6 Givens steps
#ifdef OPTIK_DEVICE_PROFILE
__device__ unsigned long long g_quad_nnls_prof[8];
// [0, 17): loop trips by the number of quads still solving; [17, 49): calls by loop trips; [49, 66): calls by quads taking part
__device__ unsigned long long g_quad_nnls_hist[66];
// wave cycles of the loop trips by the number of quads solving at the trip's start (0..16), then of the hand-overs [17]
__device__ unsigned long long g_quad_nnls_tripcyc[18];
#define QNNLS_PROBE(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); np_[slot] += now_ - nt_; nt_ = now_; } while (0)
#define QNNLS_COUNT(slot, n) np_[slot] += (n)
#else
#define QNNLS_PROBE(slot)
#define QNNLS_COUNT(slot, n)
#endif

template <int N>
OPTIK_DEV typename NnlsQuadGeom<N>::rowvec lds_col_load(const double *p) {
    const double *a = (const double *)__builtin_assume_aligned(p, 16);
    typename NnlsQuadGeom<N>::rowvec v = 0.0;
#pragma unroll
    for (int r = 0; r < NnlsQuadGeom<N>::CS; ++r) v[r] = a[r];
    return v;
}

template <int N>
OPTIK_DEV void lds_col_store(double *p, const typename NnlsQuadGeom<N>::rowvec v) {
    double *a = (double *)__builtin_assume_aligned(p, 16);
#pragma unroll
    for (int r = 0; r < NnlsQuadGeom<N>::CS; ++r) a[r] = v[r];
}

// Solves the quad's problem.  On entry the owners have written the columns to `blk` (column id c at
// blk + 8 (c - 1), rows 0 .. N) and an lds_sync() has made them visible; ids[k] = 1-based id of the
// lane's k-th column (> 2N: padding).  `live` = the quad has a problem.  On return xv[k] = the
// multiplier of the lane's k-th column; quad-uniform: mode (1 ok, 3 iteration cap), rnorm, and the
// number of solve passes.
//
// Pipe (ik_lane64.hpp): a wave with MORE problems than quads.  With a pipe the call starts with every quad idle and,
// whenever at most Pipe::MAX_RUNNING quads are still solving while problems wait (and when none is solving at all),
// calls pipe.event(idle, mode, rnorm, passes): the pipe takes the answers of the quads that have just finished,
// gives the idle quads their next problems -- it writes their blocks -- and returns whether the caller's quad got
// one; that quad starts over from step two while the others carry on where they were.  A problem's arithmetic does
// not depend on what the other quads are doing, so this only changes how many loop trips the wave runs: a straggler
// no longer holds fifteen finished quads until it is done.  The call returns when nothing runs and nothing waits.
struct NoPipe {
    static constexpr bool on = false;
    static constexpr bool warm_start = false;
    static constexpr int MAX_RUNNING = 0;
    OPTIK_DEV bool more() const { return false; }
    template <bool WARM>
    OPTIK_DEV bool event(bool, int, double, int, double *, int &) { return false; }
};

// Stop (ik_quad.hpp): launches with early exit.  stop->poll(running) is called once per loop trip; it says whether
// the quad's restart has been overtaken (another restart of its target succeeded) -- decided on a word it asked
// for one trip earlier, so the answer is there without a wait -- and the quad then leaves the loop at once: the
// caller abandons the restart, the multipliers are not used.
struct NoStop {
    static constexpr bool on = false;
    OPTIK_DEV bool poll(bool) { return false; }
};

// (Tried for the one-wave-per-SIMD kernel and dropped: the lane's four columns kept in registers as well, and a step's
// four column loads issued together -- both cost it more in spilled state than the exposed LDS latency they save:
// 27.0 -> 24.4 M and 28.2 -> 27.6 M restarts/s.)
template <int N, class Pipe = NoPipe, class Stop = NoStop>
OPTIK_DEV void nnls_quad(bool live, const int (&ids)[4], double *blk, const double *zeros, double (&xv)[4], int &mode_out,
                         double &rnorm_out, int &iters_out, Pipe *pipe = nullptr, Stop *stop = nullptr) {
    constexpr int m = N + 1, n = 2 * N;
    constexpr int CPL = 4;
    static_assert(m <= 9 && n <= 16, "row vectors hold up to sixteen entries, the permutation sixteen nibbles");
    constexpr int CS = NnlsQuadGeom<N>::CS;
    typedef typename NnlsQuadGeom<N>::rowvec dvecm;
    const double factor = 0.01;
    const int itmax = 3 * n;
    const int ql = quad_lane();
    double *const xs = blk + NnlsQuadGeom<N>::XS;
    // quad-uniform state (replicated in every lane of the quad)
    dvecm b = 0.0;
    b[m - 1] = 1.0;
    PackedIndex indx;
    indx.v = 0xFEDCBA9876543210ull;  // indx[pos] = pos
    int nsetp = 0, npp1 = 1, iter = 0, mode = 1;
    double up = 0.0;
    // per-column state of the lane's columns
    int pos[CPL];
    bool inZ[CPL], isc[CPL];
    double wv[CPL];
    double *colp[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        isc[k] = ids[k] <= n;
        pos[k] = ids[k];
        inZ[k] = isc[k];
        wv[k] = 0.0;
        xv[k] = 0.0;
        colp[k] = blk + CS * ((isc[k] ? ids[k] : 1) - 1);
        if (live && isc[k]) xs[ids[k] - 1] = 0.0;
    }
    int rem_jj = 0;  // step eleven: position being removed
    // phases: 0 = step two (recompute duals, then choose), 1 = step three (choose again),
    // 2 = step six (solve), 3 = step eleven (remove), 4 = done
    int phase = live ? 0 : 4;
    // rnorm = ||b(npp1..m)|| of the quad's problem as it stands
    auto residual_norm = [&]() {
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
                const double xsr = scale * b[r - 1];
                if (r >= k0 && r < k0 + cnt) sum += xsr * xsr;
            }
            rn = xmax * __builtin_sqrt(sum);
        }
        return rn;
    };
    lds_sync();
#ifdef OPTIK_DEVICE_PROFILE
    unsigned long long np_[8] = {0, 0, 0, 0, 0, 1, 0, 0};
    unsigned long long nt_ = __builtin_readcyclecounter();
#endif

#ifdef OPTIK_DEVICE_PROFILE
    int trips_ = 0;
    {
        const int n_live_ = __popcll(__ballot(live)) / 4;
        if ((threadIdx.x & 63u) == 0) atomicAdd(&g_quad_nnls_hist[49 + n_live_], 1ull);
    }
#endif
    // hand-over (Pipe): answers out of the quads that are done, the next problems into the idle ones.  WARM: the
    // problems may start after their first pass (ik_nnls_first.hpp) -- only the hand-over in front of the loop does
    // that, where no quad holds a problem yet and the registers its state takes inside the loop are free
    auto hand_over = [&](auto warm_tag) {
        constexpr bool WARM = decltype(warm_tag)::value;
        double wst[m + 2];
        int wj = 0;
        const bool fresh = pipe->template event<WARM>(phase == 4, mode, residual_norm(), iter, wst, wj);
        if (fresh) {
            // (the pipe has written the quad's block: the state of a problem at step two)
            b = 0.0;
            b[m - 1] = 1.0;
            indx.v = 0xFEDCBA9876543210ull;
            nsetp = 0; npp1 = 1; iter = 0; mode = 1;
            up = 0.0;
            rem_jj = 0;
            phase = 0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                pos[k] = ids[k];
                inZ[k] = isc[k];
                wv[k] = 0.0;
                xv[k] = 0.0;
                if (isc[k]) xs[ids[k] - 1] = 0.0;
            }
            if constexpr (WARM) {
                // Warm start: the owner lane's first pass already brought column wj in with a positive multiplier; the
                // pipe has written the block as that pass leaves it.  The state of the problem after steps five .. ten
                // of its first trip, as the code below would have left it:
                if (wj != 0) {
#pragma unroll
                    for (int r = 0; r < m; ++r) b[r] = wst[r];
                    // column wj took position 1, the column there (id 1) took wj's
                    indx.set(wj, indx.get(1));
                    indx.set(1, wj);
                    nsetp = 1; npp1 = 2; iter = 1;
                    up = wst[m];
#pragma unroll
                    for (int k = 0; k < CPL; ++k) {
                        const bool hit = ids[k] == wj;
                        const bool other = isc[k] && ids[k] == 1 && !hit;
                        pos[k] = hit ? 1 : (other ? wj : pos[k]);
                        inZ[k] = hit ? false : inZ[k];
                        xv[k] = hit ? wst[m + 1] : 0.0;
                        if (hit) xs[ids[k] - 1] = xv[k];
                    }
                }
            }
        }
        lds_sync();
    };
    if constexpr (Pipe::on && Pipe::warm_start) hand_over(std::true_type{});
    for (;;) {
        if constexpr (Pipe::on) {
            const int n_run = (int)__popcll(__ballot(phase < 4)) / QUAD;
            if (n_run == 0 || (n_run <= Pipe::MAX_RUNNING && pipe->more())) {
                hand_over(std::false_type{});
                if (!wave_any(phase < 4)) break;
            }
        } else {
            if constexpr (Stop::on) {
                if (stop->poll(phase < 4)) phase = 4;
            }
            if (!wave_any(phase < 4)) break;
        }
        QNNLS_COUNT(4, 1);
#ifdef OPTIK_DEVICE_PROFILE
        ++trips_;
        const int n_run_ = __popcll(__ballot(phase < 4)) / 4;
        if ((threadIdx.x & 63u) == 0) atomicAdd(&g_quad_nnls_hist[n_run_], 1ull);
        const unsigned long long trip_t0_ = __builtin_readcyclecounter();
#endif
        QNNLS_PROBE(7);
        // ---------------- steps two .. five --------------------------------------------
        if (wave_any(phase == 0 || phase == 1)) {
            const bool inA = (phase == 0 || phase == 1);
            if (inA && (nsetp + 1 > n || nsetp >= m)) phase = 4;  // iz1 > iz2 || nsetp >= m
            const bool run = (phase == 0 || phase == 1);
            if (wave_any(phase == 0)) {
                // step two: duals of the columns in Z over rows npp1 .. m
                dvecm bm = 0.0;
#pragma unroll
                for (int r = 1; r <= m; ++r) bm[r - 1] = (r >= npp1) ? b[r - 1] : 0.0;
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    const dvecm cv = lds_col_load<N>(colp[k]);
                    double sdot = 0.0;
#pragma unroll
                    for (int r = 1; r <= m; ++r) sdot += cv[r - 1] * bm[r - 1];
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
            for (int off = 2; off >= 1; off >>= 1) {
                const double ow = quad_xor(bw, off);
                const int op = quad_xor(bp, off);
                const bool better = (ow > bw) || (ow == bw && op < bp);
                bw = better ? ow : bw;
                bp = better ? op : bp;
            }
            const bool none = !(bw > 0.0);
            if (run && none) phase = 4;  // step four: every dual <= 0 -> done
            const bool cand = run && !none;
            QNNLS_PROBE(0);
            // (the last trip of most waves: every problem left has just finished)
            if (wave_any(cand)) {
                // step five: Householder construction on the chosen column j (position bp)
                const int j = cand ? indx.get(bp) : 1;
                // (w: the column, soon the transformation's weights -- u below the pivot row, up on it, 0 above)
                dvecm w = lds_col_load<N>(blk + CS * (j - 1));
                lds_sync();  // (every lane has the column before its pivot rows are rewritten below)
                bool hitk[CPL];
#pragma unroll
                for (int k = 0; k < CPL; ++k) hitk[k] = cand && inZ[k] && pos[k] == bp;
                // (all lanes run the construction; only `cand` quads keep its results)
                const double asave = blk[CS * (j - 1) + (npp1 <= m ? npp1 : m) - 1];  // w[npp1 - 1]: one read instead of a select chain
                const bool h12_live = npp1 < m;
                // the part of the column above the pivot row only feeds Lawson-Hanson's independence test (below)
                double xmax = 0.0;
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    const double av = (r <= nsetp) ? __builtin_fabs(w[r - 1]) : 0.0;
                    xmax = (av > xmax) ? av : xmax;
                }
#pragma unroll
                for (int r = 1; r <= m; ++r) w[r - 1] = (r > npp1) ? w[r - 1] : 0.0;
                double cl = __builtin_fabs(asave);
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    const double sm = __builtin_fabs(w[r - 1]);
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
                        d = w[r - 1] * clinv;
                        sm += d * d;
                    }
                    double c2 = cl * __builtin_sqrt(sm);
                    c2 = (asave > 0.0) ? -c2 : c2;
                    up = (cand && pivot) ? asave - c2 : up;
                    ulp = pivot ? c2 : ulp;
                }
                // diff(unorm + factor |ulp|, unorm) > 0 (decided by t > 3 * 2^-52 * xmax whenever
                // that holds; otherwise by the norm of the column above the pivot row, re-read from the block)
                const double t = factor * __builtin_fabs(ulp);
                bool ok1 = t > 6.7e-16 * xmax;
                if (wave_any(cand && !ok1)) {
                    const dvecm uu = lds_col_load<N>(blk + CS * (j - 1));
                    const double scale = 1.0 / xmax;
                    double sum = 0.0;
#pragma unroll
                    for (int r = 1; r <= m; ++r) {
                        const double xsr = scale * ((r <= nsetp) ? uu[r - 1] : 0.0);
                        sum += xsr * xsr;
                    }
                    const double unorm = (xmax != 0.0) ? xmax * __builtin_sqrt(sum) : 0.0;
                    const double d1 = unorm + t;
                    ok1 = d1 - unorm > 0.0;
                }
                const double hprod = up * ulp;
                const bool apply_live = cand && h12_live && !(__builtin_fabs(ulp) <= 0.0) && !(hprod >= 0.0);
                const double hb = apply_live ? 1.0 / hprod : 0.0;
#pragma unroll
                for (int r = 1; r <= m; ++r) w[r - 1] = (r == npp1) ? up : w[r - 1];
                // b := Q b: first the scalar and the pivot entry (they decide whether the column enters), then
                // b itself, in place, only if it does
                double smb = 0.0;
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    const double pr = b[r - 1] * w[r - 1];
                    smb = (r == 1) ? pr : smb + pr;
                }
                const bool actb = apply_live && ok1 && smb != 0.0;
                const double smhb = actb ? smb * hb : 0.0;
                const double bpiv = vpick(b, npp1);
                const double ztp = actb ? bpiv + smhb * up : bpiv;
                // Lawson-Hanson's test z(npp1) / A(npp1, j) > 0.  The quotient of two finite numbers is positive
                // exactly when they have the same sign and it does not underflow to zero: with both magnitudes
                // in [2^-500, 2^500] it cannot, so the signs decide; anything else takes the division itself.
                const double aztp = __builtin_fabs(ztp), aulp = __builtin_fabs(ulp);
                const bool tame = aztp >= 0x1p-500 && aztp <= 0x1p500 && aulp >= 0x1p-500 && aulp <= 0x1p500;
                bool quo_pos = (ztp > 0.0) == (ulp > 0.0);
                if (wave_any(cand && ok1 && !tame)) quo_pos = tame ? quo_pos : (ztp / ulp > 0.0);
                const bool found = cand && ok1 && quo_pos;
                // Rows the transformation leaves alone (above the pivot row) get -0.0 added: their weight is
                // an exact zero, the product a signed zero, and with the sign bit forced x + (-0.0) == x bit for
                // bit -- one OR per element instead of a select.  The same OR neutralises a whole vector.
                unsigned rowkeep[m];
#pragma unroll
                for (int r = 1; r <= m; ++r) rowkeep[r - 1] = (r < npp1) ? 0x80000000u : 0u;
                {
                    // (a column the test above rejects leaves b as it was -- oracle: z is only copied back to b when the
                    // column is found.  A select, not a zeroed factor: 0 * w would still be NaN for a column that has
                    // overflowed, and the oracle does not touch b then)
                    const bool moves = found && actb;
#pragma unroll
                    for (int r = 1; r <= m; ++r) {
                        const double add = smhb * w[r - 1];
                        const double addz = __hiloint2double((int)((unsigned)__double2hiint(add) | rowkeep[r - 1]),
                                                             __double2loint(add));
                        const double nb = b[r - 1] + addz;
                        b[r - 1] = moves ? nb : b[r - 1];
                    }
                }
                // column j takes position iz1 = nsetp + 1, the column there takes j's
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
                    if (found) {
                        indx.set(bp, indx.get(iz1));
                        indx.set(iz1, j);
                    }
                }
                nsetp = found ? npp1 : nsetp;
                npp1 = nsetp + 1;
                // the column that entered P: untouched above the pivot row, ulp on it, zeros below
                {
                    dvecm nc = lds_col_load<N>(blk + CS * (j - 1));
#pragma unroll
                    for (int r = 1; r <= m; ++r) nc[r - 1] = (r < nsetp) ? nc[r - 1] : ((r == nsetp) ? ulp : 0.0);
                    lds_sync();  // (every lane has re-read it)
                    if (found && ql == 0) lds_col_store<N>(blk + CS * (j - 1), nc);
                }
                // the transformation applied to the lane's columns still in Z (pivot row nsetp, rows below)
                // (asking for the lane's next column before this one is stored -- the compiler cannot know that they are
                // distinct blocks of LDS -- measured nothing: 32.69 against 32.74 M restarts/s, profiles/r5k_ab_pref.txt)
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    dvecm cv = lds_col_load<N>(colp[k]);
                    double sm = 0.0;
#pragma unroll
                    for (int r = 1; r <= m; ++r) {
                        const double pr = cv[r - 1] * w[r - 1];
                        sm = (r == 1) ? pr : sm + pr;
                    }
                    const bool act = found && apply_live && inZ[k] && sm != 0.0;
                    const double smh = act ? sm * hb : 0.0;
#pragma unroll
                    for (int r = 1; r <= m; ++r) {
                        const double add = smh * w[r - 1];
                        const double addz = __hiloint2double((int)((unsigned)__double2hiint(add) | rowkeep[r - 1]),
                                                             __double2loint(add));
                        cv[r - 1] = cv[r - 1] + addz;
                    }
                    if (act) lds_col_store<N>(colp[k], cv);
                    wv[k] = (cand && hitk[k]) ? 0.0 : wv[k];
                }
                // found: solve (step six); else choose again without recomputing the duals
                phase = cand ? (found ? 2 : 1) : phase;
                lds_sync();
                QNNLS_PROBE(1);
            }
        }
        // ---------------- steps six .. ten ---------------------------------------------
        if (wave_any(phase == 2)) {
            const bool run = phase == 2;
            dvecm zz = b;  // (steps five / eleven leave z := b)
            int nmax = 0;
#pragma unroll
            for (int v = 1; v <= m; ++v)
                if (wave_any(run && nsetp >= v)) nmax = v;
            // step six: solve the triangular system on set P (positions nsetp .. 1); the column at position ip
            // is read from the block.  A quad whose set P ends below ip reads the zeros instead and
            // subtracts 0 * 0 = +0, which changes nothing (x - (+0) == x bit for bit).
#pragma unroll
            for (int ip = m; ip >= 1; --ip) {
                if (ip > nmax) continue;
                const bool step = run && ip <= nsetp;
                const double *cp = step ? blk + CS * (indx.get(ip) - 1) : zeros;
                dvecm cv = 0.0;
#pragma unroll
                for (int r = 1; r <= ip; ++r) cv[r - 1] = cp[r - 1];
                const double zi = zz[ip - 1] / cv[ip - 1];
                const double zie = step ? zi : 0.0;
                zz[ip - 1] = step ? zi : zz[ip - 1];
                // (the solution component travels to the column's owner through the column's last double: padding
                // for m < CS, and for m == CS the last row -- an exact +0.0 in every column of set P, which no step
                // reads while the column is there; restored when the column leaves)
                if (step && ql == 0) const_cast<double *>(cp)[CS - 1] = zi;
#pragma unroll
                for (int r = 1; r < ip; ++r) zz[r - 1] = zz[r - 1] - zie * cv[r - 1];
            }
            if (run) {
                ++iter;
                if (iter > itmax) { mode = 3; phase = 4; }
            }
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
                const double xp = xs[(step ? indx.get(ip) : 1) - 1];
                const double zp = zz[ip - 1];
                const double tq = -xp / (zp - xp);
                const bool take = step && !(zp > 0.0) && !(alpha < tq);
                alpha = take ? tq : alpha;
                jj = take ? ip : jj;
            }
            lds_sync();  // (the scan has read the multipliers: now their owners may rewrite them)
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const bool mine = go && isc[k] && !inZ[k];
                const double zown = colp[k][CS - 1];  // z at the column's position (written in step six)
                const double nx = (1.0 - alpha) * xv[k] + alpha * zown;
                xv[k] = mine ? nx : xv[k];
                if (mine) xs[ids[k] - 1] = nx;
            }
            rem_jj = (go && jj != 0) ? jj : rem_jj;
            phase = go ? (jj == 0 ? 0 : 3) : phase;  // back to step two, or remove position jj
            lds_sync();
            QNNLS_PROBE(2);
        }
        // ---------------- step eleven ----------------------------------------------------
        if (wave_any(phase == 3)) {
            const bool run = phase == 3;
            // move the coefficient at position rem_jj from set P to set Z
            bool leaving[CPL];
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                leaving[k] = run && isc[k] && !inZ[k] && pos[k] == rem_jj;
                if (leaving[k]) { xv[k] = 0.0; xs[ids[k] - 1] = 0.0; colp[k][CS - 1] = 0.0; }
            }
            const int id_out = run ? indx.get(rem_jj) : 1;
            const int jlo = run ? rem_jj + 1 : 0x7fffffff, jhi = run ? nsetp : 0;
            // (the positions some quad of the wave has to move: m ballots instead of a wave-wide min / max butterfly)
            int wlo = 0x7fffffff, whi = 0;
            for (int v = 2; v <= m; ++v) {
                if (wave_any(run && v >= jlo && v <= jhi)) {
                    wlo = wlo < v ? wlo : v;
                    whi = v;
                }
            }
            for (int j = wlo; j <= whi; ++j) {
                QNNLS_COUNT(6, 1);
                const bool step = run && j >= jlo && j <= jhi;
                const int jm1 = j - 1 < 1 ? 1 : (j - 1 > m ? m : j - 1), jc = j > m ? m : (j < 1 ? 1 : j);
                // the column at position j moves to position j-1; Givens on its rows j-1, j
                const int ii = step ? indx.get(jc) : 1;
                double a0 = blk[CS * (ii - 1) + jm1 - 1], a1 = blk[CS * (ii - 1) + jc - 1];
                lds_sync();  // (the pivot pair is read before the owners rewrite rows j-1, j)
                double c = 1.0, s = 0.0;
                rotg(a0, a1, c, s);
                const double t = a0;
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    if (!(step && isc[k])) continue;
                    const bool is_ii = ids[k] == ii;
                    double *p0 = colp[k] + (jm1 - 1), *p1 = colp[k] + (jc - 1);
                    const double xi = *p0, yi = *p1;
                    const double nx = c * xi + s * yi;
                    const double ny = c * yi - s * xi;
                    *p0 = is_ii ? t : nx;
                    *p1 = is_ii ? 0.0 : ny;
                    if (is_ii) pos[k] = j - 1;
                }
                if (step) {
                    indx.set(jm1, ii);
                    const double bx = vpick(b, j - 1), by = vpick(b, j);
                    vput(b, j - 1, c * bx + s * by);
                    vput(b, j, c * by - s * bx);
                }
                lds_sync();
            }
            if (run) {
                npp1 = nsetp;
                --nsetp;
#pragma unroll
                for (int k = 0; k < CPL; ++k)
                    if (leaving[k]) { pos[k] = nsetp + 1; inZ[k] = true; }  // --iz1; indx[iz1] = i
                indx.set(nsetp + 1 < 1 ? 1 : nsetp + 1, id_out);
                if (nsetp <= 0) { mode = 3; phase = 4; }
            }
            {
                // is every coefficient left in P feasible?  first offending position, in order
                int bad = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    const int p = (isc[k] && !inZ[k] && xv[k] <= 0.0) ? pos[k] : 0x7fffffff;
                    bad = p < bad ? p : bad;
                }
#pragma unroll
                for (int off = 2; off >= 1; off >>= 1) { const int o = quad_xor(bad, off); bad = o < bad ? o : bad; }
                if (phase == 3) {
                    if (bad != 0x7fffffff) rem_jj = bad;  // again
                    else phase = 2;
                }
            }
            lds_sync();
            QNNLS_PROBE(3);
        }
#ifdef OPTIK_DEVICE_PROFILE
        if ((threadIdx.x & 63u) == 0) atomicAdd(&g_quad_nnls_tripcyc[n_run_], __builtin_readcyclecounter() - trip_t0_);
#endif
    }
#ifdef OPTIK_DEVICE_PROFILE
    if ((threadIdx.x & 63u) == 0) {
        for (int i_ = 0; i_ < 7; ++i_) atomicAdd(&g_quad_nnls_prof[i_], np_[i_]);
        atomicAdd(&g_quad_nnls_hist[17 + (trips_ < 31 ? trips_ : 31)], 1ull);
    }
#endif
    rnorm_out = residual_norm();
    mode_out = mode;
    iters_out = iter;
}

}  // namespace optik
