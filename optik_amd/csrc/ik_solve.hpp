// ik_solve.hpp -- one random restart per lane: seed, SLSQP state machine, NLopt
// stopping rules, classification.
//
// Restates the closure of /root/reference/crates/optik/src/lib.rs:301-391 and
// the nlopt_slsqp() driver it calls (un-vendored).  Instead
// of NLopt's reverse-communication call/return, every lane runs the same loop
//
//     evaluate f, g at x  ->  bookkeeping / stop tests  ->  (line search accepted:
//     BFGS update + next search direction)  ->  next trial point
//
// so the whole wave executes one objective evaluation per trip regardless of
// which restart is in which SLSQP phase; only the accept / reject branch
// diverges.  Numerics and decisions are bit-identical to the oracle's
// ok_solve_restart().
#pragma once

#include "ik_slsqp.hpp"

namespace optik {

// nlopt_result values / lib.rs:376-379 classification (same codes as the oracle).
enum : int32_t {
    RES_FAILURE = -1,
    RES_ROUNDOFF_LIMITED = -4,
    RES_FORCED_STOP = -5,
    RES_ITER_CAP = -100,
    RES_STOPVAL_REACHED = 2,
    RES_FTOL_REACHED = 3,
    RES_XTOL_REACHED = 4,
};

constexpr int MAX_EVALS_CAP = 100000;  // same safety cap as the oracle

// Wave-uniform solver parameters (derived from SolverConfig on the host).
struct SolveParams {
    double stopval;   // tol_f            (set_stopval,  lib.rs:345)
    double ftol_abs;  // tol_df heuristic (set_ftol_abs, lib.rs:283-293, 346)
    double xtol_abs;  // tol_dx           (set_xtol_abs1, lib.rs:347)
    int32_t ok_stopval, ok_ftol, ok_xtol;  // lib.rs:376-379: tol_* >= 0
    int32_t pad;
};

// ---- RNG: ChaCha8Rng::seed_from_u64(42), set_stream(i)  (lib.rs:358-362) -----

OPTIK_DEV uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

#define OPTIK_QR(a, b, c, d)              \
    a += b; d ^= a; d = rotl32(d, 16);    \
    c += d; b ^= c; b = rotl32(b, 12);    \
    a += b; d ^= a; d = rotl32(d, 8);     \
    c += d; b ^= c; b = rotl32(b, 7);

// One ChaCha8 block (DJB layout: words 12-13 block counter, 14-15 stream id).
OPTIK_DEV void chacha8_block(const uint32_t (&key)[8], uint64_t counter, uint64_t stream,
                             uint32_t (&out)[16]) {
    uint32_t s[16], x[16];
    s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
#pragma unroll
    for (int i = 0; i < 8; ++i) s[4 + i] = key[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32);
    s[14] = (uint32_t)stream;  s[15] = (uint32_t)(stream >> 32);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = s[i];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        OPTIK_QR(x[0], x[4], x[8], x[12]) OPTIK_QR(x[1], x[5], x[9], x[13])
        OPTIK_QR(x[2], x[6], x[10], x[14]) OPTIK_QR(x[3], x[7], x[11], x[15])
        OPTIK_QR(x[0], x[5], x[10], x[15]) OPTIK_QR(x[1], x[6], x[11], x[12])
        OPTIK_QR(x[2], x[7], x[8], x[13]) OPTIK_QR(x[3], x[4], x[9], x[14])
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

// rand 0.9 UniformFloat<f64>::new_inclusive(lo, hi).sample(): 52 mantissa bits.
// `scale` is precomputed on the host (it depends on the limits only).
OPTIK_DEV double uniform_inclusive(double lo, double scale, uint64_t bits) {
    const uint64_t m = (bits >> 12) | 0x3ff0000000000000ull;
    const double value0_1 = __longlong_as_double((long long)m) - 1.0;
    return value0_1 * scale + lo;
}

// Random configuration of restart `index` (lib.rs:86-91, 366-370); N <= 8 uses
// only keystream block 0.
template <int N>
OPTIK_DEV void restart_seed(const uint32_t (&key)[8], const double *lb, const double *scale,
                            uint64_t index, double (&q)[N]) {
    static_assert(N <= 8, "one ChaCha block per restart");
    uint32_t blk[16];
    chacha8_block(key, 0, index, blk);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const uint64_t bits = (uint64_t)blk[2 * k] | ((uint64_t)blk[2 * k + 1] << 32);
        q[k] = uniform_inclusive(lb[k], scale[k], bits);
    }
}

// ---- per-restart result ------------------------------------------------------

template <int N>
struct RestartOut {
    double x[N];   // best point (NLopt returns the best-so-far x)
    double f;      // minf
    int32_t result;
    int32_t success;
    int32_t n_evals;
    int32_t n_iters;
};

// Control words shared by a launch (HBM).  first_success implements the
// reference's should_exit flag (lib.rs:269, 308, 382-384) in its deterministic
// reading: a restart is abandoned only if a LOWER index already succeeded.
struct AbortCtl {
    const unsigned long long *first_success;  // per target, or nullptr
    unsigned long long deadline;               // wall_clock64() ticks, 0 = none
};

OPTIK_DEV bool wave_any(bool p) { return __ballot(p) != 0ull; }

// Runs one restart per lane to termination.  `active` lanes hold a restart; the
// others idle through the loop.  x holds the seed on entry.
template <int N, bool TIP>
OPTIK_DEV void solve_restart(const ChainDev &ch, const EvalParams &ep, const SolveParams &sp,
                             const Pose target, const NnlsWs<N> &ws, bool active, double (&x)[N],
                             uint64_t restart_index, const AbortCtl ctl, unsigned target_slot,
                             RestartOut<N> &out) {
    constexpr int NL = N * (N + 1) / 2;
    const double alfmin = 0.1;
    double x0[N], g[N], s[N], v[N], l[NL];
    double xbest[N], xprev[N];
    double f = 0.0, f0 = 0.0, t0 = 0.0, h3 = 0.0, alpha = 1.0;
    double minf = __builtin_huge_val(), fprev = __builtin_huge_val();
    int iter = 0, ireset = 0, line = 0, nevals = 0;
    int32_t ret = 0;
    bool first = true;
#pragma unroll
    for (int i = 0; i < N; ++i) { xbest[i] = x[i]; xprev[i] = x[i]; x0[i] = x[i]; s[i] = 0.0; v[i] = 0.0; g[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < NL; ++i) l[i] = 0.0;

    while (wave_any(active)) {
        if (active) {
            // lib.rs:308: abandon when timed out or a lower-index restart succeeded
            bool stop = false;
            if (ctl.first_success) {
                const unsigned long long fs =
                    __hip_atomic_load(ctl.first_success + target_slot, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
                stop = fs < restart_index;
            }
            if (ctl.deadline && wall_clock64() > ctl.deadline) stop = true;
            if (stop) { ret = RES_FORCED_STOP; active = false; }
        }
        double gn[N];
        double fn = 0.0;
        OPTIK_SCHED_FENCE();
        if (active) fn = eval_fg<N, TIP>(ch, ep, target, x, gn);
        OPTIK_SCHED_FENCE();
        if (active) {
            f = fn;
            ++nevals;
            // NLopt: update best point so far; stopval is tested after every evaluation
            if (f < minf) {
                minf = f;
#pragma unroll
                for (int i = 0; i < N; ++i) xbest[i] = x[i];
            }
            bool need_dir = false, reset = false;
            if (minf < sp.stopval) {
                ret = RES_STOPVAL_REACHED;
            } else if (nevals >= MAX_EVALS_CAP) {
                ret = RES_ITER_CAP;
            } else if (first) {
                // SLSQPB label 100/110: initialise, reset the BFGS matrix
                first = false;
#pragma unroll
                for (int i = 0; i < N; ++i) g[i] = gn[i];
                need_dir = true;
                reset = true;
            } else {
                // label 220: L1 merit (m = 0: the objective itself)
                const double h1 = f - t0;
                bool accept = false;
                if (__builtin_isfinite(h1)) {
                    if (h1 <= h3 / 10.0 || line > 10) accept = true;
                    else {
                        const double a = h3 / ((h3 - h1) * 2.0);
                        alpha = (a > alfmin) ? a : alfmin;
                    }
                } else {
                    const double a = alpha * 0.5;
                    alpha = (a > alfmin) ? a : alfmin;
                }
                if (accept) {
                    // line search complete (mode -1): NLopt re-evaluates f and the
                    // gradient there unless the accepted trial was the first one
                    if (line > 1) ++nevals;
                    if (!__builtin_isinf(fprev)) {
                        if (__builtin_fabs(f - fprev) < sp.ftol_abs) ret = RES_FTOL_REACHED;
                        else {
                            bool allx = true;
#pragma unroll
                            for (int i = 0; i < N; ++i)
                                allx = allx && !(__builtin_fabs(x[i] - xprev[i]) >= sp.xtol_abs);
                            if (allx) ret = RES_XTOL_REACHED;
                        }
                    }
                    fprev = f;
#pragma unroll
                    for (int i = 0; i < N; ++i) xprev[i] = x[i];
                    if (ret == 0 && nevals >= MAX_EVALS_CAP) ret = RES_ITER_CAP;
                    if (ret == 0) {
                        // label 260: BFGS update with u = g_new - g_old
                        double u[N];
#pragma unroll
                        for (int i = 0; i < N; ++i) { u[i] = gn[i] - v[i]; g[i] = gn[i]; }
                        OPTIK_SCHED_FENCE();
                        bfgs_update<N>(l, s, u);
                        OPTIK_SCHED_FENCE();
                        need_dir = true;
                    }
                }
            }
            // labels 110/130: (reset,) search direction, descent test
            while (need_dir) {
                if (reset) {
                    ++ireset;
                    if (ireset > 5) {
                        // label 255 with acc = 0 -> mode 8; NLopt's relaxed test vs (f0, x0)
                        ret = RES_ROUNDOFF_LIMITED;
                        if (__builtin_fabs(f - f0) < sp.ftol_abs && !__builtin_isinf(f0)) ret = RES_FTOL_REACHED;
                        else {
                            bool allx = true;
#pragma unroll
                            for (int i = 0; i < N; ++i)
                                allx = allx && !(__builtin_fabs(x[i] - x0[i]) >= sp.xtol_abs);
                            if (allx) ret = RES_XTOL_REACHED;
                        }
                        break;
                    }
#pragma unroll
                    for (int i = 0; i < NL; ++i) l[i] = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) l[lidx<N>(i, i)] = 1.0;
                }
                ++iter;
                double lo[N], hi[N];
#pragma unroll
                for (int i = 0; i < N; ++i) { lo[i] = ch.lb[i] - x[i]; hi[i] = ch.ub[i] - x[i]; }
                OPTIK_SCHED_FENCE();
                const int lmode = lsq_box<N>(ws, l, g, lo, hi, s);
                OPTIK_SCHED_FENCE();
                if (lmode != 1) {
                    // NLopt: modes 5,6,7 -> ROUNDOFF_LIMITED; 3,4,9 -> FAILURE
                    ret = (lmode == 5 || lmode == 6 || lmode == 7) ? RES_ROUNDOFF_LIMITED : RES_FAILURE;
                    break;
                }
                double gs = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) { v[i] = g[i]; x0[i] = x[i]; }
                f0 = f;
#pragma unroll
                for (int i = 0; i < N; ++i) gs += g[i] * s[i];
                t0 = f;
                h3 = gs;  // h3 = gs - h1 * h4 with h1 = 0 (no constraints)
                if (h3 >= 0.0) { reset = true; continue; }
                line = 0;
                alpha = 1.0;
                break;
            }
            if (ret == 0) {
                // label 190: next trial point x = x0 + alpha * s, clipped (NLopt)
                ++line;
                h3 = alpha * h3;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    s[i] *= alpha;
                    double xi = x0[i];
                    xi += s[i];
                    if (xi < ch.lb[i]) xi = ch.lb[i];
                    else if (xi > ch.ub[i]) xi = ch.ub[i];
                    x[i] = xi;
                }
            } else {
                active = false;
            }
        }
    }

#pragma unroll
    for (int i = 0; i < N; ++i) out.x[i] = xbest[i];
    out.f = minf;
    out.result = ret;
    out.success = (sp.ok_stopval && ret == RES_STOPVAL_REACHED) || (sp.ok_ftol && ret == RES_FTOL_REACHED)
                  || (sp.ok_xtol && ret == RES_XTOL_REACHED);
    out.n_evals = nevals;
    out.n_iters = iter;
}

}  // namespace optik
