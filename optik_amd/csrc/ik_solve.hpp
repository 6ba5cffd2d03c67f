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
constexpr unsigned REFILL_BATCH = 8;   // idle lanes a wave accumulates before it refills

// Optional per-wave phase timers (build with -DOPTIK_PROFILE; tools/phase_profile.py).
#ifdef OPTIK_PROFILE
#define OPTIK_PROF_DECL unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long prof_t_ = 0
#define OPTIK_PROF_BEGIN() prof_t_ = __builtin_readcyclecounter()
#define OPTIK_PROF_END(slot) prof_[slot] += __builtin_readcyclecounter() - prof_t_
#define OPTIK_PROF_COUNT(slot, n) prof_[slot] += (n)
#define OPTIK_PROF_SUB_BEGIN() const unsigned long long prof_s_ = __builtin_readcyclecounter()
#define OPTIK_PROF_SUB_END(slot) prof_[slot] += __builtin_readcyclecounter() - prof_s_
#define OPTIK_PROF_FLUSH(ptr)                                                          \
    if ((ptr) && (threadIdx.x & 63u) == 0) {                                           \
        for (int i_ = 0; i_ < 8; ++i_) atomicAdd((ptr) + i_, prof_[i_]);               \
    }
#else
#define OPTIK_PROF_DECL
#define OPTIK_PROF_BEGIN()
#define OPTIK_PROF_END(slot)
#define OPTIK_PROF_COUNT(slot, n)
#define OPTIK_PROF_SUB_BEGIN()
#define OPTIK_PROF_SUB_END(slot)
#define OPTIK_PROF_FLUSH(ptr)
#endif

// Wave-uniform solver parameters (derived from SolverConfig on the host).
struct SolveParams {
    double stopval;   // tol_f            (set_stopval,  lib.rs:345)
    double ftol_abs;  // tol_df heuristic (set_ftol_abs, lib.rs:283-293, 346)
    double xtol_abs;  // tol_dx           (set_xtol_abs1, lib.rs:347)
    int32_t ok_stopval, ok_ftol, ok_xtol;  // lib.rs:376-379: tol_* >= 0
    // nlopt_stop_x of NLopt >= 2.6.2 (what rust-nlopt 0.8 bundles: 2.7.1) returns 1 first of all
    // when ||x - oldx|| <= xtol_rel * ||x||, i.e. with xtol_rel = 0 when the step is exactly
    // zero; NLopt 2.5 only has the per-coordinate xtol_abs test.  1 = the newer rule (default).
    // It can only matter when ftol_abs <= 0: a zero step leaves f unchanged and the ftol
    // test, which comes first, fires on |f - fprev| = 0 < ftol_abs.
    int32_t stop_x_zero;
};

// nlopt_stop_x(x, oldx) with xtol_rel = 0 and xtol_abs[i] = tol_dx (nlopt/src/util/stop.c).
template <int N>
OPTIK_DEV bool stop_x(const SolveParams &sp, const double (&x)[N], const double (&oldx)[N]) {
    bool zero = sp.stop_x_zero != 0, allx = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        zero = zero && (x[i] == oldx[i]);
        allx = allx && !(__builtin_fabs(x[i] - oldx[i]) >= sp.xtol_abs);
    }
    return zero || allx;
}
// the previous iterate is only needed when one of the two tests can fire after the ftol test
OPTIK_DEV bool xprev_live(const SolveParams &sp) { return sp.xtol_abs >= 0.0 || (sp.stop_x_zero && !(sp.ftol_abs > 0.0)); }

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

// ---- work queue, per-restart results -----------------------------------------

// Everything a launch shares (kernel arguments -> SGPRs).  Work items are the
// T * R (target, restart) pairs, item w = t * R + (index - restart_begin); lanes
// pull items from one global counter, so a lane that finishes a restart starts
// the next one immediately and a wave never waits for its slowest restart (the
// GPU analogue of rayon's work stealing, lib.rs:297-300).  Results are keyed by
// the item, so which lane ran it is irrelevant.
struct WorkQueue {
    unsigned long long *next_item;       // global counter, zeroed before the launch
    unsigned long long total_items;      // T * R
    unsigned long long n_restarts;       // R (per target)
    unsigned long long restart_begin;
    const double *targets;               // [T][7]
    const double *x0;                    // [T][n]
    // first_success implements the reference's should_exit flag (lib.rs:269, 308,
    // 382-384) in its deterministic reading: a restart is abandoned only if a
    // LOWER index of the same target already succeeded.
    unsigned long long *first_success;   // [T] or nullptr
    // 1: the reference's multi-thread reading of should_exit (find_any, lib.rs:409-412): ANY
    // success of the target makes every other restart of it give up at its next evaluation --
    // the answer is then whichever restart got there first (a valid solution, not a fixed one)
    int find_any;
    // 1: work items are restart-major (item = r * T + t: every target's low restart indices
    // first) instead of target-major -- with early exit most higher indices are then never
    // started (cooperative kernel only; the output column stays t * R + r)
    int restart_major;
    unsigned long long n_targets;        // T
    unsigned long long deadline;         // wall_clock64() ticks, 0 = none
    int quality;                         // selection key: 1 = ||x - x0||_2, 0 = index
    int lanes;                           // lanes of a wave that take work items (1 .. 64): a launch too small to
                                         // fill the chip spreads out, so that a wave does not pay for the phases
                                         // and NNLS pass counts of 63 other restarts (single-call latency)
    double *out_x;                       // [n][T*R]  best point (NLopt returns best-so-far x)
    double *out_f;                       // [T*R]     minf
    double *out_key;                     // [T*R]     selection key, +inf unless success
    int32_t *out_status;                 // [T*R]
    int32_t *out_evals;                  // [T*R]
    unsigned long long *prof;            // [8] phase cycle totals (OPTIK_PROFILE builds), else null
    // find_any launches of ONE target (a single Robot::ik call): the first restart to succeed also writes its answer
    // to this host-visible block and then stores claim_seq into its first word, so that the host can return while
    // the other restarts are still noticing the flag (quad solver; null: no such block).
    // Layout: [0] sequence word, [1] restart index, [2] f (as a double), [3 .. 3 + n) x
    unsigned long long *claim;
    unsigned long long claim_seq;
};

// Wave-aggregated fetch of one work item per requesting lane: one atomic per wave.
OPTIK_DEV unsigned long long fetch_items(unsigned long long *counter, bool want) {
    const unsigned long long mask = __ballot(want);
    const unsigned n = (unsigned)__popcll(mask);
    unsigned long long base = 0;
    if (n) {
        const int leader = __ffsll((long long)mask) - 1;
        const int lane = (int)(threadIdx.x & 63u);
        if (lane == leader) base = atomicAdd(counter, (unsigned long long)n);
        base = __shfl(base, leader, 64);
    }
    const unsigned rank = (unsigned)__popcll(mask & ((1ull << (threadIdx.x & 63u)) - 1ull));
    return base + rank;
}

// One 64-lane wave solving restarts until the queue is empty.
template <int N, bool TIP>
OPTIK_DEV void solve_wave(const ChainDev &ch, const EvalParams &ep, const SolveParams &sp,
                          const uint32_t (&key)[8], const double (&scale)[MAX_DOF], const WorkQueue &wq,
                          const NnlsWs<N> &ws) {
    constexpr int NL = N * (N + 1) / 2;
    const double alfmin = 0.1;
    // SLSQP state of the lane's current restart
    double x[N], x0[N], g[N], s[N], l[NL];
    double xbest[N], xprev[N];
    double f = 0.0, f0 = 0.0, t0 = 0.0, h3 = 0.0, alpha = 1.0;
    double minf = __builtin_huge_val(), fprev = __builtin_huge_val();
    int ireset = 0, line = 0, nevals = 0;  // (Kraft's iter only feeds maxiter, which NLopt leaves unbounded)
    bool first = true;
    // the work item
    Pose target;
    unsigned long long item = 0, index = 0;
    unsigned tslot = 0;
    bool active = false, want = (int)(threadIdx.x & 63u) < wq.lanes;
#pragma unroll
    for (int i = 0; i < N; ++i) { x[i] = 0.0; x0[i] = 0.0; g[i] = 0.0; s[i] = 0.0; xbest[i] = 0.0; xprev[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < NL; ++i) l[i] = 0.0;
    target.t = V3{0, 0, 0};
    target.q = Q4{0, 0, 0, 1};
    OPTIK_PROF_DECL;

    for (;;) {
        OPTIK_PROF_BEGIN();
        // ---- refill: lanes without a restart pull the next work item -------------
        // (the seed generation below runs for the whole wave, so wait until several lanes
        // are idle -- or none is busy -- before paying for it)
        const unsigned n_want = (unsigned)__popcll(__ballot(want));
        if (n_want >= (unsigned)(wq.lanes < REFILL_BATCH ? wq.lanes : REFILL_BATCH) || (n_want > 0 && !wave_any(active))) {
            const unsigned long long it = fetch_items(wq.next_item, want);
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
                    restart_seed<N>(key, ch.lb, scale, index, x);
                    if (index == 0) {
                        const double *x0p = wq.x0 + (size_t)tslot * N;
#pragma unroll
                        for (int i = 0; i < N; ++i) x[i] = x0p[i];
                    }
#pragma unroll
                    for (int i = 0; i < N; ++i) { xbest[i] = x[i]; xprev[i] = x[i]; x0[i] = x[i]; s[i] = 0.0; g[i] = 0.0; }
                    f = 0.0; f0 = 0.0; t0 = 0.0; h3 = 0.0; alpha = 1.0;
                    minf = __builtin_huge_val(); fprev = __builtin_huge_val();
                    ireset = 0; line = 0; nevals = 0;
                    first = true;
                    active = true;
                }
            }
        }
        OPTIK_PROF_END(0);  // refill
        if (!wave_any(active)) break;
        OPTIK_PROF_COUNT(7, 1);  // trips

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
            if (stop) ret = RES_FORCED_STOP;
        }
        double gn[N];
        double fn = 0.0;
        const bool do_eval = active && ret == 0;
        OPTIK_SCHED_FENCE();
        OPTIK_PROF_BEGIN();
        if (do_eval) fn = eval_fg<N, TIP>(ch, ep, target, x, gn);
        OPTIK_PROF_END(1);  // eval
        OPTIK_SCHED_FENCE();
        OPTIK_PROF_BEGIN();
        if (do_eval) {
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
                        else if (stop_x<N>(sp, x, xprev)) ret = RES_XTOL_REACHED;
                    }
                    fprev = f;
#pragma unroll
                    for (int i = 0; i < N; ++i) xprev[i] = x[i];
                    if (ret == 0 && nevals >= MAX_EVALS_CAP) ret = RES_ITER_CAP;
                    if (ret == 0) {
                        // label 260: BFGS update with u = g_new - g_old
                        double u[N];
#pragma unroll
                        for (int i = 0; i < N; ++i) { u[i] = gn[i] - g[i]; g[i] = gn[i]; }
                        OPTIK_SCHED_FENCE();
                        OPTIK_PROF_SUB_BEGIN();
                        bfgs_update<N>(l, s, u);
                        OPTIK_PROF_SUB_END(4);  // BFGS (inside slot 2)
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
                        else if (stop_x<N>(sp, x, x0)) ret = RES_XTOL_REACHED;
                        break;
                    }
#pragma unroll
                    for (int i = 0; i < NL; ++i) l[i] = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) l[lidx<N>(i, i)] = 1.0;
                }
                double lo[N], hi[N];
#pragma unroll
                for (int i = 0; i < N; ++i) { lo[i] = ch.lb[i] - x[i]; hi[i] = ch.ub[i] - x[i]; }
                OPTIK_SCHED_FENCE();
                OPTIK_PROF_SUB_BEGIN();
                unsigned long long nnls_cycles = 0;
                const int lmode = lsq_box<N>(ws, l, g, lo, hi, s, nnls_cycles);
                OPTIK_PROF_SUB_END(5);  // LSQ incl. NNLS (inside slot 2)
                OPTIK_PROF_COUNT(6, nnls_cycles);
                OPTIK_SCHED_FENCE();
                if (lmode != 1) {
                    // NLopt: modes 5,6,7 -> ROUNDOFF_LIMITED; 3,4,9 -> FAILURE
                    ret = (lmode == 5 || lmode == 6 || lmode == 7) ? RES_ROUNDOFF_LIMITED : RES_FAILURE;
                    break;
                }
                // (g is also Kraft's v: the gradient at the start of the line search)
                double gs = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) x0[i] = x[i];
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
            }
        }
        OPTIK_PROF_END(2);  // bookkeeping + BFGS + direction
        OPTIK_PROF_BEGIN();
        // ---- a restart ended: classify (lib.rs:376-379), publish, free the lane ----
        if (active && ret != 0) {
            const bool success = (sp.ok_stopval && ret == RES_STOPVAL_REACHED)
                                 || (sp.ok_ftol && ret == RES_FTOL_REACHED)
                                 || (sp.ok_xtol && ret == RES_XTOL_REACHED);
            if (wq.out_x) {
#pragma unroll
                for (int i = 0; i < N; ++i) wq.out_x[(size_t)i * wq.total_items + item] = xbest[i];
            }
            if (wq.out_f) wq.out_f[item] = minf;
            if (wq.out_status) wq.out_status[item] = ret;
            if (wq.out_evals) wq.out_evals[item] = nevals;
            // selection key (lib.rs:402-407): Quality = ||x - x0||_2, Speed = index
            double k = __builtin_huge_val();
            if (success) {
                if (wq.quality) {
                    const double *x0p = wq.x0 + (size_t)tslot * N;
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) { const double d = xbest[i] - x0p[i]; acc += d * d; }
                    k = __builtin_sqrt(acc);
                } else {
                    k = (double)index;
                    if (wq.first_success) atomicMin(wq.first_success + tslot, index);
                }
            }
            if (wq.out_key) wq.out_key[item] = k;
            active = false;
            want = true;
        }
        OPTIK_PROF_END(3);  // publish
    }
    OPTIK_PROF_FLUSH(wq.prof);
}

}  // namespace optik
