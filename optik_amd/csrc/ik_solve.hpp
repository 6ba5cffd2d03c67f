// ik_solve.hpp -- what every single-launch solver shares: NLopt's result codes and stopping rules, the restart seeds
// (ChaCha8), the work queue of a launch.
//
// Restates pieces of the closure of /root/reference/crates/optik/src/lib.rs:301-391 and of the nlopt_slsqp()
// driver it calls (un-vendored).  Instead of NLopt's reverse-communication call/return, every
// solver runs the same loop
//
//     evaluate f, g at x  ->  bookkeeping / stop tests  ->  (line search accepted:
//     BFGS update + next search direction)  ->  next trial point
//
// so a wave executes one objective evaluation per trip regardless of which restart is in which SLSQP phase
// (ik_lane64.hpp: a restart per lane; ik_quad.hpp: a restart per quad of lanes).  Numerics and decisions are
// bit-identical to the oracle's ok_solve_restart().
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

// Optional per-wave phase timers (build with -DOPTIK_PROFILE; phase_profile.py (a rounds 3-5 tool: git history)).
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

}  // namespace optik
