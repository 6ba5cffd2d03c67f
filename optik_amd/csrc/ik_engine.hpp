// ik_engine.hpp -- streaming ("continuous batching") form of the restart solver.
//
// The single-kernel solver (ik_solve.hpp) keeps a restart in one lane from seed to
// termination; a wave then pays for every SLSQP phase any of its 64 lanes is in
// (measured: 36 % VALU lane utilisation, 2 waves per CU because of the NNLS LDS).
// For throughput the same arithmetic is re-cut along the phases instead:
//
//   * a pool of C restart *slots* lives in HBM as struct-of-arrays planes, tiled by 64 slots
//     with neighbouring planes interleaved in pairs (ENG_D_AT below: plane k of slot s at
//     [(s/64) * ceil(ND/2) * 128 + (k/2) * 128 + (s%64) * 2 + k%2]): thread s of every per-slot
//     kernel owns slot s, so all state traffic is coalesced, mostly 16 bytes per lane;
//   * one *trip* = five kernels over a sub-pool, each doing one phase for the slots that
//     are in it:
//       eng_eval_kernel     objective + gradient at x, NLopt bookkeeping / stop tests,
//                           line-search accept/reject (rejected: next trial point)
//       eng_update_kernel   accepted: BFGS update, LSQ factor, the rows of the bounded
//                           dual problem streamed into the slot's record; direction when
//                           the step stays inside the box
//       eng_bucket_kernel   this trip's problems ordered by predicted pass count, slots
//                           to refill, in-use count
//       eng_nnls_coop_kernel  Lawson-Hanson NNLS, one problem per four lanes
//                           (ik_nnls_coop.hpp)
//       eng_finish_kernel   refill of finished slots from the work queue; LDP tail,
//                           descent test and next trial point from the NNLS answers
//     (the last restarts of a run are finished by the quad solver without kernel boundaries:
//     eng_tail_quad_kernel, ik_quad_tail.hpp);
//   * jobs (one optik_hip_ik_batch call each) submitted before a run share the pool:
//     a slot that finishes a restart of one job may continue with another job's.
//
// Per-restart arithmetic and decisions are the same functions the single-kernel
// solver calls, so results stay bit-identical to the CPU oracle.
#pragma once

#include "ik_nnls_coop.hpp"
#include "ik_solve.hpp"

namespace optik {

// slot states
enum : int32_t {
    ST_EMPTY = 0,          // no restart, queue exhausted
    ST_EVAL_FIRST = 1,     // fresh restart: first evaluation pending
    ST_EVAL_TRIAL = 2,     // line-search trial point pending evaluation
    ST_UPDATE_FIRST = 3,   // first evaluation done: initialise B = I, first direction
    ST_UPDATE_ACCEPT = 4,  // line search accepted: BFGS update + next direction
    ST_NNLS = 5,           // direction needs the bounded solve (deferred to the NNLS kernel)
    ST_REFILL = 6,         // restart published; slot wants the next work item
    ST_DEAD = 7,           // terminated inside an update (status plane); publish next trip
};

template <int N>
struct EngLayout {
    static constexpr int NL = N * (N + 1) / 2;
    // double planes
    static constexpr int X = 0, X0 = X + N, G = X0 + N, S = G + N, GN = S + N, XB = GN + N, XP = XB + N,
                         L = XP + N, F0 = L + NL, H3 = F0 + 1, AL = H3 + 1, FP = AL + 1, MF = FP + 1,
                         FC = MF + 1, ND = FC + 1;
    // int32 planes
    static constexpr int STATE = 0, LINE = 1, IRESET = 2, ITER = 3, NEVALS = 4, STATUS = 5, JOB = 6, NNIT = 7, NI = 8;
};

// One submitted optik_hip_ik_batch call.
struct EngJob {
    const double *targets;             // [T][7]
    const double *x0;                  // [T][n]
    unsigned long long item_base;      // first global work item of the job
    unsigned long long n_items;        // T * R
    unsigned long long n_restarts;     // R
    unsigned long long restart_begin;
    double *out_x;                     // [n][T*R]
    double *out_f;
    double *out_key;
    int32_t *out_status;
    int32_t *out_evals;
    unsigned long long *first_success; // [T] or null (Speed early exit)
    int quality;
    // queue order of the job's work items: 0 = target-major (item = t*R + r), 1 = restart-major
    // (item = r*T + t: every target's low restart indices first -- with Speed early exit most
    // higher indices are then abandoned before they are ever evaluated)
    int restart_major;
    unsigned long long n_targets;      // T
    int find_any;                      // Speed early exit on ANY success of the target (lib.rs:409-412 with several threads)
    int pad_fa;
};

// jobs pooled in one run (the table is staged in the eval kernel's LDS, 120 bytes a job); more
// submissions than this are executed as consecutive runs by optik_hip_engine_run
constexpr int ENG_MAX_JOBS = 256;
// The slot pool runs as a few sub-pools, each with its own stream, lists and trip loop, all
// pulling work items from the one queue: while one sub-pool is in its (VALU-bound) NNLS
// kernel the others' latency-bound per-slot kernels fill the idle issue slots.
constexpr int ENG_MAX_POOLS = 4;
// Bounded sub-problems are listed by the iteration count their restart's previous
// sub-problem took (76% repeat it, 92% within one): the NNLS kernel walks the classes from
// the largest down, so a wave's 16 problems take similar numbers of passes and the long
// ones start first.  Scheduling only -- every problem is solved independently.
constexpr int NN_CLASSES = 8;
// counters of a sub-pool: class sizes per trip parity, then slots to refill, slots in use
constexpr int ENG_POOL_COUNTERS = 2 * NN_CLASSES + 3;
constexpr unsigned NN_NONE = 0xffffffffu;  // nn_cls entry of a slot without a problem
// A problem that needs more than the launch's pass budget is suspended and continues in
// the next trip's launch (its slot just stays in ST_NNLS): no launch waits for the rare
// 10+ pass problem.  Carry record: the transformed matrix [112], then b[8], up, nsetp, iter,
// xv[16] at +16, pos[16] at +32.
constexpr int NN_CARRY_STATE = 112;
constexpr int NN_CARRY = NN_CARRY_STATE + 48;

struct EngArgs {
    const ChainDev *chain;
    EvalParams ep;
    SolveParams sp;
    uint32_t key[8];
    double scale[MAX_DOF];
    double *d;                          // ND planes of C doubles
    int32_t *i32;                       // NI planes of C ints
    unsigned long long *item;           // [C] local item (t * R + r) of the slot's restart
    unsigned long long C;               // pool capacity = stride of the planes
    unsigned long long slot_base;       // first slot of this sub-pool
    unsigned long long n_slots;         // live prefix [slot_base, slot_base + n_slots) the per-slot kernels cover (shrinks while the pool drains)
    const EngJob *jobs;                 // [n_jobs] in device memory
    int n_jobs;
    // max_time expired (lib.rs:308: every callback of every restart then returns None): restarts
    // in flight are published as FORCED_STOP at their next evaluation, queued ones are never started
    int abort;
    unsigned long long total_items;
    unsigned long long *next_item;      // global queue head
    // bounded sub-problems: one record per slot (a slot has at most one outstanding); the
    // lists of a trip are double-buffered by trip parity (the finish kernel of trip s and a
    // suspended solve both defer into the list of trip s+1)
    unsigned int *nn_cls[2];            // [sub-pool size] predicted class of the slot's problem for the trip, or NN_NONE
                                        // (a plain store per emission: no list atomics in the per-slot kernels)
    unsigned int *nn_class_count[2];    // [NN_CLASSES] class sizes (bucket kernel)
    unsigned int *nn_order[2];          // [NN_CLASSES][C] the trip's slots by predicted class (bucket kernel)
    double *nn_prob;                    // [C][2n][n+1] dual problem of the slot, one contiguous block
    double *nn_y;                       // [C][2n] multipliers
    double *nn_meta;                    // [C][2] {mode + 8 * passes, rnorm | -1 = suspended, resume from nn_carry}
    double *nn_carry;                   // [C][NN_CARRY] matrix and state of a suspended solve
    int nn_budget;                      // solve passes per problem per launch (stragglers continue next trip)
    int nn_slack;                       // ... and per problem: at most its predicted count + nn_slack
    int nn_pred_viol;                   // prediction also from the number of violated bounds
    int pad5;
    int parity;                         // list consumed by this trip's NNLS kernel
    int pad6;
    unsigned int *n_active;             // slots holding a restart or waiting for one (bucket kernel; reset every trip)
    // slots whose restart was published this trip (or never started) and want the next work item:
    // listed by the bucket kernel, handed out one per lane by the finish kernel (a finished
    // restart per 39 evaluations would otherwise drag 4 of 5 waves through the refill code for
    // one or two lanes)
    unsigned int *refill_count;
    unsigned int *refill_list;          // [sub-pool size]
    unsigned int *host_in_use;          // pinned host word the finish kernel copies the in-use count to (last trip of a chunk), or null
    unsigned long long *nn_total;       // running count of bounded sub-problems solved
    unsigned long long *prof;           // OPTIK_PROFILE builds: cycle counters, else null
    unsigned long long *prof2;          // OPTIK_PROFILE builds: direction-search sub-phases
    unsigned int *trip_log;             // diagnostics (OPTIK_ENG_TRIP_LOG): [trip][2] = {slots in use, sub-problems}
    int trip;
    int pad4;
    unsigned long long *trace;          // OPTIK_NNLS_TRACE builds: per-wave {start, end, hw id, passes} of one trip
    unsigned long long tail_deadline_ticks;  // tail kernel: wall-clock ticks after its start at which max_time expires (0 = none)
    // max_time on the device's own clock: *deadline_word = wall_clock64() value at which the run's budget
    // expires (written once at the start of the run; 0 or a null pointer = none).  Every evaluation and every
    // refill compares it with the clock (lib.rs:308 checks at every callback), so the run reacts within a
    // trip instead of at the host's next look (`abort`, chunks of four trips queued two deep).
    const unsigned long long *deadline_word;
    // objective + gradient evaluations actually executed (NLopt's per-restart count, out_evals, also
    // counts the re-evaluation of an accepted trial point that the kernels skip): 64 counters, one
    // per workgroup index mod 64, so that the per-wave additions do not queue on one L2 word
    unsigned long long *exec_evals;
    // same-trip continuation of the NNLS launch: problems that used up their (tight) pass
    // budget in the main launch are appended to one of NN_CONT_SHARDS lists (by workgroup
    // index, so that the appending waves do not queue on one L2 word) and a second, small
    // launch of the same kernel (cont_pass = 1) resumes them from their carry records.  A
    // wave runs as many passes as the slowest of its 16 problems: with the cap at the
    // predicted count almost every problem of a wave is done when the wave is.
    unsigned int *cont_count;           // [NN_CONT_SHARDS], reset by the eval kernel
    unsigned int *cont_list;            // [NN_CONT_SHARDS][cont_cap] slots
    unsigned int cont_cap;
    int cont_pass;                      // 0 = main launch, 1 = continuation launch
};
constexpr int NN_CONT_SHARDS = 8;
constexpr int ENG_EXEC_SHARDS = 64;

// double planes: tiled by 64 slots, two planes interleaved per lane -- the planes of slots
// [64 t, 64 t + 64) are contiguous ([t][plane / 2][64][2]).  A wave's ~80 planes sit in one
// 42 KB block instead of 80 places C * 8 bytes apart, and neighbouring planes of a lane are
// adjacent, so most accesses are 16 bytes per lane (8-byte-per-lane loads and stores are
// issue-bound at ~7 B/clk/CU, about 4.3 TB/s over the chip: measured on the update kernel).
#ifndef OPTIK_ENG_PLANE_MAJOR
#define ENG_D_AT(base, nd, C, p, s)                                                                        \
    (base)[((size_t)(s) >> 6) * (size_t)((((nd) + 1) / 2) * 128) + (size_t)((p) >> 1) * 128u +           \
           (((size_t)(s) & 63u) << 1) + (size_t)((p) & 1)]
#define ENG_D(plane, k) ENG_D_AT(a.d, EngLayout<N>::ND, a.C, (plane) + (k), slot)
#else
#define ENG_D(plane, k) a.d[(size_t)((plane) + (k)) * a.C + slot]
#define ENG_D_AT(base, nd, C, p, s) (base)[(size_t)(p) * (C) + (s)]
#endif
#define ENG_I(plane) a.i32[(size_t)(plane) * a.C + slot]

// lib.rs:260-264, 308: has the run's time budget expired (host's flag, or the device clock past the deadline)?
OPTIK_DEV bool eng_timed_out(const EngArgs &a) {
    if (a.abort) return true;
    if (!a.deadline_word) return false;
    const unsigned long long dl = *a.deadline_word;
    return dl != 0ull && (unsigned long long)wall_clock64() > dl;
}

// Outcome of the direction search for one slot.
enum : int { DIR_OK = 0, DIR_DEFER = 1, DIR_DEAD = 2 };

// Lists the slot for the NNLS launch of trip parity `parity`, with its predicted class.
template <int N>
OPTIK_DEV void list_problem(const EngArgs &a, int parity, size_t slot, int pred) {
    a.nn_cls[parity][slot - a.slot_base] = (unsigned)(pred < 0 ? 0 : (pred >= NN_CLASSES ? NN_CLASSES - 1 : pred));
}

// Problem record of a slot, in the order the update kernel produces it: for row i of E^-1
// its entries j >= i, then h_lo[i], h_hi[i]; padded to whole 64-byte lines.  The -E^-1 block
// of the dual matrix is the exact negation and is rebuilt by the reader.
template <int N>
__host__ __device__ constexpr int rec_stride() { return (N * (N + 1) / 2 + 2 * N + 7) / 8 * 8; }
template <int N>
OPTIK_DEV constexpr int rec_row(int i) { return i * (N + 2) - (i * (i - 1)) / 2; }   // offset of row i
template <int N>
OPTIK_DEV constexpr int rec_g(int i, int j) { return rec_row<N>(i) + (j - i); }      // E^-1(i, j), j >= i
template <int N>
OPTIK_DEV constexpr int rec_hlo(int i) { return rec_row<N>(i) + (N - i); }
template <int N>
OPTIK_DEV constexpr int rec_hhi(int i) { return rec_row<N>(i) + (N - i) + 1; }

// The per-slot kernels see 64 consecutive slots per wave, i.e. one contiguous block of
// records, but each lane touches only its own record: 42 scattered 8-byte accesses per lane
// (measured: 8.7 M write requests per update launch, L2 82 % busy, the kernel bound by it).
// Instead the lanes stage their records in an LDS window of the wave and the wave moves the
// block with whole-line accesses.  The first REC_DIRECT elements of a record bypass the
// window (it would not fit beside a second workgroup otherwise).
#ifndef OPTIK_REC_STAGED
#define OPTIK_REC_STAGED 38
#endif
template <int N>
struct RecIo {
    static constexpr int LEN = (N * (N + 1) / 2 + 2 * N + 1) / 2 * 2;  // record length, even
    static constexpr int STAGED = LEN < OPTIK_REC_STAGED ? LEN : OPTIK_REC_STAGED;  // elements of a record in the window
    static constexpr int DIRECT = LEN - STAGED;                        // leading elements accessed in HBM
    static constexpr int ROW = STAGED + 1;                             // window row stride (odd: no bank conflicts)
    static constexpr int WINDOW = 64 * ROW;                            // doubles per wave
    double *rec;   // the lane's record in HBM
    double *win;   // the lane's row of the wave's window, or null: every element in HBM
    OPTIK_DEV void put(int e, double v) const {
        if (e < DIRECT || !win) rec[e] = v;
        else win[e - DIRECT] = v;
    }
    OPTIK_DEV double get(int e) const { return (e < DIRECT || !win) ? rec[e] : win[e - DIRECT]; }
    // window -> HBM for the records of the lanes in `mask` (all 64 lanes call; slot0 = the wave's first slot)
    static OPTIK_DEV void flush(double *recs, size_t slot0, const double *window, unsigned long long mask) {
        const unsigned lane = threadIdx.x & 63u;
#pragma unroll
        for (int t = 0; t < STAGED / 2; ++t) {
            const unsigned p = (unsigned)t * 64u + lane;
            const unsigned r = p / (STAGED / 2), e = (p % (STAGED / 2)) * 2;
            if ((mask >> r) & 1ull) {
                double *dst = recs + (slot0 + r) * rec_stride<N>() + DIRECT + e;
                dst[0] = window[r * ROW + e];
                dst[1] = window[r * ROW + e + 1];
            }
            if (t % 4 == 3) OPTIK_SCHED_FENCE();  // (a few transfers in flight, not all 19: registers)
        }
        OPTIK_SCHED_FENCE();
    }
    // HBM -> window, same shape
    static OPTIK_DEV void fill(const double *recs, size_t slot0, double *window, unsigned long long mask) {
        const unsigned lane = threadIdx.x & 63u;
#pragma unroll
        for (int t = 0; t < STAGED / 2; ++t) {
            const unsigned p = (unsigned)t * 64u + lane;
            const unsigned r = p / (STAGED / 2), e = (p % (STAGED / 2)) * 2;
            if ((mask >> r) & 1ull) {
                const double *src = recs + (slot0 + r) * rec_stride<N>() + DIRECT + e;
                window[r * ROW + e] = src[0];
                window[r * ROW + e + 1] = src[1];
            }
            if (t % 4 == 3) OPTIK_SCHED_FENCE();
        }
        OPTIK_SCHED_FENCE();
    }
};

// LDP tail (Lawson-Hanson ch. 23) from the NNLS answer: transformed-space step.  The rows
// of +-E^-1 and h are read back from the slot's problem record (the NNLS kernel leaves it
// untouched), so the finishing pass does not recompute them.
template <int N>
OPTIK_DEV int ldp_from_record(const RecIo<N> &rec, const double *y_mem, const double *meta, double (&s)[N]) {
    constexpr int M = 2 * N;
    int mode = ((int)meta[0]) & 7;
    if (mode == 1 && meta[1] <= 0.0) mode = 4;
    if (mode != 1) return mode;
    double y[M];
#pragma unroll
    for (int r = 0; r < M; ++r) y[r] = y_mem[r];
    double hy = 0.0;
#pragma unroll
    for (int r = 0; r < M; ++r) hy += rec.get(r < N ? rec_hlo<N>(r) : rec_hhi<N>(r - N)) * y[r];
    double fac = 1.0 - hy;
    const double d1 = 1.0 + fac;
    if (d1 - 1.0 <= 0.0) return 4;
    fac = 1.0 / fac;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r <= j; ++r) acc += rec.get(rec_g<N>(r, j)) * y[r];
#pragma unroll
        for (int r = 0; r <= j; ++r) acc += (-rec.get(rec_g<N>(r, j))) * y[N + r];
        s[j] = fac * acc;
        OPTIK_SCHED_FENCE();
    }
    return 1;
}

// Outcome of one LSQ pass: the three below, or "not a descent direction: reset and repeat".
enum : int { DIR_RESET = 3 };

// One pass of Kraft labels 110/130 with the factor l: LSQ direction, descent test.  The rows
// of the dual problem [G E^-1; h] stream into the slot's record as they are computed; a
// direction whose step leaves the box needs NNLS: the slot is listed for the cooperative
// kernel and deferred.  `resume` completes a deferred pass with the NNLS answer instead.
template <int N>
OPTIK_DEV int direction_pass(const EngArgs &a, const ChainDev &ch, size_t slot, int emit_parity, bool resume,
                             const double (&l)[N * (N + 1) / 2], const double (&g)[N], const double (&x)[N],
                             double (&s)[N], double &h3, int32_t &status, int pred, const RecIo<N> &rec
#ifdef OPTIK_PROFILE
                             , unsigned long long *dp = nullptr
#endif
                             ) {
#ifdef OPTIK_PROFILE
#define DIR_PROBE(k) do { if (dp) { OPTIK_SCHED_FENCE(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); dp[k] = clock64(); OPTIK_SCHED_FENCE(); } } while (0)
#else
#define DIR_PROBE(k)
#endif
    DIR_PROBE(0);
    double E[N][N], fv[N];
    int lmode = lsq_factor<N>(l, g, E, fv);
    OPTIK_SCHED_FENCE();
    DIR_PROBE(1);
    double lo[N], hi[N];  // the box around x (formed where first needed: 14 registers less across the LDP tail)
    if (lmode == 1) {
        if (resume) {
            lmode = ldp_from_record<N>(rec, a.nn_y + slot * 2 * N, a.nn_meta + slot * 2, s);
            OPTIK_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < N; ++i) { lo[i] = ch.lb[i] - x[i]; hi[i] = ch.ub[i] - x[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) { lo[i] = ch.lb[i] - x[i]; hi[i] = ch.ub[i] - x[i]; }
            OPTIK_SCHED_FENCE();
            int nviol = 0;  // bounds the unconstrained step violates (the positive duals NNLS starts from)
            const bool need = lsq_bound_rows<N>(E, fv, lo, hi, [&](int i, const double (&row)[N], double h_lo, double h_hi) {
#pragma unroll
                for (int r = i; r < N; ++r) rec.put(rec_g<N>(i, r), row[r]);
                rec.put(rec_hlo<N>(i), h_lo);
                rec.put(rec_hhi<N>(i), h_hi);
                nviol += (h_lo > 0.0 ? 1 : 0) + (h_hi > 0.0 ? 1 : 0);
            });
            DIR_PROBE(2);
            if (need) {
                // predicted solve passes: what the restart's previous problem took, or the number
                // of violated bounds if that is larger (each enters the active set in its own pass)
                list_problem<N>(a, emit_parity, slot, (a.nn_pred_viol && nviol > pred) ? nviol : pred);
                a.nn_meta[slot * 2 + 1] = 0.0;  // a fresh problem, not a resumed one
                return DIR_DEFER;
            }
#pragma unroll
            for (int j = 0; j < N; ++j) s[j] = 0.0;
        }
    }
    if (lmode != 1) {
        // NLopt: modes 5,6,7 -> ROUNDOFF_LIMITED; 3,4,9 -> FAILURE
        status = (lmode == 5 || lmode == 6 || lmode == 7) ? RES_ROUNDOFF_LIMITED : RES_FAILURE;
        return DIR_DEAD;
    }
    lsq_finish<N>(E, fv, lo, hi, s);
    OPTIK_SCHED_FENCE();
    DIR_PROBE(3);
    double gs = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) gs += g[i] * s[i];
    h3 = gs;
    return (h3 >= 0.0) ? DIR_RESET : DIR_OK;
}

// Kraft labels 110/130 for one slot: (reset,) LSQ direction, descent test, repeated with a
// reset factor while the direction is not a descent direction.  The caller has l in the
// slot planes already (a reset stores the identity here); the first pass runs on the
// caller's l, every reset pass on the constant identity -- l is not carried around the
// loop, which would pin its 2 x 28 registers for the whole search.  `resume` re-enters at
// the LSQ call of a deferred pass (its ++iter / reset are done).
template <int N>
OPTIK_DEV int direction_search(const EngArgs &a, const ChainDev &ch, size_t slot, int emit_parity, bool resume,
                               const double (&l)[N * (N + 1) / 2], const double (&g)[N], const double (&x)[N],
                               double f, int &ireset, int &iter, bool reset, double (&s)[N], double &h3,
                               int32_t &status, int pred, const RecIo<N> &rec
#ifdef OPTIK_PROFILE
                               , unsigned long long *dp = nullptr
#endif
                               ) {
    using EL = EngLayout<N>;
    constexpr int NL = N * (N + 1) / 2;
    const SolveParams &sp = a.sp;
    bool have0 = false;  // a completed LSQ in this call: Kraft's (f0, x0) = (f, x)
    if (!reset) {
        if (!resume) ++iter;
#ifdef OPTIK_PROFILE
        const int r = direction_pass<N>(a, ch, slot, emit_parity, resume, l, g, x, s, h3, status, pred, rec, dp);
#else
        const int r = direction_pass<N>(a, ch, slot, emit_parity, resume, l, g, x, s, h3, status, pred, rec);
#endif
        if (r != DIR_RESET) return r;
        have0 = true;
    }
    double ident[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) ident[i] = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) ident[lidx<N>(i, i)] = 1.0;
    for (;;) {
        ++ireset;
        if (ireset > 5) {
            // label 255 with acc = 0 -> mode 8; NLopt's relaxed test vs (f0, x0) = (f, x)
            status = RES_ROUNDOFF_LIMITED;
            if (have0 && __builtin_fabs(f - f) < sp.ftol_abs && !__builtin_isinf(f)) status = RES_FTOL_REACHED;  // f0 = f
            else if (have0 && (sp.stop_x_zero || !(0.0 >= sp.xtol_abs))) status = RES_XTOL_REACHED;  // |x - x0| = 0 everywhere
            return DIR_DEAD;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) ENG_D(EL::L, i) = ident[i];
        ++iter;
        const int r = direction_pass<N>(a, ch, slot, emit_parity, false, ident, g, x, s, h3, status, pred, rec);
        if (r != DIR_RESET) return r;
        have0 = true;
    }
}

// Stores the result of a successful direction search: label 190 with alpha = 1.
// (STORE_G = false: the caller read g from the slot's plane and did not change it)
template <int N, bool STORE_G = true>
OPTIK_DEV void store_direction(const EngArgs &a, const ChainDev &ch, size_t slot,
                               const double (&l)[N * (N + 1) / 2], const double (&g)[N], const double (&x)[N],
                               const double (&s)[N], double f, double h3, int ireset, int iter) {
    using E = EngLayout<N>;
    (void)l;  // (already in the slot planes)
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (STORE_G) ENG_D(E::G, i) = g[i];
        const double si = s[i] * 1.0;  // s *= alpha (alpha = 1)
        ENG_D(E::S, i) = si;
        ENG_D(E::X0, i) = x[i];
        double xi = x[i];
        xi += si;
        if (xi < ch.lb[i]) xi = ch.lb[i];
        else if (xi > ch.ub[i]) xi = ch.ub[i];
        ENG_D(E::X, i) = xi;
    }
    ENG_D(E::F0, 0) = f;
    ENG_D(E::H3, 0) = 1.0 * h3;
    ENG_D(E::AL, 0) = 1.0;
    ENG_I(E::LINE) = 1;
    ENG_I(E::IRESET) = ireset;
    ENG_I(E::ITER) = iter;
    ENG_I(E::STATE) = ST_EVAL_TRIAL;
}

template <int N, bool STORE_G = true>
OPTIK_DEV void store_deferred(const EngArgs &a, size_t slot, const double (&l)[N * (N + 1) / 2],
                              const double (&g)[N], int ireset, int iter) {
    using E = EngLayout<N>;
    (void)l;  // (already in the slot planes)
    if (STORE_G) {
#pragma unroll
        for (int i = 0; i < N; ++i) ENG_D(E::G, i) = g[i];
    }
    ENG_I(E::IRESET) = ireset;
    ENG_I(E::ITER) = iter;
    ENG_I(E::STATE) = ST_NNLS;
}

// ---- kernel 1: evaluate + decide ------------------------------------------------

// Returns whether the objective was evaluated.
template <int N, bool TIP>
OPTIK_DEV bool eng_eval_body(const EngArgs &a, const ChainDev &ch, const EngJob *jobs, size_t slot) {
    using E = EngLayout<N>;
    const double alfmin = 0.1;
    int st = ENG_I(E::STATE);
    if (st != ST_EVAL_FIRST && st != ST_EVAL_TRIAL && st != ST_DEAD) return false;
    const EngJob &J = jobs[ENG_I(E::JOB)];  // (job table staged in LDS: no dependent HBM round trip)
    const unsigned long long item = a.item[slot];
    const unsigned long long tslot = item / J.n_restarts;
    const unsigned long long index = J.restart_begin + (item - tslot * J.n_restarts);
    int32_t ret = 0;
    int nevals = ENG_I(E::NEVALS);
    double minf = ENG_D(E::MF, 0);
    if (st == ST_DEAD) {
        ret = ENG_I(E::STATUS);
    } else {
        // lib.rs:308: abandon when a lower-index restart of the same target succeeded
        if (J.first_success) {
            const unsigned long long fs = __hip_atomic_load(J.first_success + tslot, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
            if (J.find_any ? (fs != ~0ull) : (fs < index)) ret = RES_FORCED_STOP;
        }
        if (eng_timed_out(a)) ret = RES_FORCED_STOP;  // lib.rs:308: timed out
    }
    const bool evaluated = ret == 0;
    if (ret == 0) {
        double f;
        {
            double x[N];
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = ENG_D(E::X, i);
            const Pose target = load_pose(J.targets + (size_t)tslot * 7);
            // the gradient goes straight to the g_new plane (the update kernel reads it only after
            // an evaluation that hands the slot over to it, which is the one that wrote it last);
            // neither it nor x stays in registers across the evaluation
            f = eval_fg_stream<N, TIP>(ch, a.ep, target, x, [&](int k, double v) { ENG_D(E::GN, k) = v; });
        }
        ++nevals;
        // NLopt: update best point so far; stopval is tested after every evaluation
        if (f < minf) {
            minf = f;
            ENG_D(E::MF, 0) = f;
#pragma unroll
            for (int i = 0; i < N; ++i) ENG_D(E::XB, i) = ENG_D(E::X, i);
        }
        if (minf < a.sp.stopval) {
            ret = RES_STOPVAL_REACHED;
        } else if (nevals >= MAX_EVALS_CAP) {
            ret = RES_ITER_CAP;
        } else if (st == ST_EVAL_FIRST) {
            ENG_D(E::FC, 0) = f;
            ENG_I(E::STATE) = ST_UPDATE_FIRST;
        } else {
            // label 220: L1 merit (m = 0: the objective itself)
            const double t0 = ENG_D(E::F0, 0);
            const double h3 = ENG_D(E::H3, 0);
            double alpha = ENG_D(E::AL, 0);
            const int line = ENG_I(E::LINE);
            const double h1 = f - t0;
            bool accept = false;
            if (__builtin_isfinite(h1)) {
                if (h1 <= h3 / 10.0 || line > 10) accept = true;
                else {
                    const double al = h3 / ((h3 - h1) * 2.0);
                    alpha = (al > alfmin) ? al : alfmin;
                }
            } else {
                const double al = alpha * 0.5;
                alpha = (al > alfmin) ? al : alfmin;
            }
            if (accept) {
                if (line > 1) ++nevals;  // NLopt re-evaluates the accepted point unless it was trial 1
                const double fprev = ENG_D(E::FP, 0);
                if (!__builtin_isinf(fprev)) {
                    if (__builtin_fabs(f - fprev) < a.sp.ftol_abs) ret = RES_FTOL_REACHED;
                    else if (xprev_live(a.sp)) {
                        double xc[N], xp[N];
#pragma unroll
                        for (int i = 0; i < N; ++i) { xc[i] = ENG_D(E::X, i); xp[i] = ENG_D(E::XP, i); }
                        if (stop_x<N>(a.sp, xc, xp)) ret = RES_XTOL_REACHED;
                    }
                }
                ENG_D(E::FP, 0) = f;
                if (xprev_live(a.sp)) {
#pragma unroll
                    for (int i = 0; i < N; ++i) ENG_D(E::XP, i) = ENG_D(E::X, i);
                }
                if (ret == 0 && nevals >= MAX_EVALS_CAP) ret = RES_ITER_CAP;
                if (ret == 0) {
                    ENG_D(E::FC, 0) = f;
                    ENG_I(E::STATE) = ST_UPDATE_ACCEPT;
                }
            } else {
                // label 190: next trial point x = x0 + alpha * s, clipped (NLopt)
                ENG_I(E::LINE) = line + 1;
                ENG_D(E::H3, 0) = alpha * h3;
                ENG_D(E::AL, 0) = alpha;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const double si = ENG_D(E::S, i) * alpha;
                    ENG_D(E::S, i) = si;
                    double xi = ENG_D(E::X0, i);
                    xi += si;
                    if (xi < ch.lb[i]) xi = ch.lb[i];
                    else if (xi > ch.ub[i]) xi = ch.ub[i];
                    ENG_D(E::X, i) = xi;
                }
            }
        }
        ENG_I(E::NEVALS) = nevals;
    }
    if (ret != 0) {
        // the restart ended: classify (lib.rs:376-379) and publish
        const bool success = (a.sp.ok_stopval && ret == RES_STOPVAL_REACHED)
                             || (a.sp.ok_ftol && ret == RES_FTOL_REACHED)
                             || (a.sp.ok_xtol && ret == RES_XTOL_REACHED);
        double xb[N];
#pragma unroll
        for (int i = 0; i < N; ++i) xb[i] = ENG_D(E::XB, i);
        if (J.out_x) {
#pragma unroll
            for (int i = 0; i < N; ++i) J.out_x[(size_t)i * J.n_items + item] = xb[i];
        }
        if (J.out_f) J.out_f[item] = minf;
        if (J.out_status) J.out_status[item] = ret;
        if (J.out_evals) J.out_evals[item] = nevals;
        double k = __builtin_huge_val();
        if (success) {
            if (J.quality) {
                const double *x0p = J.x0 + (size_t)tslot * N;
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) { const double d = xb[i] - x0p[i]; acc += d * d; }
                k = __builtin_sqrt(acc);
            } else {
                k = (double)index;
                if (J.first_success) atomicMin(J.first_success + tslot, index);
            }
        }
        if (J.out_key) J.out_key[item] = k;
        ENG_I(E::STATE) = ST_REFILL;
    }
    return evaluated;
}

// Gives `slot` the work item `it` of the queue (or leaves it empty when the queue is
// exhausted): restart seed, bookkeeping planes.  Returns false when the item needs no slot --
// a lower-index restart of its target has already succeeded (lib.rs:308 at the restart's
// first callback): its outputs keep their initial "no solution" values and the caller asks
// for the next item right away (a Speed batch abandons most of its restarts this way; each
// used to cost its slot a whole trip).
template <int N>
OPTIK_DEV bool refill_slot(const EngArgs &a, const ChainDev &ch, size_t slot, unsigned long long it) {
    using E = EngLayout<N>;
    int st;
    if (it < a.total_items) {
        // the job of work item `it`: the last one whose first item is <= it (binary search)
        int job = 0;
        for (int lo = 1, hi = a.n_jobs; lo < hi;) {
            const int mid = (lo + hi) >> 1;
            if (it >= a.jobs[mid].item_base) { job = mid; lo = mid + 1; } else hi = mid;
        }
        const EngJob &J = a.jobs[job];
        const unsigned long long qi = it - J.item_base;
        unsigned long long tslot, r;
        if (J.restart_major) { r = qi / J.n_targets; tslot = qi - r * J.n_targets; }
        else { tslot = qi / J.n_restarts; r = qi - tslot * J.n_restarts; }
        const unsigned long long item = tslot * J.n_restarts + r;  // output column
        const unsigned long long index = J.restart_begin + r;
        const bool late = eng_timed_out(a);  // timed out before the restart was issued (lib.rs:393)
        bool skip = late;
        if (!skip && J.first_success) {
            const unsigned long long fs = __hip_atomic_load(J.first_success + tslot, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
            skip = J.find_any ? (fs != ~0ull) : (fs < index);  // abandoned before it started (key is +inf from the job's set-up)
        }
        if (skip) {
            if (J.out_status) J.out_status[item] = RES_FORCED_STOP;
            if (J.out_evals) J.out_evals[item] = 0;
            if (late && J.out_key) J.out_key[item] = __builtin_huge_val();
            return false;
        }
        double x[N];
        restart_seed<N>(a.key, ch.lb, a.scale, index, x);
        if (index == 0) {  // lib.rs:366-370: restart 0 starts from the caller's seed
            const double *x0p = J.x0 + (size_t)tslot * N;
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = x0p[i];
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            ENG_D(E::X, i) = x[i];
            ENG_D(E::XB, i) = x[i];
            ENG_D(E::XP, i) = x[i];
        }
        ENG_D(E::MF, 0) = __builtin_huge_val();
        ENG_D(E::FP, 0) = __builtin_huge_val();
        ENG_I(E::NEVALS) = 0;
        ENG_I(E::ITER) = 0;
        ENG_I(E::IRESET) = 0;
        ENG_I(E::LINE) = 0;
        ENG_I(E::NNIT) = 1;
        ENG_I(E::JOB) = job;
        a.item[slot] = item;
        st = ST_EVAL_FIRST;
    } else {
        st = ST_EMPTY;
    }
    ENG_I(E::STATE) = st;
    return true;
}

// ---- kernel 2: update (BFGS + unconstrained direction) and refill ---------------

template <int N>
OPTIK_DEV void eng_update_body(const EngArgs &a, const ChainDev &ch, size_t slot, size_t local, bool in_range,
                                double *window) {
    using E = EngLayout<N>;
    int st = in_range ? ENG_I(E::STATE) : ST_EMPTY;
#ifdef OPTIK_PROFILE
    // phase timers of the update kernel (tools/engine_phase_profile.py): wave cycles between
    // probes, summed over the waves that ran a direction search
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long prof_factor = 0, prof_rows = 0, prof_tail = 0;
#define ENG_PROBE(k) do { OPTIK_SCHED_FENCE(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pt[k] = clock64(); OPTIK_SCHED_FENCE(); } while (0)
    ENG_PROBE(0);
#else
#define ENG_PROBE(k)
#endif

    ENG_PROBE(1);
#ifdef OPTIK_PROFILE
    const bool prof_wave = wave_any(st == ST_UPDATE_FIRST || st == ST_UPDATE_ACCEPT);
#endif
    // the lane's problem record: staged in the wave's LDS window, moved to HBM at the end
    const unsigned lane_id = threadIdx.x & 63u;
    const RecIo<N> rec{a.nn_prob + slot * rec_stride<N>(), window + lane_id * RecIo<N>::ROW};
    bool emitted = false;
    if (st == ST_UPDATE_FIRST || st == ST_UPDATE_ACCEPT) {
        double l[E::NL], g[N], x[N], s[N];
        const double f = ENG_D(E::FC, 0);
        int ireset = ENG_I(E::IRESET), iter = ENG_I(E::ITER);
#pragma unroll
        for (int i = 0; i < N; ++i) { x[i] = ENG_D(E::X, i); g[i] = ENG_D(E::GN, i); }
        if (st == ST_UPDATE_ACCEPT) {
            // label 260: BFGS update with u = g_new - g_old
            double u[N];
#pragma unroll
            for (int i = 0; i < E::NL; ++i) l[i] = ENG_D(E::L, i);
#pragma unroll
            for (int i = 0; i < N; ++i) { s[i] = ENG_D(E::S, i); u[i] = g[i] - ENG_D(E::G, i); }
            OPTIK_SCHED_FENCE();
            ENG_PROBE(2);
            bfgs_update<N>(l, s, u);
            OPTIK_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < E::NL; ++i) ENG_D(E::L, i) = l[i];
        } else {
#pragma unroll
            for (int i = 0; i < E::NL; ++i) l[i] = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) s[i] = 0.0;
        }
        double h3 = 0.0;
        int32_t status = 0;
        ENG_PROBE(3);
#ifdef OPTIK_PROFILE
        unsigned long long dpt[4] = {0, 0, 0, 0};
        const int out = direction_search<N>(a, ch, slot, a.parity, false, l, g, x, f, ireset, iter,
                                            st == ST_UPDATE_FIRST, s, h3, status, ENG_I(E::NNIT), rec, dpt);
        ENG_PROBE(5);  // (reused below: end of the search, before the stores)
        if (dpt[1] && dpt[0]) pt[0] += 0;  // keep dpt live
        prof_factor = (dpt[1] && dpt[0]) ? dpt[1] - dpt[0] : 0;
        prof_rows = (dpt[2] && dpt[1]) ? dpt[2] - dpt[1] : 0;
        prof_tail = dpt[2] ? pt[5] - dpt[2] : 0;
#else
        const int out = direction_search<N>(a, ch, slot, a.parity, false, l, g, x, f, ireset, iter,
                                            st == ST_UPDATE_FIRST, s, h3, status, ENG_I(E::NNIT), rec);
#endif
        emitted = out == DIR_DEFER;
        if (out == DIR_OK) store_direction<N>(a, ch, slot, l, g, x, s, f, h3, ireset, iter);
        else if (out == DIR_DEFER) store_deferred<N>(a, slot, l, g, ireset, iter);
        else { ENG_I(E::STATUS) = status; ENG_I(E::ITER) = iter; ENG_I(E::STATE) = ST_DEAD; }
        st = ST_EVAL_TRIAL;  // (any non-empty value: the slot still holds a restart)
        ENG_PROBE(4);
    }
    RecIo<N>::flush(a.nn_prob, (size_t)a.slot_base + (local - lane_id), window, __ballot(emitted));
#ifdef OPTIK_PROFILE
    ENG_PROBE(5);
    if (a.prof && prof_wave) {
        // per-lane probes differ only by divergence; lane maxima describe the wave
        unsigned long long d[5];
        d[0] = pt[1] - pt[0];                       // refill + entry loads
        d[1] = pt[2] ? pt[2] - pt[1] : 0;           // plane loads of an accepted step
        d[2] = (pt[3] && pt[2]) ? pt[3] - pt[2] : 0;  // BFGS
        d[3] = (pt[4] && pt[3]) ? pt[4] - pt[3] : 0;  // direction search + stores
        d[4] = pt[5] - pt[0];                       // whole body
        unsigned long long d2[3] = {prof_factor, prof_rows, prof_tail};
        if (a.prof2) for (int k = 0; k < 2; ++k) {
            unsigned long long v = d2[k];
            for (int off = 32; off >= 1; off >>= 1) { const unsigned long long o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
            if ((threadIdx.x & 63u) == 0) atomicAdd(a.prof2 + k, v);
        }
        for (int k = 0; k < 5; ++k) {
            unsigned long long v = d[k];
            for (int off = 32; off >= 1; off >>= 1) { const unsigned long long o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
            if ((threadIdx.x & 63u) == 0) atomicAdd(a.prof + k, v);
        }
        if ((threadIdx.x & 63u) == 0) atomicAdd(a.prof + 7, 1ull);
    }
#endif
#undef ENG_PROBE
    // count the slots that still hold a restart (termination test on the host)

}

// ---- kernel 3: cooperative NNLS over this trip's deferred problems ---------------

template <int N, int CPL>
OPTIK_DEV void eng_nnls_coop_body(const EngArgs &a, double *wave_lds) {
    constexpr int m = N + 1, n = 2 * N;
    constexpr int G = COOP_COLS / CPL;        // lanes per problem
    constexpr unsigned PPW = 64 / G;          // problems per wave
    const unsigned int *cls_cnt = a.nn_class_count[a.parity];
    unsigned cnt = 0;
    unsigned shard_cnt[NN_CONT_SHARDS];
    if (a.cont_pass) {
#pragma unroll
        for (int c = 0; c < NN_CONT_SHARDS; ++c) {
            const unsigned v = a.cont_count[c];
            shard_cnt[c] = v < a.cont_cap ? v : a.cont_cap;
            cnt += shard_cnt[c];
        }
    } else {
        for (int c = 0; c < NN_CLASSES; ++c) cnt += cls_cnt[c];
    }
    double *ybuf = a.nn_y;
    double *meta = a.nn_meta;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned group = lane / G, gl = lane % G;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64u;
    const unsigned n_waves = (gridDim.x * blockDim.x) / 64u;
    if (blockIdx.x == 0 && threadIdx.x == 0 && !a.cont_pass) {
        if (a.nn_total) atomicAdd(a.nn_total, (unsigned long long)cnt);
        if (a.trip_log) { a.trip_log[2 * a.trip] = *a.n_active; a.trip_log[2 * a.trip + 1] = cnt; }
    }
    const unsigned int *order = a.nn_order[a.parity];
#ifdef OPTIK_NNLS_TRACE
    const unsigned long long t_begin = wall_clock64();
    const unsigned long long c_begin = clock64();
    int max_passes = -1, sum_max = 0, sum_passes = 0, n_problems = 0;
#endif
    for (unsigned q0 = wave * PPW; q0 < cnt; q0 += n_waves * PPW) {
        const bool live = q0 + group < cnt;
        // the slot of the (q0 + group)-th problem in class order, largest predicted pass count first
        unsigned q = 0;
        int cls = NN_CLASSES - 1;
        if (a.cont_pass) {
            // the (q0 + group)-th suspended problem over the shard lists
            unsigned i = q0 + group;
            bool placed = !live;
#pragma unroll
            for (int c = 0; c < NN_CONT_SHARDS; ++c) {
                if (!placed && i < shard_cnt[c]) { q = a.cont_list[(size_t)c * a.cont_cap + i]; placed = true; }
                if (!placed) i -= shard_cnt[c];
            }
        } else {
            unsigned i = q0 + group;
            bool placed = !live;
            for (int c = NN_CLASSES - 1; c >= 0; --c) {
                const unsigned cc = cls_cnt[c];
                if (!placed && i < cc) { q = order[(size_t)c * a.C + i]; placed = true; cls = c; }
                if (!placed) i -= cc;
            }
        }
        // pass budget of the problem in this launch: a little more than its predicted count (a
        // wave runs as many passes as its slowest problem: the ~8 % that need more than
        // predicted + slack continue next trip among the long ones instead of holding 15 others)
        const int want = cls + a.nn_slack < 1 ? 1 : cls + a.nn_slack;
        const int budget = (a.cont_pass || cls == NN_CLASSES - 1 || want > a.nn_budget) ? a.nn_budget : want;
        const bool resume = live && meta[(size_t)q * 2 + 1] < 0.0;
        dvec8 col[CPL];
        CoopCarry<CPL> cs;
        const double *cr = a.nn_carry + (size_t)q * NN_CARRY + NN_CARRY_STATE;
        // the matrix: the emitted problem, or the transformed one of a suspended solve
        const double *rec = a.nn_prob + (size_t)q * rec_stride<N>();
        const double *cmat = a.nn_carry + (size_t)q * NN_CARRY;
        cs.b = 0.0;
        cs.up = 0.0;
        cs.nsetp = 0;
        cs.iter = 0;
        if (resume) {
#pragma unroll
            for (int r = 0; r < 8; ++r) cs.b[r] = cr[r];
            cs.up = cr[8];
            cs.nsetp = (int)cr[9];
            cs.iter = (int)cr[10];
        }
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            col[k] = 0.0;
            cs.xv[k] = 0.0;
            cs.pos[k] = 0;
            const unsigned c = gl * CPL + k;
            if (live && c < (unsigned)n) {
                // The emitted record is read whether or not the problem turns out to be a suspended
                // one (8 %: their record is stale but still the slot's memory): these loads then do
                // not wait for the meta word that says which -- one memory latency less at the head
                // of every wave.
                // column c < N: row c of E^-1 on rows c .. N-1, h_lo[c] below; column N + c: its
                // negation, h_hi[c] below
                const bool neg = c >= (unsigned)N;
                const int cc = (int)(neg ? c - N : c);
                const int off = cc * (N + 2) - (cc * (cc - 1)) / 2 - cc;  // rec_row(cc) - cc
                // (unconditional loads of the N + 2 doubles at rec[off ..]: for r < cc they are entries of
                // earlier rows of the same record, discarded below -- adjacent loads the compiler can pair
                // into 16-byte accesses instead of N + 1 predicated 8-byte ones)
                double rowv[N + 2];
#pragma unroll
                for (int r = 0; r < N + 2; ++r) rowv[r] = rec[off + r];
#pragma unroll
                for (int r = 0; r < N; ++r) col[k][r] = (r >= cc) ? (neg ? -rowv[r] : rowv[r]) : 0.0;
                col[k][N] = neg ? rowv[N + 1] : rowv[N];  // h_lo / h_hi follow the row
                if (resume) {
#pragma unroll
                    for (int r = 0; r < m; ++r) col[k][r] = cmat[(size_t)c * m + r];
                }
            }
            if (resume) {
                cs.xv[k] = cr[16 + c];
                cs.pos[k] = (int)cr[32 + c];
            }
        }
        int mode, iters;
        double rnorm;
        // a suspended problem continues in the next trip: matrix and state go to the slot's
        // carry record (stored from inside the solver) and the slot is listed again
        auto park = [&](const dvec8 (&pcol)[CPL], const CoopCarry<CPL> &st) {
            double *cm = a.nn_carry + (size_t)q * NN_CARRY;
            double *cw = cm + NN_CARRY_STATE;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const unsigned c = gl * CPL + k;
                if (c < (unsigned)n) {
                    double *pc = cm + (size_t)c * m;
#pragma unroll
                    for (int r = 0; r < m; ++r) pc[r] = pcol[k][r];
                }
                cw[16 + c] = st.xv[k];
                cw[32 + c] = (double)st.pos[k];
            }
            if (gl == 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r) cw[r] = st.b[r];
                cw[8] = st.up;
                cw[9] = (double)st.nsetp;
                cw[10] = (double)st.iter;
                list_problem<N>(a, a.parity ^ 1, q, NN_CLASSES - 1);  // scheduled with the long ones
                meta[(size_t)q * 2] = (double)(NNLS_SUSPENDED + 8 * st.iter);
                meta[(size_t)q * 2 + 1] = -1.0;
            }
        };
        nnls_coop<N, CPL>(live, resume, budget, (int)(gl * CPL), col, cs, mode, rnorm, iters,
                          wave_lds + group * COOP_WIN, wave_lds + PPW * COOP_WIN, park);
        if (live && mode != NNLS_SUSPENDED) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const unsigned c = gl * CPL + k;
                if (c < (unsigned)n) ybuf[(size_t)q * n + c] = cs.xv[k];
            }
            if (gl == 0) {
                meta[(size_t)q * 2] = (double)(mode + 8 * iters);
                meta[(size_t)q * 2 + 1] = rnorm;
                // finished after all: take it off the next trip's list (park put it there)
                if (a.cont_pass) a.nn_cls[a.parity ^ 1][q - a.slot_base] = NN_NONE;
            }
        }
        if (!a.cont_pass && a.cont_count) {
            // suspended in the main launch: hand the problem to this trip's continuation launch
            const bool sus = live && mode == NNLS_SUSPENDED && gl == 0;
            const unsigned long long sm = __ballot(sus);
            if (sm) {
                const unsigned shard = blockIdx.x % NN_CONT_SHARDS;
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(a.cont_count + shard, (unsigned)__popcll(sm));
                base = (unsigned)__shfl((int)base, 0, 64);
                const unsigned at = base + (unsigned)__popcll(sm & ((1ull << lane) - 1ull));
                if (sus && at < a.cont_cap) a.cont_list[(size_t)shard * a.cont_cap + at] = q;
            }
        }
#ifdef OPTIK_NNLS_TRACE
        {
            int mp = live ? iters : -1;
            int sp = (live && gl == 0) ? iters : 0, np = (live && gl == 0) ? 1 : 0;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const int o = __shfl_xor(mp, off, 64); mp = o > mp ? o : mp;
                sp += __shfl_xor(sp, off, 64); np += __shfl_xor(np, off, 64);
            }
            max_passes = mp > max_passes ? mp : max_passes;
            if (mp > 0) sum_max += mp;
            sum_passes += sp; n_problems += np;
        }
#endif
    }
#ifdef OPTIK_NNLS_TRACE
    if (a.trace && lane == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.trace[(size_t)wave * 4 + 0] = t_begin;
        a.trace[(size_t)wave * 4 + 1] = wall_clock64();
        a.trace[(size_t)wave * 4 + 2] = ((unsigned long long)(n_problems & 0xff) << 56) | ((unsigned long long)(sum_passes & 0xffff) << 40) |
                                        ((unsigned long long)(xcc & 0xff) << 32) | hw;
        a.trace[(size_t)wave * 4 + 3] = ((unsigned long long)(clock64() - c_begin) << 16) | (unsigned long long)(sum_max & 0xffff);  // (passes the wave ran: sum over its batches of the largest count)
    }
#endif
}

// ---- kernel 2b: list the trip's problems by predicted class (counting sort) ---------

#ifndef OPTIK_BUCKET_SUB
#define OPTIK_BUCKET_SUB 4
#endif
constexpr int BUCKET_SUB = OPTIK_BUCKET_SUB;  // 64-slot batches per wave (one atomic instruction per wave)

// One pass over the sub-pool's slots after the update kernel: the slots with a problem for
// this trip go to their class list (and their nn_cls entry is cleared for the trip after
// next), the slots that want a work item to the refill list, and the slots in use are counted.
// Wave-aggregated: lanes 0 .. 9 of a wave own the ten counters, one atomic instruction per
// 256 slots.
OPTIK_DEV void eng_bucket_body(const EngArgs &a) {
    unsigned int *cls_cnt = a.nn_class_count[a.parity];
    unsigned int *order = a.nn_order[a.parity];
    unsigned int *lcls = a.nn_cls[a.parity];
    const int32_t *state = a.i32 + a.slot_base;  // plane 0 = STATE
    const unsigned n = (unsigned)a.n_slots;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64u;
    const unsigned n_waves = (gridDim.x * blockDim.x) / 64u;
    constexpr int K_REFILL = NN_CLASSES, K_USED = NN_CLASSES + 1;
    for (unsigned i0 = wave * (64u * BUCKET_SUB); i0 < n; i0 += n_waves * (64u * BUCKET_SUB)) {
        int cls[BUCKET_SUB];
        bool refill[BUCKET_SUB];
        unsigned rank[BUCKET_SUB], rrank[BUCKET_SUB];
        unsigned mine = 0;  // lane k: members of counter k in this wave's batches
#pragma unroll
        for (int b = 0; b < BUCKET_SUB; ++b) {
            const unsigned i = i0 + (unsigned)b * 64u + lane;
            const unsigned c = i < n ? lcls[i] : NN_NONE;
            const int st = i < n ? state[i] : ST_EMPTY;
            cls[b] = (int)c;  // NN_NONE -> -1
            if (c != NN_NONE) lcls[i] = NN_NONE;
            refill[b] = st == ST_REFILL;
            rank[b] = (unsigned)__shfl((int)mine, cls[b] < 0 ? 0 : cls[b], 64);
            rrank[b] = (unsigned)__shfl((int)mine, K_REFILL, 64);
#pragma unroll
            for (int k = 0; k < NN_CLASSES; ++k) {
                const unsigned long long mk = __ballot(cls[b] == k);
                if ((int)lane == k) mine += (unsigned)__popcll(mk);
                if (cls[b] == k) rank[b] += (unsigned)__popcll(mk & below);
            }
            const unsigned long long mr = __ballot(refill[b]);
            if ((int)lane == K_REFILL) mine += (unsigned)__popcll(mr);
            rrank[b] += (unsigned)__popcll(mr & below);
            const unsigned long long mu = __ballot(st != ST_EMPTY);
            if ((int)lane == K_USED) mine += (unsigned)__popcll(mu);
        }
        unsigned base = 0;
        if (lane <= (unsigned)K_USED && mine) {
            unsigned int *ctr = lane < (unsigned)NN_CLASSES ? cls_cnt + lane : ((int)lane == K_REFILL ? a.refill_count : a.n_active);
            base = atomicAdd(ctr, mine);
        }
        const unsigned rbase = (unsigned)__shfl((int)base, K_REFILL, 64);
#pragma unroll
        for (int b = 0; b < BUCKET_SUB; ++b) {
            const unsigned slot = (unsigned)a.slot_base + i0 + (unsigned)b * 64u + lane;
            const unsigned cb = (unsigned)__shfl((int)base, cls[b] < 0 ? 0 : cls[b], 64);
            if (cls[b] >= 0) order[(size_t)cls[b] * a.C + cb + rank[b]] = slot;
            if (refill[b]) a.refill_list[rbase + rrank[b]] = slot;
        }
    }
}

// ---- pool compaction (drain phase) -----------------------------------------------------
// Once the queue is empty the restarts still running are scattered over the pool and every
// wave of the per-slot kernels keeps a few live lanes.  The host then shrinks the live
// prefix: restarts above the new bound move into free slots below it (a slot is a plain
// record; which slot a restart occupies has no effect on its arithmetic).

struct CompactArgs {
    double *d;
    int32_t *i32;
    unsigned long long *item;
    unsigned long long C, slot_base, n_slots, n_new;  // live prefix [slot_base, +n_slots) -> [slot_base, +n_new)
    int nd, ni;
    unsigned int *counts;   // [0] free slots listed, [1] restarts to move
    unsigned int *free_list, *move_list;
    // a slot waiting for the next NNLS launch takes its problem record and class entry along
    double *nn_prob, *nn_meta, *nn_carry;
    double *nn_y;           // multipliers that have to move with their slot, or null
    unsigned int *nn_cls;   // class entries of the next trip, by sub-pool slot
    int rec_len;            // doubles per problem record
    int ny;                 // doubles per multiplier vector (2n)
};

OPTIK_DEV void compact_scan_body(const CompactArgs &c) {
    const unsigned long long local = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long slot = c.slot_base + local;
    const bool in = local < c.n_slots;
    const int st = in ? c.i32[slot] : ST_EMPTY;  // plane 0 = STATE
    const bool is_free = in && local < c.n_new && st == ST_EMPTY;
    const bool is_move = in && local >= c.n_new && st != ST_EMPTY;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long mf = __ballot(is_free), mm = __ballot(is_move);
    unsigned bf = 0, bm = 0;
    if (lane == 0) {
        if (mf) bf = atomicAdd(c.counts + 0, (unsigned)__popcll(mf));
        if (mm) bm = atomicAdd(c.counts + 1, (unsigned)__popcll(mm));
    }
    bf = (unsigned)__shfl((int)bf, 0, 64);
    bm = (unsigned)__shfl((int)bm, 0, 64);
    if (is_free) c.free_list[bf + (unsigned)__popcll(mf & below)] = (unsigned)slot;
    if (is_move) c.move_list[bm + (unsigned)__popcll(mm & below)] = (unsigned)slot;
}

OPTIK_DEV void compact_move_body(const CompactArgs &c) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.counts[1]) return;
    const size_t src = c.move_list[i], dst = c.free_list[i];
    for (int p = 0; p < c.nd; ++p) ENG_D_AT(c.d, c.nd, c.C, p, dst) = ENG_D_AT(c.d, c.nd, c.C, p, src);
    for (int p = 0; p < c.ni; ++p) c.i32[(size_t)p * c.C + dst] = c.i32[(size_t)p * c.C + src];
    c.item[dst] = c.item[src];
    if (c.i32[src] == ST_NNLS) {
        for (int k = 0; k < c.rec_len; ++k) c.nn_prob[dst * c.rec_len + k] = c.nn_prob[src * c.rec_len + k];
        c.nn_meta[dst * 2] = c.nn_meta[src * 2];
        c.nn_meta[dst * 2 + 1] = c.nn_meta[src * 2 + 1];
        if (c.nn_y)
            for (int k = 0; k < c.ny; ++k) c.nn_y[dst * c.ny + k] = c.nn_y[src * c.ny + k];
        if (c.nn_meta[src * 2 + 1] < 0.0)  // suspended solve: matrix and state live in the carry record
            for (int k = 0; k < NN_CARRY; ++k) c.nn_carry[dst * NN_CARRY + k] = c.nn_carry[src * NN_CARRY + k];
        c.nn_cls[dst - c.slot_base] = c.nn_cls[src - c.slot_base];
        c.nn_cls[src - c.slot_base] = NN_NONE;
    }
    c.i32[src] = ST_EMPTY;
}

// ---- kernel 4: finish the deferred directions with the NNLS answers ----------------

template <int N>
OPTIK_DEV void eng_finish_body(const EngArgs &a, const ChainDev &ch, size_t slot, size_t local, bool in_range) {
    using E = EngLayout<N>;
    // refill: lane i of the sub-pool takes the i-th listed slot and the next work item of the
    // queue (one atomic per wave)
    {
        const unsigned n_refill = *a.refill_count;
        bool want = local < n_refill;
        const size_t rslot = want ? (size_t)a.refill_list[local] : 0;
        while (wave_any(want)) {
            const unsigned long long it = fetch_items(a.next_item, want);
            if (want) want = !refill_slot<N>(a, ch, rslot, it);
        }
    }
    // the answered problems: each lane reads its record straight from HBM (staging the wave's
    // block in LDS as the update kernel does costs more here than the scattered reads: measured
    // 0.067 against 0.039 ms per launch)
    int code = 0;
    bool need = in_range && ENG_I(E::STATE) == ST_NNLS;
    if (need) {
        code = (int)a.nn_meta[slot * 2];
        need = (code & 7) != NNLS_SUSPENDED;  // else still being solved (listed for the next trip)
    }
    const RecIo<N> rec{a.nn_prob + slot * rec_stride<N>(), nullptr};
    if (need) {
        double l[E::NL], g[N], x[N], s[N];
        const double f = ENG_D(E::FC, 0);
        int ireset = ENG_I(E::IRESET), iter = ENG_I(E::ITER);
#pragma unroll
        for (int i = 0; i < E::NL; ++i) l[i] = ENG_D(E::L, i);
#pragma unroll
        for (int i = 0; i < N; ++i) { x[i] = ENG_D(E::X, i); g[i] = ENG_D(E::G, i); s[i] = 0.0; }
        double h3 = 0.0;
        int32_t status = 0;
        const int passes = code >> 3;
        ENG_I(E::NNIT) = passes;
        const int out = direction_search<N>(a, ch, slot, a.parity ^ 1, true, l, g, x, f, ireset, iter, false, s, h3,
                                            status, passes, rec);
        if (out == DIR_OK) store_direction<N, false>(a, ch, slot, l, g, x, s, f, h3, ireset, iter);
        else if (out == DIR_DEFER) store_deferred<N, false>(a, slot, l, g, ireset, iter);
        else { ENG_I(E::STATUS) = status; ENG_I(E::ITER) = iter; ENG_I(E::STATE) = ST_DEAD; }
    }
}

// Lists the slots of [0, n_slots) that still hold a restart (one atomic per wave).
OPTIK_DEV void tail_list_body(const int32_t *state, unsigned long long n_slots, unsigned int *count,
                              unsigned int *list) {
    const unsigned long long slot = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int st = slot < n_slots ? state[slot] : ST_EMPTY;
    const bool live = st != ST_EMPTY && st != ST_REFILL;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned long long m = __ballot(live);
    unsigned base = 0;
    if (lane == 0 && m) base = atomicAdd(count, (unsigned)__popcll(m));
    base = (unsigned)__shfl((int)base, 0, 64);
    if (live) list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (unsigned)slot;
}

}  // namespace optik
