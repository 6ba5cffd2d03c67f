// ik_lane64.hpp -- the restart solver with one restart per LANE and the wave's bounded sub-problems pipelined,
// in class order, through its sixteen quads of lanes.
//
// Where the quad solver (ik_quad.hpp) spends a wave's instructions (MI355X, cost of every phase by
// duplication, profiles/r4a_defer_experiment.txt): every scalar of Kraft's / NLopt's state machine, the
// chain product, the error terms and the log maps are REPLICATED in the four lanes of a restart
// (0.73 Mflop per restart at the wave level against 0.39 for a per-lane formulation), and a wave runs the
// Lawson-Hanson loop of the dual problem until the slowest of its sixteen problems is done (4.7 - 5.4 loop
// trips per call against 3.1 per problem) -- the sixteen are whatever the wave's restarts happen to need.
//
// Here a wave holds SIXTY-FOUR restarts, one per lane, and runs them phase by phase:
//
//   evaluation, NLopt bookkeeping, BFGS, LSQ factor,   per lane on the lane's own restart: no replication, no
//   rows of E^-1, LDP tail, back-substitution          cross-lane traffic; the building blocks of ik_slsqp.hpp /
//                                                      ik_eval.hpp
//   bounded dual problems (NNLS)                       the lanes that need one (~30 of 64 per trip) leave the
//                                                      packed problem (rows of E^-1, h: 42 doubles) in LDS and run
//                                                      Lawson-Hanson's FIRST pass on it themselves
//                                                      (ik_nnls_first.hpp): 45 % of the problems end there, the
//                                                      others start in a quad after that pass; the
//                                                      wave ranks those by predicted pass count (what the restart's
//                                                      previous problem took, or the number of violated bounds if
//                                                      larger: 76 % repeat it, 92 % within one) and PIPELINES them
//                                                      through its sixteen quads: a quad expands the next problem of
//                                                      the ranking into its 1 KB block and runs ik_nnls_quad.hpp on it;
//                                                      whenever fourteen or fewer quads are still solving, the finished
//                                                      ones' answers go back to their owner lanes and the idle quads
//                                                      take the next problems (Lane64Pipe).  A problem that needs more
//                                                      passes than predicted keeps its quad busy, not the wave.
//
// The price: the per-lane state (~80 doubles) plus the working set of an evaluation need more than 256
// registers, and 16 blocks + 64 packed problems fill 38 KB of LDS: ONE wave per SIMD (four per CU).
//
// Bit-exactness: the per-lane blocks execute the oracle's operation sequence ; the NNLS is ik_nnls_quad.hpp's; the LDP tail is lsq_dual's.  A direction that is not a descent
// direction (Kraft: reset B and search again, 5e-5 of the trips) repeats in the wave's NEXT trip instead of a
// second pass of this one -- the lane sits out one evaluation, its arithmetic is the same.
//
// Restates, per lane: /root/reference/crates/optik/src/lib.rs:301-391 (the restart closure) with NLopt's SLSQP
// (un-vendored; oracle/optik_oracle.c is the CPU statement of the same arithmetic).
#pragma once

#include "ik_lane.hpp"
#include "ik_solve.hpp"
#include "ik_nnls_quad.hpp"

namespace optik {

template <int N>
struct Lane64Geom {
    static constexpr int NG = N * (N + 1) / 2;  // entries of E^-1 (upper triangular, by row)
    static constexpr int REC = NG + 2 * N;      // ... then h_lo[N], h_hi[N]
    // the record of lane p, value v, at rec[v * 64 + p]: a wave-wide store of "my v-th value" is 512 contiguous bytes
    OPTIK_DEV static constexpr int g(int i, int j) { return i * N - (i * (i - 1)) / 2 + (j - i); }  // E^-1(i, j), j >= i
    OPTIK_DEV static constexpr int hlo(int i) { return NG + i; }
    OPTIK_DEV static constexpr int hhi(int i) { return NG + N + i; }
    // where a quad leaves {mode, rnorm, solve passes} of its problem: the spare doubles of its block
    static constexpr int META = NnlsQuadGeom<N>::XS + 2 * N;
    static_assert(NnlsQuadGeom<N>::XS + 2 * N + 3 <= NnlsQuadGeom<N>::STRIDE, "three spare doubles per block");
};

}  // namespace optik
#include "ik_nnls_first.hpp"
namespace optik {

#ifndef OPTIK_LANE_FIRST_PASS_MIN
#define OPTIK_LANE_FIRST_PASS_MIN 16  // (with the warm start, 0 / 8 / 16 / 24: Panda 32.10 / 32.29 / 32.30 / 32.16 M, UR10 54.7 / 54.3 / 54.8 / 53.9 M)
#endif
#ifndef OPTIK_LANE_WARM_START
#define OPTIK_LANE_WARM_START 1   // a problem the first pass does not end starts in its quad AFTER that pass (the quad re-forms the state)
#endif
#ifndef OPTIK_LANE_FIRST_PASS
#define OPTIK_LANE_FIRST_PASS 1   // every lane runs the first NNLS pass on its own problem; only the unsolved ones go through the quads
#endif

// doubles of LDS per wave: the NNLS blocks of its 16 quads (+ the column of zeros), the 64 packed problems
template <int N>
constexpr int lane64_block_lds() { return nnls_quad_wave_lds<N>(); }
template <int N>
constexpr int lane64_rec_lds() { return Lane64Geom<N>::REC * 64; }

#ifndef OPTIK_LANE_CLASSES
#define OPTIK_LANE_CLASSES 8
#endif
constexpr int LANE64_CLASSES = OPTIK_LANE_CLASSES;  // predicted pass classes 1 .. 7 (0: no bounded problem this trip)
#ifndef OPTIK_LANE_PIPE
#define OPTIK_LANE_PIPE 1         // 0: rounds of sixteen problems, each round to its end (comparisons)
#endif
#ifndef LANE64_MAX_RUNNING
#define LANE64_MAX_RUNNING 14     // hand-over as soon as this few of the sixteen quads are still solving (and problems wait).  Since the first pass
                                  // runs per lane a call seldom has more problems than quads: 6: -1 %, 10: -0.4 %, 13 .. 15: the same
#endif
// (refill, tools/ab_kernel_path.sh, M restarts/s at the driver's command, three interleaved runs each: neither 32.38,
// reserve 8 alone 32.52, kept target alone 32.41, both 32.79, reserve 16 + kept target 32.79 -- profiles/r5i_ab_refill.txt;
// 64 items per atomic handed out over many refills: 31.8, the waves that still hold items when the queue is dry end late)
#ifndef OPTIK_LANE_RESERVE
#define OPTIK_LANE_RESERVE 8      // work items a wave draws AHEAD of the refill that hands them out (0: every refill waits for its own atomic)
#endif
#ifndef OPTIK_LANE_KEEP_TARGET
#define OPTIK_LANE_KEEP_TARGET 1  // 1: a lane whose next restart has the target of its last one keeps the pose it holds
#endif
#ifndef OPTIK_LANE_REFILL
#define OPTIK_LANE_REFILL 4       // idle lanes a wave accumulates before it refills (the seed generation runs for the whole wave)
                                  // (with the reserve, round 5: 2 / 3 / 4 / 6 idle lanes 32.86 / 32.84 / 32.82 / 32.56 M: profiles/r5m_ab_refill_threshold.txt)
#endif

// The wave's problems in rank order through its sixteen quads (ik_nnls_quad.hpp: Pipe): what lane64_wave hands to the
// NNLS -- the answers of finished quads back to their owner lanes, the next problems into the idle quads.
template <int N, class Expand, class ReadBack>
struct Lane64Pipe {
    static constexpr bool on = true;
    static constexpr bool warm_start = true;
    static constexpr int MAX_RUNNING = LANE64_MAX_RUNNING;
    int n_prob, next, hold;  // problems, the next rank to hand out (wave-uniform), the rank the quad holds or -1
    int qi, ql, lane, rank;
    bool has, got;
    double *bk;
    int *lor, *where;
    Expand *expand_fn;
    ReadBack *read_fn;
    OPTIK_DEV bool more() const { return next < n_prob; }
    // (warm start, ik_nnls_first.hpp: wst receives Q e_m, the pivot weight and the multiplier of a problem whose first
    // column is already in, wj that column's id -- 0: the problem starts at step two)
    template <bool WARM>
    OPTIK_DEV bool event(bool idle, int mode, double rn, int passes, double *wst, int &wj) {
        wj = 0;
        const bool fin = idle && hold >= 0;
        if (wave_any(fin)) {
            // (the multipliers are in the block by column id; mode, rnorm and the pass count next to them)
            if (fin && ql == 0) {
                bk[Lane64Geom<N>::META] = (double)mode;
                bk[Lane64Geom<N>::META + 1] = rn;
                bk[Lane64Geom<N>::META + 2] = (double)passes;
                where[hold] = 0x100 | qi;
            }
            lds_sync();
            if (has && !got) {
                const int w = where[rank];
                if (w & 0x100) { (*read_fn)(w & 0xff); got = true; }
            }
            lds_sync();  // (before the blocks are rewritten)
            if (fin) hold = -1;
        }
        const bool free_q = idle && hold < 0;
        const unsigned long long fm = __ballot(free_q);
        const int mine = (int)__popcll(fm & ((1ull << (lane & ~3)) - 1ull)) / QUAD;
        const int pr = next + mine;
        const bool live = free_q && pr < n_prob;
        next += (int)__popcll(fm) / QUAD;
        next = next < n_prob ? next : n_prob;
        if (wave_any(live)) {
            const int p = lor[live ? pr : 0];
            (*expand_fn)(std::integral_constant<bool, WARM>{}, live, p, wst, wj);
            if (live) hold = pr;
        }
        return live;
    }
};

template <int N, bool TIP>
OPTIK_DEV void lane64_wave(const ChainDev &ch, const EvalParams &ep, const SolveParams &sp, const uint32_t (&key)[8],
                           const double (&scale)[MAX_DOF], const WorkQueue &wq,
                           double *nnls_lds /* lane64_block_lds<N>() doubles, the last 16 zero */,
                           double *rec_lds /* lane64_rec_lds<N>() doubles */, int *lor_lds /* 64 ints: lane of rank */,
                           int *where_lds /* 64 ints: by rank, 0x100 | the quad whose block holds the problem's answer */) {
    typedef Lane64Geom<N> G;
    constexpr int NL = N * (N + 1) / 2;
    constexpr int NS = (N > 4) ? 2 : 1;
    constexpr int CS = NnlsQuadGeom<N>::CS;
    static_assert(N <= 7, "the throughput form of chains with at most seven joints");
    const double alfmin = 0.1;
    const int lane = (int)(threadIdx.x & 63u);
    // SLSQP state of the lane's restart (names as in the oracle)
    double x[N], x0[N], g[N], s[N], l[NL];
    double xbest[N], xprev[N];
    double f = 0.0, f0 = 0.0, t0 = 0.0, h3 = 0.0, alpha = 1.0;
    double minf = __builtin_huge_val(), fprev = __builtin_huge_val();
    int ireset = 0, line = 0, nevals = 0;
    int pred = 1;          // solve passes of the restart's previous bounded problem
    bool first = true;
    bool again = false;    // the last direction was not a descent direction: reset B and search again, no evaluation
    Pose target;
    unsigned long long item = 0, index = 0;
    unsigned tslot = OPTIK_LANE_KEEP_TARGET ? ~0u : 0u;  // (~0: the lane holds no target pose yet)
    bool active = false, want = lane < wq.lanes;
#if OPTIK_LANE_RESERVE > 0
    unsigned long long rbase = 0, rend = 0;  // (wave-uniform) the items the wave has drawn and not handed out yet
    unsigned long long pend = 0;             // (lane 0) the base of the reserve that is on its way
    bool r_pending = false;
    bool dry = false;                        // (wave-uniform) an item past the launch's last one has been handed out
#endif
#pragma unroll
    for (int i = 0; i < N; ++i) { x[i] = 0.0; x0[i] = 0.0; g[i] = 0.0; s[i] = 0.0; xbest[i] = 0.0; xprev[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < NL; ++i) l[i] = 0.0;
    target.t = V3{0, 0, 0};
    target.q = Q4{0, 0, 0, 1};

    // (-DOPTIK_PROFILE: wave cycles per part of a trip -- 0 refill, 1 evaluation, 4 bookkeeping + BFGS, 5 LSQ factor +
    // records, 2 first NNLS pass, 6 ranking + NNLS, 3 LDP tail .. publish, 7 trips; phase_profile.py (a rounds 3-5 tool: git history) lane)
#ifdef OPTIK_PROFILE
    unsigned long long lp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long lt_ = __builtin_readcyclecounter();
#define LANE_PROF(slot) do { const unsigned long long n_ = __builtin_readcyclecounter(); lp_[slot] += n_ - lt_; lt_ = n_; } while (0)
#else
#define LANE_PROF(slot)
#endif
    for (;;) {
        // ---- refill: lanes without a restart pull the next work item ----------------------------------
        // (the seed generation runs for the whole wave: wait until several lanes are idle -- or none is busy)
        const unsigned n_want = (unsigned)__popcll(__ballot(want));
        if (n_want >= (unsigned)(wq.lanes < OPTIK_LANE_REFILL ? wq.lanes : OPTIK_LANE_REFILL) || (n_want > 0 && !wave_any(active))) {
#if OPTIK_LANE_RESERVE > 0
            // The wave keeps a RESERVE of work items: the atomic on the launch's one counter (a device-scope round trip of
            // microseconds, with nothing to cover it at one wave per SIMD) was issued a trip or more ago, right after the
            // previous refill; only a refill that wants more than the reserve holds waits for a second, synchronous one.
            unsigned long long it;
            {
                if (r_pending) { rbase = __shfl(pend, 0); rend = rbase + OPTIK_LANE_RESERVE; r_pending = false; }
                const unsigned long long wm = __ballot(want);
                const unsigned rank = (unsigned)__popcll(wm & ((1ull << lane) - 1ull));
                const unsigned long long have = rend - rbase;
                const unsigned long long take = n_want < have ? n_want : have;
                it = rbase + rank;
                rbase += take;
                if (wave_any(want && rank >= take)) {
                    const unsigned long long it2 = fetch_items(wq.next_item, want && rank >= take);
                    if (rank >= take) it = it2;
                }
                dry = dry || wave_any(want && it >= wq.total_items);
            }
#else
            const unsigned long long it = fetch_items(wq.next_item, want);
#endif
            if (want) {
                want = false;
                if (it < wq.total_items) {
                    unsigned long long r;
#if OPTIK_LANE_KEEP_TARGET
                    const unsigned tprev = tslot;
#endif
                    if (wq.restart_major) { r = it / wq.n_targets; tslot = (unsigned)(it - r * wq.n_targets); }
                    else { tslot = (unsigned)(it / wq.n_restarts); r = it - (unsigned long long)tslot * wq.n_restarts; }
                    item = (unsigned long long)tslot * wq.n_restarts + r;  // output column
                    index = wq.restart_begin + r;
#if OPTIK_LANE_KEEP_TARGET
                    // (a target-major launch: nearly every lane's next restart has the target of its last one)
                    if (tslot != tprev) target = load_pose(wq.targets + (size_t)tslot * 7);
#else
                    target = load_pose(wq.targets + (size_t)tslot * 7);
#endif
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
                    ireset = 0; line = 0; nevals = 0; pred = 1;
                    first = true;
                    again = false;
                    active = true;
                }
            }
        }
#if OPTIK_LANE_RESERVE > 0
        // (the next refill's items are asked for NOW -- a trip or more before they are handed out)
        if (!r_pending && rbase == rend && !dry) {
            if (lane == 0) pend = atomicAdd(wq.next_item, (unsigned long long)OPTIK_LANE_RESERVE);
            r_pending = true;
        }
#endif
        if (!wave_any(active)) break;
        LANE_PROF(0);

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
        const bool stepping = active && ret == 0;
        const bool do_eval = stepping && !again;
        double gn[N];
        double fn = 0.0;
        OPTIK_SCHED_FENCE_LANE64();
        if (do_eval) fn = eval_fg<N, TIP>(ch, ep, target, x, gn);
        OPTIK_SCHED_FENCE_LANE64();
        LANE_PROF(1);

        // ---- NLopt bookkeeping and Kraft's line search (labels 100 / 220), per lane ---------------------
        bool need_dir = stepping && again, reset = stepping && again;
        again = false;
        if (do_eval) {
            f = fn;
            ++nevals;
            // NLopt: update best point so far; stopval is tested after every evaluation
            if (f < minf) {
                minf = f;
#pragma unroll
                for (int i = 0; i < N; ++i) xbest[i] = x[i];
            }
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
                    // line search complete (mode -1): NLopt re-evaluates f and the gradient there unless the
                    // accepted trial was the first one
                    if (line > 1) ++nevals;
                    if (!__builtin_isinf(fprev)) {
                        if (__builtin_fabs(f - fprev) < sp.ftol_abs) ret = RES_FTOL_REACHED;
                        else if (xprev_live(sp) && stop_x<N>(sp, x, xprev)) ret = RES_XTOL_REACHED;
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
                        OPTIK_SCHED_FENCE_LANE64();
                        bfgs_update<N>(l, s, u);
                        OPTIK_SCHED_FENCE_LANE64();
                        need_dir = true;
                    }
                }
            }
        }
        OPTIK_SCHED_FENCE_LANE64();

        LANE_PROF(4);
        // ---- labels 110/130: (reset,) search direction, descent test -- one pass per trip -----------------
        if (wave_any(need_dir)) {
            if (need_dir && reset) {
                ++ireset;
                if (ireset > 5) {
                    // label 255 with acc = 0 -> mode 8; NLopt's relaxed test vs (f0, x0)
                    ret = RES_ROUNDOFF_LIMITED;
                    if (__builtin_fabs(f - f0) < sp.ftol_abs && !__builtin_isinf(f0)) ret = RES_FTOL_REACHED;
                    else if (stop_x<N>(sp, x, x0)) ret = RES_XTOL_REACHED;
                    need_dir = false;
                } else {
#pragma unroll
                    for (int i = 0; i < NL; ++i) l[i] = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) l[lidx<N>(i, i)] = 1.0;
                }
            }
            double E[N][N], fv[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                fv[i] = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) E[i][j] = 0.0;
            }
            int lmode = lsq_factor<N>(l, g, E, fv);
            OPTIK_SCHED_FENCE_LANE64();
            // rows of E^-1 and the bound rows they give: into the lane's packed problem in LDS
            int nviol = 0;
            bool need_nnls;
            {
                double lo[N], hi[N];
#pragma unroll
                for (int i = 0; i < N; ++i) { lo[i] = ch.lb[i] - x[i]; hi[i] = ch.ub[i] - x[i]; }
                double *const rp = rec_lds + lane;
                need_nnls = lsq_bound_rows<N>(E, fv, lo, hi, [&](int i, const double (&row)[N], double h_lo, double h_hi) {
#pragma unroll
                    for (int j = i; j < N; ++j) rp[64 * G::g(i, j)] = row[j];
                    rp[64 * G::hlo(i)] = h_lo;
                    rp[64 * G::hhi(i)] = h_hi;
                    nviol += (h_lo > 0.0 ? 1 : 0) + (h_hi > 0.0 ? 1 : 0);
                });
            }
            const bool has_any = need_dir && lmode == 1 && need_nnls;
            // 45 % of the bounded problems end after ONE pass with one active bound: that pass per lane, on the lane's own
            // record (ik_nnls_first.hpp: the same numbers as the quads would form); the rest through the quads
            bool solved1 = false, warm1 = false;  // (warm: the first column is in, a quad continues from there)
            int y1_id = 1;
            double y1_val = 0.0, rn1 = 1.0;
            LANE_PROF(5);
#if OPTIK_LANE_FIRST_PASS
            OPTIK_SCHED_FENCE();  // (a phase of its own: interleaved with its neighbours it costs them their registers)
            // (only when the wave has more problems than quads: with fewer, a call is as long as its longest problem
            // whatever is taken out of it, and the pass costs the whole wave ~800 instructions)
            if ((int)__popcll(__ballot(has_any)) > OPTIK_LANE_FIRST_PASS_MIN) {
                if (has_any) {
                    const int fp = nnls_first_pass<N>(rec_lds + lane, y1_id, y1_val, rn1);
                    solved1 = fp == FIRST_SOLVED;
                    warm1 = OPTIK_LANE_WARM_START && fp == FIRST_WARM;
                }
            }
            OPTIK_SCHED_FENCE();
#endif
            const bool has = has_any && !solved1;
            OPTIK_SCHED_FENCE_LANE64();
            LANE_PROF(2);

            // ---- the wave's bounded problems by predicted class, the largest first -----------------------
            int cls = 0;
            if (has) {
                cls = nviol > pred ? nviol : pred;
                // (after the first pass: one pass done + one per dual still positive; +0.9 %, three of three runs)
                if (warm1) { const int c2 = 1 + (int)y1_val; cls = c2 > cls ? c2 : cls; }
                cls = cls < 1 ? 1 : (cls > LANE64_CLASSES - 1 ? LANE64_CLASSES - 1 : cls);
            }
            int rank = 0, n_prob = 0;
            {
                const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
                for (int c = LANE64_CLASSES - 1; c >= 1; --c) {
                    const unsigned long long mc = __ballot(cls == c);
                    if (cls == c) rank = n_prob + (int)__popcll(mc & below);
                    n_prob += (int)__popcll(mc);
                }
            }
            if (has) lor_lds[rank] = lane | (warm1 ? (y1_id << 8) : 0);  // (lane of rank; the column that is already in, or 0)
            where_lds[lane] = 0;  // (no problem of this trip is solved yet)
            lds_sync();

            double y[2 * N];
#pragma unroll
            for (int r = 0; r < 2 * N; ++r) y[r] = 0.0;
            int nmode = 1;
            double rnorm = 1.0;
            const int nq = wave_quads();  // (16; the emulation of tests/emu runs partial waves)
            const int qi = lane >> 2, ql = lane & 3;
            // quad qi's columns of the problem of lane p: from the owner's packed record into the quad's block
            int ids[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sl = k & 1, rr = ql + 4 * sl;
                ids[k] = (sl < NS && rr < N) ? ((k >= 2) ? N : 0) + rr + 1 : 0x7fff;
            }
            double *const bk = nnls_lds + (unsigned)qi * NnlsQuadGeom<N>::STRIDE;
            // (what a warm start hands to nnls_quad's state: Q e_m, the transformation's pivot weight, the multiplier and
            // the id of the column that is in -- quad-uniform; wj = 0: a cold start)
            // (wst: N + 1 doubles of Q e_m, then up, then the multiplier; wj: the column that is in, 0 for a cold start --
            // locals of nnls_quad's hand-over, not carried through its loop)
            auto expand = [&](auto warm_tag, bool live_any, int enc, double *wst, int &wj) {
                constexpr bool WARM = decltype(warm_tag)::value;
                const int p = enc & 0xff;
                const double *const rq = rec_lds + p;
                const int jw = (WARM && live_any) ? (enc >> 8) : 0;
                wj = live_any ? jw : wj;
                if (WARM && wave_any(live_any && jw != 0)) {
                    // warm: column jw is in (ik_nnls_first.hpp: the owner lane's own first pass, formed again here from
                    // the same record -- the same numbers), the quad's block gets the columns as that pass leaves them
                    const bool wq = live_any && jw != 0;
                    OPTIK_SCHED_FENCE();
                    double w[N + 1], bq[N + 1], up, ulp, hb, yv;
                    bool al;
                    (void)first_pass_column<N>(rq, wq ? jw : 1, w, bq, up, ulp, hb, al, yv);
                    if (wq) {
                        wst[N + 1] = up;
                        wst[N + 2] = yv;
#pragma unroll
                        for (int r = 0; r <= N; ++r) wst[r] = bq[r];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int sl = k & 1;
                        if (sl < NS && ql + 4 * sl < N) {
                            double nv[N + 1];
                            first_pass_other_column<N>(rq, ids[k], w, hb, al, nv);
                            OPTIK_SCHED_FENCE();  // (a column at a time)
                            if (wq) {
                                double *c = bk + CS * (ids[k] - 1);
                                const bool isj = ids[k] == jw;
#pragma unroll
                                for (int r = 0; r <= N; ++r) c[r] = isj ? ((r == 0) ? ulp : 0.0) : nv[r];
                            }
                        }
                    }
                    OPTIK_SCHED_FENCE();
                }
                const bool live = live_any && jw == 0;
                // (row rr of E^-1 goes out twice: as the lower bound's column and, negated, as the upper bound's --
                // one read of every entry, two stores)
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const int rr = ql + 4 * sl;
                    if (live && rr < N) {
                        double *clo = bk + CS * (ids[sl] - 1), *chi = bk + CS * (ids[2 + sl] - 1);
                        // (row rr of E^-1 is zero before column rr; entry (rr, j) of the packed triangle otherwise)
                        const int tri = rr * N - (rr * (rr - 1)) / 2 - rr;  // G::g(rr, j) - j
#pragma unroll
                        for (int j = 0; j < N; ++j) {
                            const double e = rq[64 * (j >= rr ? tri + j : 0)];
                            const double v = (j >= rr) ? e : 0.0;
                            clo[j] = v;
                            chi[j] = (j >= rr) ? -v : 0.0;
                        }
                        clo[N] = rq[64 * (G::NG + rr)];
                        chi[N] = rq[64 * (G::NG + N + rr)];
                    }
                }
            };
            // the owner of a solved problem reads its answer back from the block of the quad that solved it
            auto read_back = [&](int q) {
                const double *ob = nnls_lds + (unsigned)q * NnlsQuadGeom<N>::STRIDE;
#pragma unroll
                for (int r = 0; r < 2 * N; ++r) y[r] = ob[NnlsQuadGeom<N>::XS + r];
                nmode = (int)ob[G::META];
                rnorm = ob[G::META + 1];
                pred = (int)ob[G::META + 2];
            };
#if OPTIK_LANE_PIPE
            {
                // The wave's problems in rank order through its sixteen quads, the rounds overlapping: whenever at most
                // LANE64_MAX_RUNNING quads are still solving, the answers of the finished ones go back to their owners
                // and the idle quads take the next problems -- a problem that needs more passes than its class
                // predicted keeps ITS quad busy, not the wave (ik_nnls_quad.hpp: Pipe).
                typedef Lane64Pipe<N, decltype(expand), decltype(read_back)> Pipe;
                Pipe pipe{n_prob, 0, -1, qi, ql, lane, rank, has, false, bk, lor_lds, where_lds, &expand, &read_back};
                int iters, qmode;
                double xv[4], qrnorm;
                nnls_quad<N, Pipe>(false, ids, bk, nnls_lds + 16 * NnlsQuadGeom<N>::STRIDE, xv, qmode,
                                                            qrnorm, iters, &pipe);
            }
#else
            for (int r0 = 0; r0 < n_prob; r0 += nq) {
                // quad qi takes the problem of rank r0 + qi
                const int pr = r0 + qi;
                const bool live = pr < n_prob;
                double wdummy[N + 3];
                int wjd = 0;
                expand(std::false_type{}, live, lor_lds[live ? pr : 0] & 0xff, wdummy, wjd);  // (rounds: every problem from step two)
                int iters, qmode;
                double xv[4], qrnorm;
                nnls_quad<N>(live, ids, bk, nnls_lds + 16 * NnlsQuadGeom<N>::STRIDE, xv, qmode, qrnorm, iters);
                if (live && ql == 0) {
                    bk[G::META] = (double)qmode;
                    bk[G::META + 1] = qrnorm;
                    bk[G::META + 2] = (double)iters;
                }
                lds_sync();
                // the owners of this round's problems read their answers back
                if (has && rank >= r0 && rank < r0 + nq) read_back(rank - r0);
                lds_sync();  // (before the next round rewrites the blocks)
            }
#endif
            OPTIK_SCHED_FENCE_LANE64();
            LANE_PROF(6);

            // ---- LDP tail (lsq_dual), back-substitution, descent test, per lane ------------------------------
            double sn[N];
#pragma unroll
            for (int j = 0; j < N; ++j) sn[j] = 0.0;
            if (solved1) {
                // (what a quad would have handed back: one multiplier, mode 1, one pass)
#pragma unroll
                for (int r = 0; r < 2 * N; ++r) y[r] = (r == y1_id - 1) ? y1_val : 0.0;
                nmode = 1;
                rnorm = rn1;
                pred = 1;
            }
            if (has_any) {
                const double *const rp = rec_lds + lane;
                int mode = nmode;
                if (mode == 1 && rnorm <= 0.0) mode = 4;
                if (mode == 1) {
                    double hy = 0.0;
#pragma unroll
                    for (int r = 0; r < 2 * N; ++r) hy += rp[64 * (r < N ? G::hlo(r) : G::hhi(r - N))] * y[r];
                    double fac = 1.0 - hy;
                    const double d1 = 1.0 + fac;
                    if (d1 - 1.0 <= 0.0) mode = 4;
                    else {
                        fac = 1.0 / fac;
#pragma unroll
                        for (int j = 0; j < N; ++j) {
                            double acc = 0.0;
#pragma unroll
                            for (int r = 0; r <= j; ++r) acc += rp[64 * G::g(r, j)] * y[r];
#pragma unroll
                            for (int r = 0; r <= j; ++r) acc += (-rp[64 * G::g(r, j)]) * y[N + r];
                            sn[j] = fac * acc;
                            OPTIK_SCHED_FENCE_LANE64();
                        }
                    }
                }
                lmode = mode;
            }
            {
                double lo[N], hi[N];
#pragma unroll
                for (int i = 0; i < N; ++i) { lo[i] = ch.lb[i] - x[i]; hi[i] = ch.ub[i] - x[i]; }
                lsq_finish<N>(E, fv, lo, hi, sn);
            }
            OPTIK_SCHED_FENCE_LANE64();
            if (need_dir) {
                if (lmode != 1) {
                    // NLopt: modes 5,6,7 -> ROUNDOFF_LIMITED; 3,4,9 -> FAILURE
                    ret = (lmode == 5 || lmode == 6 || lmode == 7) ? RES_ROUNDOFF_LIMITED : RES_FAILURE;
                } else {
                    // (g is also Kraft's v: the gradient at the start of the line search)
                    double gs = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) { s[i] = sn[i]; x0[i] = x[i]; }
                    f0 = f;
#pragma unroll
                    for (int i = 0; i < N; ++i) gs += g[i] * s[i];
                    t0 = f;
                    h3 = gs;  // h3 = gs - h1 * h4 with h1 = 0 (no constraints)
                    if (h3 >= 0.0) {
                        again = true;  // not a descent direction: reset B and search again (next trip)
                    } else {
                        line = 0;
                        alpha = 1.0;
                    }
                }
            }
            lds_sync();  // (the records are read: the next trip may rewrite them)
        }
        OPTIK_SCHED_FENCE_LANE64();
        if (stepping && ret == 0 && !again) {
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
        // ---- a restart ended: classify (lib.rs:376-379), publish, free the lane ---------------------------
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
            again = false;
        }
#ifdef OPTIK_PROFILE
        LANE_PROF(3);
        lp_[7] += 1;
#endif
    }
#ifdef OPTIK_PROFILE
    if (wq.prof && (threadIdx.x & 63u) == 0)
        for (int i_ = 0; i_ < 8; ++i_) atomicAdd(wq.prof + i_, lp_[i_]);
#endif
#undef LANE_PROF
}

}  // namespace optik
