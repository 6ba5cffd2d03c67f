// ik_coop.hpp -- the latency-oriented restart solver: one restart per GROUP of four lanes.
//
// The single-kernel solver of round 1 (ik_solve.hpp: one restart per lane, Lawson-Hanson NNLS
// on a per-lane matrix in LDS) spends half of every iteration in that NNLS even when a wave
// holds a single restart (tools/phase_profile.py, 512 restarts: 67 k of 130 k cycles per
// iteration; 80 KB of LDS per wave; 600 B of scratch).  Here the lane that owns a restart
// (the leader of its group) runs the same per-lane SLSQP step -- evaluation, NLopt bookkeeping,
// line search, BFGS update, LSQ factor, bound rows, LDP tail, back-substitution: the device
// functions of ik_eval.hpp / ik_slsqp.hpp the other paths call -- and the bounded dual problem
// is solved by the whole group with the register-resident cooperative NNLS of the streaming
// engine (ik_nnls_coop.hpp): the leader leaves the problem record in a 512-byte LDS window of
// its group, the four lanes take four columns each, the multipliers come back through the
// window.  No per-lane LDS matrix, 18 KB of LDS per wave, and up to 16 restarts per wave.
//
// Same operations on the same operands in the same order as every other path: results are
// bit-identical to the CPU oracle (tests/test_gpu_parity.py runs this kernel as path "kernel").
#pragma once

#include "ik_engine.hpp"

namespace optik {

constexpr int COOP_GROUP = COOP_COLS / 4;           // lanes per restart (4 columns per lane)
constexpr int COOP_GROUPS_PER_WAVE = 64 / COOP_GROUP;
#ifndef OPTIK_COOP_REFILL_BATCH
#define OPTIK_COOP_REFILL_BATCH 1
#endif
constexpr unsigned COOP_REFILL_BATCH = OPTIK_COOP_REFILL_BATCH;  // idle groups a wave accumulates before it refills

// doubles of the group's LDS window: the problem record, the multipliers, {mode + 8 passes, rnorm}
// (parking the leader's LSQ factor here across the solve does not help: 396 against 372 B of scratch)
// ... then the evaluation's hand-over area: x and the target pose from the leader, the joint frames
// of the forward pass (N x 7), the gradient components coming back (eval_fg_group)
template <int N>
constexpr int coop_rec_lds() { return rec_stride<N>() + 2 * N + 2 + (N + 7 + 7 * N + N + 1) / 2 * 2; }

// Objective and gradient at the leader's x, computed by the four lanes of its group together:
// everyone walks the chain, each lane takes the Jacobian columns k = gl, gl + 4 (ik_eval.hpp:
// eval_fg_group).  All 64 lanes call; `do_eval` is the leader's flag.  fn / gn are valid in the leader.
template <int N, bool TIP>
OPTIK_DEV void coop_eval(const ChainDev &ch, const EvalParams &ep, bool do_eval, const Pose &target, const double (&x)[N],
                         double *grec, double &fn, double (&gn)[N]) {
    if (!wave_any(do_eval)) return;
    const unsigned lane = threadIdx.x & 63u;
    const int gl = (int)(lane % COOP_GROUP);
    double *const gx = grec + rec_stride<N>() + 2 * N + 2;
    double *const gt = gx + N;
    double *const gframes = gt + 7;
    double *const ggrad = gframes + 7 * N;
    if (do_eval) {
#pragma unroll
        for (int i = 0; i < N; ++i) gx[i] = x[i];
        gt[0] = target.t.x; gt[1] = target.t.y; gt[2] = target.t.z;
        gt[3] = target.q.i; gt[4] = target.q.j; gt[5] = target.q.k; gt[6] = target.q.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool grp = __shfl((int)do_eval, (int)(lane & ~(unsigned)(COOP_GROUP - 1)), 64) != 0;
    if (grp) {
        double xg[N];
#pragma unroll
        for (int i = 0; i < N; ++i) xg[i] = gx[i];
        const Pose tg = load_pose(gt);
        const double fg = eval_fg_group<N, TIP, COOP_GROUP>(ch, ep, tg, xg, gl, gframes, ggrad);
        fn = do_eval ? fg : fn;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (do_eval) {
#pragma unroll
        for (int i = 0; i < N; ++i) gn[i] = ggrad[i];
    }
}

// NLopt bookkeeping and Kraft's line search after an evaluation (labels 100 / 220 / 260): what
// solve_wave does between its evaluation and its direction search, for the lane `do_eval` holds.
template <int N>
OPTIK_DEV void coop_after_eval(const SolveParams &sp, bool do_eval, double fn, const double (&gn)[N], double (&x)[N],
                               double (&g)[N], double (&s)[N], double (&l)[N * (N + 1) / 2], double (&xbest)[N],
                               double (&xprev)[N], double &f, double &t0, double &h3, double &alpha, double &minf,
                               double &fprev, int &line, int &nevals, bool &first, int32_t &ret, bool &need_dir,
                               bool &reset) {
    const double alfmin = 0.1;
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
                // line search complete (mode -1): NLopt re-evaluates f and the gradient there
                // unless the accepted trial was the first one
                if (line > 1) ++nevals;
                if (!__builtin_isinf(fprev)) {
                    if (__builtin_fabs(f - fprev) < sp.ftol_abs) ret = RES_FTOL_REACHED;
                    // (xprev_live: when neither x test can fire after the ftol test the engine does not
                    // keep the previous iterate, and a slot handed over from it carries a stale one)
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
                    OPTIK_SCHED_FENCE();
                    bfgs_update<N>(l, s, u);
                    OPTIK_SCHED_FENCE();
                    need_dir = true;
                }
            }
        }
    }
}

// Labels 110/130: (reset,) search direction, descent test.  Wave-uniform loop: the leaders
// prepare their LSQ problems, the whole wave solves the bounded ones (one per group of four
// lanes, ik_nnls_coop.hpp), the leaders finish; a leader whose direction is not a descent
// direction resets and goes round again (at most five times, Kraft's ireset).
template <int N>
OPTIK_DEV void coop_direction(const ChainDev &ch, const SolveParams &sp, double *nnls_lds, double *grec, bool &need_dir,
                              bool &reset, int &ireset, double (&l)[N * (N + 1) / 2], const double (&g)[N],
                              const double (&x)[N], double (&x0)[N], double (&s)[N], double f, double &f0, double &t0,
                              double &h3, double &alpha, int &line, int32_t &ret, unsigned long long *nnls_cycles = nullptr) {
    constexpr int NL = N * (N + 1) / 2;
    constexpr int n = 2 * N;
    constexpr int CPL = 4;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned group = lane / COOP_GROUP, gl = lane % COOP_GROUP;
    double *const gy = grec + rec_stride<N>();
    double *const gmeta = gy + 2 * N;
    const RecIo<N> rec{grec, nullptr};
        while (wave_any(need_dir)) {
        bool pass = need_dir;
        if (pass && reset) {
            ++ireset;
            if (ireset > 5) {
                // label 255 with acc = 0 -> mode 8; NLopt's relaxed test vs (f0, x0)
                ret = RES_ROUNDOFF_LIMITED;
                if (__builtin_fabs(f - f0) < sp.ftol_abs && !__builtin_isinf(f0)) ret = RES_FTOL_REACHED;
                else if (stop_x<N>(sp, x, x0)) ret = RES_XTOL_REACHED;
                need_dir = false;
                pass = false;
            } else {
#pragma unroll
                for (int i = 0; i < NL; ++i) l[i] = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) l[lidx<N>(i, i)] = 1.0;
            }
        }
        double E[N][N], fv[N], lo[N], hi[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            lo[i] = ch.lb[i] - x[i];
            hi[i] = ch.ub[i] - x[i];
            fv[i] = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) E[i][j] = 0.0;
        }
        int lmode = 1;
        bool need_nnls = false;
        OPTIK_SCHED_FENCE();
        if (pass) {
            lmode = lsq_factor<N>(l, g, E, fv);
            if (lmode == 1) {
                need_nnls = lsq_bound_rows<N>(E, fv, lo, hi, [&](int i, const double (&row)[N], double h_lo, double h_hi) {
#pragma unroll
                    for (int r = i; r < N; ++r) rec.put(rec_g<N>(i, r), row[r]);
                    rec.put(rec_hlo<N>(i), h_lo);
                    rec.put(rec_hhi<N>(i), h_hi);
                });
            }
        }
        OPTIK_SCHED_FENCE();
        // ---- the bounded dual problems of this round, one per group, all 64 lanes ------
        if (wave_any(need_nnls)) {
            const unsigned long long t_nn = nnls_cycles ? __builtin_readcyclecounter() : 0ull;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool live = __shfl((int)need_nnls, (int)(lane & ~(unsigned)(COOP_GROUP - 1)), 64) != 0;
            dvec8 col[CPL];
            CoopCarry<CPL> cs;
            cs.b = 0.0;
            cs.up = 0.0;
            cs.nsetp = 0;
            cs.iter = 0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                col[k] = 0.0;
                cs.xv[k] = 0.0;
                cs.pos[k] = 0;
                const unsigned c = gl * CPL + k;
                if (live && c < (unsigned)n) {
                    // column c < N: row c of E^-1 on rows c .. N-1, h_lo[c] below; column N + c: its
                    // negation, h_hi[c] below (as eng_nnls_coop_body reads a record)
                    const bool neg = c >= (unsigned)N;
                    const int cc = (int)(neg ? c - N : c);
                    const int off = cc * (N + 2) - (cc * (cc - 1)) / 2 - cc;  // rec_row(cc) - cc
#pragma unroll
                    for (int r = 0; r < N; ++r) {
                        const double v = (r >= cc) ? grec[off + r] : 0.0;
                        col[k][r] = neg ? ((r >= cc) ? -v : 0.0) : v;
                    }
                    col[k][N] = grec[off + N + (neg ? 1 : 0)];
                }
            }
            int mode, iters;
            double rnorm;
            auto park = [&](const dvec8 (&)[CPL], const CoopCarry<CPL> &) {};  // (never suspended: no budget)
            nnls_coop<N, CPL>(live, false, 0x3fffffff, (int)(gl * CPL), col, cs, mode, rnorm, iters,
                              nnls_lds + group * COOP_WIN, nnls_lds + COOP_GROUPS_PER_WAVE * COOP_WIN, park);
            if (live) {
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    const unsigned c = gl * CPL + k;
                    if (c < (unsigned)n) gy[c] = cs.xv[k];
                }
                if (gl == 0) { gmeta[0] = (double)(mode + 8 * iters); gmeta[1] = rnorm; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (nnls_cycles) *nnls_cycles += __builtin_readcyclecounter() - t_nn;
        }
        OPTIK_SCHED_FENCE();
        if (pass) {
            if (lmode == 1) {
                if (need_nnls) {
                    lmode = ldp_from_record<N>(rec, gy, gmeta, s);
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) s[j] = 0.0;
                }
            }
            if (lmode != 1) {
                // NLopt: modes 5,6,7 -> ROUNDOFF_LIMITED; 3,4,9 -> FAILURE
                ret = (lmode == 5 || lmode == 6 || lmode == 7) ? RES_ROUNDOFF_LIMITED : RES_FAILURE;
                need_dir = false;
            } else {
                lsq_finish<N>(E, fv, lo, hi, s);
                OPTIK_SCHED_FENCE();
                // (g is also Kraft's v: the gradient at the start of the line search)
                double gs = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) x0[i] = x[i];
                f0 = f;
#pragma unroll
                for (int i = 0; i < N; ++i) gs += g[i] * s[i];
                t0 = f;
                h3 = gs;  // h3 = gs - h1 * h4 with h1 = 0 (no constraints)
                if (h3 >= 0.0) {
                    reset = true;  // not a descent direction: reset B and repeat
                } else {
                    line = 0;
                    alpha = 1.0;
                    need_dir = false;
                }
            }
        }
    }
}

template <int N, bool TIP>
OPTIK_DEV void coop_wave(const ChainDev &ch, const EvalParams &ep, const SolveParams &sp,
                         const uint32_t (&key)[8], const double (&scale)[MAX_DOF], const WorkQueue &wq,
                         double *nnls_lds /* coop_wave_lds<4>() doubles */, double *rec_lds /* 16 x coop_rec_lds<N>() */) {
    constexpr int NL = N * (N + 1) / 2;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned group = lane / COOP_GROUP, gl = lane % COOP_GROUP;
    const bool leader = gl == 0 && (int)group < wq.lanes;  // wq.lanes = restarts (groups) a wave holds at a time
    double *const grec = rec_lds + group * coop_rec_lds<N>();

    // SLSQP state of the leader's current restart (names as in solve_wave)
    double x[N], x0[N], g[N], s[N], l[NL];
    double xbest[N], xprev[N];
    double f = 0.0, f0 = 0.0, t0 = 0.0, h3 = 0.0, alpha = 1.0;
    double minf = __builtin_huge_val(), fprev = __builtin_huge_val();
    int ireset = 0, line = 0, nevals = 0;
    bool first = true;
    Pose target;
    unsigned long long item = 0, index = 0;
    unsigned tslot = 0;
    bool active = false, want = leader;
#pragma unroll
    for (int i = 0; i < N; ++i) { x[i] = 0.0; x0[i] = 0.0; g[i] = 0.0; s[i] = 0.0; xbest[i] = 0.0; xprev[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < NL; ++i) l[i] = 0.0;
    target.t = V3{0, 0, 0};
    target.q = Q4{0, 0, 0, 1};
    OPTIK_PROF_DECL;  // (-DOPTIK_PROFILE: tools/phase_profile.py; slots: 0 refill, 1 eval, 4 bookkeeping + BFGS, 5 direction, 6 NNLS of it, 3 publish, 7 trips)

    for (;;) {
        OPTIK_PROF_BEGIN();
        // ---- refill: leaders without a restart pull the next work item -----------------
        const unsigned n_want = (unsigned)__popcll(__ballot(want));
        // (a wave holds at most 16 restarts here: an idle group is 1/16 of it, and drawing a seed costs the
        // wave ~2 % of a trip -- refill as soon as one group is free; waiting for 8 as the 64-lane kernel
        // does left a quarter of the groups idle)
        if (n_want >= COOP_REFILL_BATCH || (n_want > 0 && !wave_any(active))) {
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
        OPTIK_PROF_END(0);
        if (!wave_any(active)) break;
        OPTIK_PROF_COUNT(7, 1);

        int32_t ret = 0;
        if (active) {
            // lib.rs:308: abandon when timed out or another restart of the target succeeded
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
        coop_eval<N, TIP>(ch, ep, do_eval, target, x, grec, fn, gn);
        OPTIK_PROF_END(1);
        OPTIK_SCHED_FENCE();
        bool need_dir = false, reset = false;
        OPTIK_PROF_BEGIN();
        coop_after_eval<N>(sp, do_eval, fn, gn, x, g, s, l, xbest, xprev, f, t0, h3, alpha, minf, fprev, line, nevals,
                           first, ret, need_dir, reset);
        OPTIK_PROF_END(4);
        OPTIK_PROF_BEGIN();
#ifdef OPTIK_PROFILE
        unsigned long long nnls_cycles = 0;
        coop_direction<N>(ch, sp, nnls_lds, grec, need_dir, reset, ireset, l, g, x, x0, s, f, f0, t0, h3, alpha, line, ret,
                          &nnls_cycles);
        OPTIK_PROF_COUNT(6, nnls_cycles);
#else
        coop_direction<N>(ch, sp, nnls_lds, grec, need_dir, reset, ireset, l, g, x, x0, s, f, f0, t0, h3, alpha, line, ret);
#endif
        OPTIK_PROF_END(5);
        OPTIK_PROF_BEGIN();
        if (do_eval && ret == 0) {
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
        // ---- a restart ended: classify (lib.rs:376-379), publish, free the group ----------
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
        OPTIK_PROF_END(3);
    }
    OPTIK_PROF_FLUSH(wq.prof);
}

}  // namespace optik
