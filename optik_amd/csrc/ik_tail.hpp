// ik_tail.hpp -- the last restarts of an engine run, finished without kernel boundaries.
//
// When a run's queue is empty and only a couple of thousand restarts are left, every trip of
// the phase kernels costs the latency of five dependent launches (~0.1 ms) for a handful of
// lanes.  The tail kernel takes those restarts over: one restart per lane (spread one per wave
// when there are fewer restarts than resident waves), each lane runs its restart to the end
// with the per-lane SLSQP step of the single-kernel path (ik_solve.hpp: same device functions,
// same order of operations -- the per-lane LDS NNLS instead of the cooperative one, which the
// parity tests show to be bit-identical), ~40 us per iteration.  State comes from the engine's
// slot planes at a trip boundary:
//   ST_EVAL_FIRST  restart seeded, not evaluated yet
//   ST_EVAL_TRIAL  line-search trial point waiting for its evaluation
//   ST_NNLS        direction deferred to an NNLS launch (pending or suspended): the LSQ call is
//                  simply made here from (l, g, x) -- the carry record is not needed
//   ST_DEAD        terminated, waiting to be published
// Results are published through the slot's job exactly as eng_eval_body does.
#pragma once

#include "ik_coop.hpp"

namespace optik {

template <int N, bool TIP>
OPTIK_DEV void tail_wave(const EngArgs &a, const ChainDev &ch, const EngJob *jobs, const unsigned int *list,
                         unsigned count, int lanes, const NnlsWs<N> &ws) {
    using E = EngLayout<N>;
    constexpr int NL = N * (N + 1) / 2;
    const double alfmin = 0.1;
    const SolveParams &sp = a.sp;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64u;
    const unsigned entry = wave * (unsigned)lanes + lane;
    bool active = (int)lane < lanes && entry < count;
    const size_t slot = active ? (size_t)list[entry] : 0;

    // SLSQP state of the lane's restart (names as in solve_wave)
    double x[N], x0[N], g[N], s[N], l[NL], xbest[N], xprev[N];
    double f = 0.0, f0 = 0.0, t0 = 0.0, h3 = 0.0, alpha = 1.0;
    double minf = __builtin_huge_val(), fprev = __builtin_huge_val();
    int ireset = 0, line = 0, nevals = 0;
    bool first = false, pending_dir = false;
    int32_t ret = 0;
    Pose target;
    target.t = V3{0, 0, 0};
    target.q = Q4{0, 0, 0, 1};
    unsigned long long item = 0, index = 0, tslot = 0;
    int job = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) { x[i] = 0.0; x0[i] = 0.0; g[i] = 0.0; s[i] = 0.0; xbest[i] = 0.0; xprev[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < NL; ++i) l[i] = 0.0;
    if (active) {
        const int st = ENG_I(E::STATE);
        if (st == ST_EMPTY || st == ST_REFILL) {
            active = false;
        } else {
            job = ENG_I(E::JOB);
            const EngJob &J = jobs[job];
            item = a.item[slot];
            tslot = item / J.n_restarts;
            index = J.restart_begin + (item - tslot * J.n_restarts);
            target = load_pose(J.targets + (size_t)tslot * 7);
            nevals = ENG_I(E::NEVALS);
            ireset = ENG_I(E::IRESET);
            line = ENG_I(E::LINE);
            minf = ENG_D(E::MF, 0);
            fprev = ENG_D(E::FP, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                x[i] = ENG_D(E::X, i);
                xbest[i] = ENG_D(E::XB, i);
                xprev[i] = ENG_D(E::XP, i);
            }
            if (st == ST_DEAD) {
                ret = ENG_I(E::STATUS);
            } else if (st == ST_EVAL_FIRST || st == ST_FRESH0 || st == ST_FRESH1) {
                first = true;
            } else {
#pragma unroll
                for (int i = 0; i < NL; ++i) l[i] = ENG_D(E::L, i);
#pragma unroll
                for (int i = 0; i < N; ++i) g[i] = ENG_D(E::G, i);
                if (st == ST_NNLS) {
                    pending_dir = true;  // resume at the LSQ call (labels 110/130)
                    f = ENG_D(E::FC, 0);
                } else {  // ST_EVAL_TRIAL
#pragma unroll
                    for (int i = 0; i < N; ++i) { x0[i] = ENG_D(E::X0, i); s[i] = ENG_D(E::S, i); }
                    t0 = ENG_D(E::F0, 0);
                    h3 = ENG_D(E::H3, 0);
                    alpha = ENG_D(E::AL, 0);
                }
            }
        }
    }
    const EngJob &J = jobs[job];
    // max_time (lib.rs:308): restarts still running at the deadline are abandoned
    const unsigned long long deadline = a.tail_deadline_ticks ? (unsigned long long)wall_clock64() + a.tail_deadline_ticks : 0ull;

    unsigned n_exec = 0;  // evaluations this lane executed
    while (wave_any(active)) {
        if (active && ret == 0 && deadline && (unsigned long long)wall_clock64() > deadline) ret = RES_FORCED_STOP;
        if (active && ret == 0 && J.first_success) {
            // lib.rs:308: abandon when a lower-index restart of the same target succeeded
            const unsigned long long fs = __hip_atomic_load(J.first_success + tslot, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
            if (J.find_any ? (fs != ~0ull) : (fs < index)) ret = RES_FORCED_STOP;
        }
        double gn[N];
        double fn = 0.0;
        const bool do_eval = active && ret == 0 && !pending_dir;
        OPTIK_SCHED_FENCE();
        if (do_eval) { fn = eval_fg<N, TIP>(ch, a.ep, target, x, gn); ++n_exec; }
        OPTIK_SCHED_FENCE();
        if (active && ret == 0) {
            bool need_dir = false, reset = false;
            if (pending_dir) {
                pending_dir = false;
                need_dir = true;
            } else {
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
                            const double al = h3 / ((h3 - h1) * 2.0);
                            alpha = (al > alfmin) ? al : alfmin;
                        }
                    } else {
                        const double al = alpha * 0.5;
                        alpha = (al > alfmin) ? al : alfmin;
                    }
                    if (accept) {
                        if (line > 1) ++nevals;  // NLopt re-evaluates the accepted point unless it was trial 1
                        if (!__builtin_isinf(fprev)) {
                            if (__builtin_fabs(f - fprev) < sp.ftol_abs) ret = RES_FTOL_REACHED;
                            else if (xprev_live(sp) && stop_x<N>(sp, x, xprev)) ret = RES_XTOL_REACHED;
                        }
                        fprev = f;
                        if (xprev_live(sp)) {
#pragma unroll
                            for (int i = 0; i < N; ++i) xprev[i] = x[i];
                        }
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
            // labels 110/130: (reset,) search direction, descent test
            bool have0 = false;  // a completed LSQ in this step: Kraft's (f0, x0) = (f, x)
            while (need_dir) {
                if (reset) {
                    ++ireset;
                    if (ireset > 5) {
                        // label 255 with acc = 0 -> mode 8; NLopt's relaxed test vs (f0, x0) = (f, x)
                        ret = RES_ROUNDOFF_LIMITED;
                        if (have0 && __builtin_fabs(f - f0) < sp.ftol_abs && !__builtin_isinf(f0)) ret = RES_FTOL_REACHED;
                        else if (have0 && (sp.stop_x_zero || !(0.0 >= sp.xtol_abs))) ret = RES_XTOL_REACHED;  // |x - x0| = 0 everywhere
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
                unsigned long long nnls_cycles = 0;
                const int lmode = lsq_box<N>(ws, l, g, lo, hi, s, nnls_cycles);
                OPTIK_SCHED_FENCE();
                if (lmode != 1) {
                    // NLopt: modes 5,6,7 -> ROUNDOFF_LIMITED; 3,4,9 -> FAILURE
                    ret = (lmode == 5 || lmode == 6 || lmode == 7) ? RES_ROUNDOFF_LIMITED : RES_FAILURE;
                    break;
                }
                double gs = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) x0[i] = x[i];
                f0 = f;
                have0 = true;
#pragma unroll
                for (int i = 0; i < N; ++i) gs += g[i] * s[i];
                t0 = f;
                h3 = gs;
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
        // ---- the restart ended: classify (lib.rs:376-379), publish, free the slot ----
        if (active && ret != 0) {
            const bool success = (sp.ok_stopval && ret == RES_STOPVAL_REACHED)
                                 || (sp.ok_ftol && ret == RES_FTOL_REACHED)
                                 || (sp.ok_xtol && ret == RES_XTOL_REACHED);
            if (J.out_x) {
#pragma unroll
                for (int i = 0; i < N; ++i) J.out_x[(size_t)i * J.n_items + item] = xbest[i];
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
                    for (int i = 0; i < N; ++i) { const double d = xbest[i] - x0p[i]; acc += d * d; }
                    k = __builtin_sqrt(acc);
                } else {
                    k = (double)index;
                    if (J.first_success) atomicMin(J.first_success + tslot, index);
                }
            }
            if (J.out_key) J.out_key[item] = k;
            ENG_I(E::STATE) = ST_EMPTY;
            active = false;
        }
    }
    if (a.exec_evals) {
        unsigned tot = n_exec;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) tot += (unsigned)__shfl_xor((int)tot, off, 64);
        if (lane == 0 && tot) atomicAdd(a.exec_evals + (blockIdx.x % ENG_EXEC_SHARDS), (unsigned long long)tot);
    }
}

// The same hand-over with one restart per group of four lanes and the cooperative NNLS
// (ik_coop.hpp): no per-lane LDS matrix, half the time per iteration.  `groups` restarts per wave.
template <int N, bool TIP>
OPTIK_DEV void tail_wave_coop(const EngArgs &a, const ChainDev &ch, const EngJob *jobs, const unsigned int *list,
                              unsigned count, int groups, double *nnls_lds, double *rec_lds) {
    using E = EngLayout<N>;
    constexpr int NL = N * (N + 1) / 2;
    const SolveParams &sp = a.sp;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64u;
    const unsigned group = lane / COOP_GROUP, gl = lane % COOP_GROUP;
    const unsigned entry = wave * (unsigned)groups + group;
    bool active = gl == 0 && (int)group < groups && entry < count;
    const size_t slot = active ? (size_t)list[entry] : 0;
    double *const grec = rec_lds + group * coop_rec_lds<N>();

    // SLSQP state of the lane's restart (names as in solve_wave)
    double x[N], x0[N], g[N], s[N], l[NL], xbest[N], xprev[N];
    double f = 0.0, f0 = 0.0, t0 = 0.0, h3 = 0.0, alpha = 1.0;
    double minf = __builtin_huge_val(), fprev = __builtin_huge_val();
    int ireset = 0, line = 0, nevals = 0;
    bool first = false, pending_dir = false;
    int32_t ret = 0;
    Pose target;
    target.t = V3{0, 0, 0};
    target.q = Q4{0, 0, 0, 1};
    unsigned long long item = 0, index = 0, tslot = 0;
    int job = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) { x[i] = 0.0; x0[i] = 0.0; g[i] = 0.0; s[i] = 0.0; xbest[i] = 0.0; xprev[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < NL; ++i) l[i] = 0.0;
    if (active) {
        const int st = ENG_I(E::STATE);
        if (st == ST_EMPTY || st == ST_REFILL) {
            active = false;
        } else {
            job = ENG_I(E::JOB);
            const EngJob &J = jobs[job];
            item = a.item[slot];
            tslot = item / J.n_restarts;
            index = J.restart_begin + (item - tslot * J.n_restarts);
            target = load_pose(J.targets + (size_t)tslot * 7);
            nevals = ENG_I(E::NEVALS);
            ireset = ENG_I(E::IRESET);
            line = ENG_I(E::LINE);
            minf = ENG_D(E::MF, 0);
            fprev = ENG_D(E::FP, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                x[i] = ENG_D(E::X, i);
                xbest[i] = ENG_D(E::XB, i);
                xprev[i] = ENG_D(E::XP, i);
            }
            if (st == ST_DEAD) {
                ret = ENG_I(E::STATUS);
            } else if (st == ST_EVAL_FIRST || st == ST_FRESH0 || st == ST_FRESH1) {
                first = true;
            } else {
#pragma unroll
                for (int i = 0; i < NL; ++i) l[i] = ENG_D(E::L, i);
#pragma unroll
                for (int i = 0; i < N; ++i) g[i] = ENG_D(E::G, i);
                if (st == ST_NNLS) {
                    pending_dir = true;  // resume at the LSQ call (labels 110/130)
                    f = ENG_D(E::FC, 0);
                } else {  // ST_EVAL_TRIAL
#pragma unroll
                    for (int i = 0; i < N; ++i) { x0[i] = ENG_D(E::X0, i); s[i] = ENG_D(E::S, i); }
                    t0 = ENG_D(E::F0, 0);
                    h3 = ENG_D(E::H3, 0);
                    alpha = ENG_D(E::AL, 0);
                }
            }
        }
    }
    const EngJob &J = jobs[job];
    // max_time (lib.rs:308): restarts still running at the deadline are abandoned
    const unsigned long long deadline = a.tail_deadline_ticks ? (unsigned long long)wall_clock64() + a.tail_deadline_ticks : 0ull;

    unsigned n_exec = 0;  // evaluations this lane executed
    while (wave_any(active)) {
        if (active && ret == 0 && deadline && (unsigned long long)wall_clock64() > deadline) ret = RES_FORCED_STOP;
        if (active && ret == 0 && J.first_success) {
            // lib.rs:308: abandon when another restart of the same target succeeded
            const unsigned long long fs = __hip_atomic_load(J.first_success + tslot, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
            if (J.find_any ? (fs != ~0ull) : (fs < index)) ret = RES_FORCED_STOP;
        }
        double gn[N];
        double fn = 0.0;
        const bool stepping = active && ret == 0;
        const bool do_eval = stepping && !pending_dir;
        OPTIK_SCHED_FENCE();
        coop_eval<N, TIP>(ch, a.ep, do_eval, target, x, grec, fn, gn);
        if (do_eval) ++n_exec;
        OPTIK_SCHED_FENCE();
        bool need_dir = stepping && pending_dir, reset = false;  // a deferred direction resumes at its LSQ call
        pending_dir = false;
        coop_after_eval<N>(sp, do_eval, fn, gn, x, g, s, l, xbest, xprev, f, t0, h3, alpha, minf, fprev, line, nevals,
                           first, ret, need_dir, reset);
        coop_direction<N>(ch, sp, nnls_lds, grec, need_dir, reset, ireset, l, g, x, x0, s, f, f0, t0, h3, alpha, line, ret);
        if (stepping && ret == 0) {
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
        // ---- the restart ended: classify (lib.rs:376-379), publish, free the slot ----
        if (active && ret != 0) {
            const bool success = (sp.ok_stopval && ret == RES_STOPVAL_REACHED)
                                 || (sp.ok_ftol && ret == RES_FTOL_REACHED)
                                 || (sp.ok_xtol && ret == RES_XTOL_REACHED);
            if (J.out_x) {
#pragma unroll
                for (int i = 0; i < N; ++i) J.out_x[(size_t)i * J.n_items + item] = xbest[i];
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
                    for (int i = 0; i < N; ++i) { const double d = xbest[i] - x0p[i]; acc += d * d; }
                    k = __builtin_sqrt(acc);
                } else {
                    k = (double)index;
                    if (J.first_success) atomicMin(J.first_success + tslot, index);
                }
            }
            if (J.out_key) J.out_key[item] = k;
            ENG_I(E::STATE) = ST_EMPTY;
            active = false;
        }
    }
    if (a.exec_evals) {
        unsigned tot = n_exec;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) tot += (unsigned)__shfl_xor((int)tot, off, 64);
        if (lane == 0 && tot) atomicAdd(a.exec_evals + (blockIdx.x % ENG_EXEC_SHARDS), (unsigned long long)tot);
    }
}

}  // namespace optik
