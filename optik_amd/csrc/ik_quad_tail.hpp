// ik_quad_tail.hpp -- the last restarts of an engine run, finished by the quad solver.
//
// When an engine run's queue is empty its slot pool drains: every trip of the five phase kernels
// covers fewer slots at the same launch latencies.  The last restarts are therefore handed to a kernel
// without launch boundaries: the quad solver (ik_quad.hpp), fed from the slot pool instead of the work
// queue.  A quad that is free pulls the next entry of the list of live slots, reads the
// restart's state from the slot planes (each lane its own joints / rows; the restart's scalars one
// per lane, as the solver keeps them) and carries on from the slot's state at the trip boundary:
//   ST_EVAL_FIRST              seeded, not evaluated yet
//   ST_EVAL_TRIAL              line-search trial point waiting for its evaluation
//   ST_NNLS                    direction deferred to an NNLS launch (pending or suspended): the LSQ
//                              call is made again from (l, g, x) -- the carry record is not needed
//   ST_DEAD                    terminated, waiting to be published
// Results are published through the slot's job exactly as eng_eval_body does.
// Arithmetic and decisions are quad_wave's: same bits as every other path.
#pragma once

#include "ik_engine.hpp"
#include "ik_launch.hpp"
#include "ik_quad.hpp"

namespace optik {

struct EngTail : EngTailData {
    static constexpr bool on = true;

    // The restart of `slot_u` into the quad's registers (lane q: joints / rows q, q + 4).  Returns whether
    // the slot held one.
    template <int N>
    OPTIK_DEV bool import(unsigned slot_u, int q, double (&x)[QuadDims<N>::NS], double (&x0)[QuadDims<N>::NS],
                          double (&g)[QuadDims<N>::NS], double (&sv)[QuadDims<N>::NS],
                          double (&Lr)[QuadDims<N>::NS][QuadDims<N>::NM], double (&dg)[QuadDims<N>::NS], double &pa,
                          double &pb, int &ia, int &ib, bool &first, bool &pending, int32_t &ret, double *xb,
                          double *xp) const {
        constexpr int NS = QuadDims<N>::NS, NM = QuadDims<N>::NM;
        using E = EngLayout<N>;
        const EngTailData &a = *this;  // (ENG_D / ENG_I address a.d / a.i32 / a.C by `slot`)
        const size_t slot = slot_u;
        const int st = ENG_I(E::STATE);
        if (st == ST_EMPTY || st == ST_REFILL) return false;
        const int job = ENG_I(E::JOB);
        const EngJob &J = jobs[job];
        const unsigned long long it = item[slot];
        const unsigned long long ts = it / J.n_restarts;
        const unsigned long long rr = it - ts * J.n_restarts;
        first = false;
        pending = false;
        ret = 0;
        double f0v = 0.0, h3v = 0.0, alv = 1.0, fc = 0.0;
        const bool fresh = st == ST_EVAL_FIRST;
        const bool has_factor = !fresh && st != ST_DEAD;
        const bool trial = has_factor && st != ST_NNLS;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int r = q + 4 * s;
            const bool val = r < N;
            const int jc = val ? r : N - 1;
            const double xv = ENG_D(E::X, jc);
            x[s] = val ? xv : 0.0;
            xb[s * 64] = val ? ENG_D(E::XB, jc) : 0.0;
            xp[s * 64] = val ? ENG_D(E::XP, jc) : 0.0;
            x0[s] = x[s];
            g[s] = 0.0;
            sv[s] = 0.0;
            dg[s] = 1.0;
#pragma unroll
            for (int i = 0; i < NM; ++i) Lr[s][i] = 0.0;
            if (has_factor && val) {
                dg[s] = ENG_D(E::L, lidx<N>(jc, jc));
#pragma unroll
                for (int i = 0; i < N - 1; ++i)
                    if (slot_has<N>(s, i) && i < jc) Lr[s][i] = ENG_D(E::L, lidx<N>(i, jc));  // l(i, row): column i of L
                g[s] = ENG_D(E::G, jc);
                if (trial) { x0[s] = ENG_D(E::X0, jc); sv[s] = ENG_D(E::S, jc); }
            }
        }
        if (st == ST_DEAD) ret = ENG_I(E::STATUS);
        else if (fresh) first = true;
        else if (st == ST_NNLS) { pending = true; fc = ENG_D(E::FC, 0); }
        else { f0v = ENG_D(E::F0, 0); h3v = ENG_D(E::H3, 0); alv = ENG_D(E::AL, 0); }
        // the restart's scalars, one per lane (ik_quad.hpp): pa f0 | t0 | h3 | alpha, pb minf | fprev | f | --,
        // ia ireset | line | nevals | target slot, ib restart number low | high | slot | job
        pa = (q < 2) ? f0v : ((q == 2) ? h3v : alv);
        pb = (q == 0) ? ENG_D(E::MF, 0) : ((q == 1) ? ENG_D(E::FP, 0) : ((q == 2) ? fc : 0.0));
        ia = (q == 0) ? ENG_I(E::IRESET) : ((q == 1) ? ENG_I(E::LINE) : ((q == 2) ? ENG_I(E::NEVALS) : (int)(unsigned)ts));
        ib = (q == 0) ? (int)(unsigned)(rr & 0xffffffffull)
                      : ((q == 1) ? (int)(unsigned)(rr >> 32) : ((q == 2) ? (int)slot_u : job));
        return true;
    }

    // the slot gives its restart up (published by the caller)
    template <int N>
    OPTIK_DEV void release(unsigned slot_u) const {
        using E = EngLayout<N>;
        const EngTailData &a = *this;
        const size_t slot = slot_u;
        ENG_I(E::STATE) = ST_EMPTY;
    }
};

}  // namespace optik
