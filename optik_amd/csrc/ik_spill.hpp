// ik_spill.hpp -- the last restarts of a lane-per-restart launch, finished by the quad solver inside the same kernel.
//
// A launch of ik_lane_kernel (ik_lane64.hpp: one restart per lane, 64 per wave) ends with its longest restart: once
// the work queue is dry a wave keeps running for the few lanes that still hold one, and a trip with few live lanes
// costs most of a full one (~40 us) -- 3.8 ms of fill + drain per launch in round 4, and a lone 65 536-restart launch
// is nothing but drain.  The reference keeps every rayon worker busy until the index range is exhausted
// (/root/reference/crates/optik/src/lib.rs:297-300); here the waves change FORM for the launch's end:
//
//   * when the queue is dry and at most `spill_at` of a wave's lanes still hold a restart, those lanes write the
//     restart's SLSQP state -- at the trip boundary, i.e. in front of an evaluation -- to a slot of the spill pool
//     (slot = the lane's own global number: no allocation) and append the slot to the launch's spill list;
//   * the wave then runs the quad solver (ik_quad.hpp: four lanes per restart, sixteen restarts per wave, ~13 us per
//     iteration instead of ~40) fed from that list: a quad that is free draws a ticket, waits for that list entry,
//     reads the state from the slot planes (each lane its own joints / rows) and carries on where the lane left off --
//     same arithmetic, same bits.  The list is shared by the launch's waves: a wave whose own restarts were short
//     finishes those of its neighbours, and the launch ends when every wave has left the lane form and the list is
//     consumed.
//
// (A first version ran the tail as a second kernel behind the launch: the spilled restarts then WAIT for the launch's
// last wave, and every threshold was slower than no spill at all -- profiles/r5c_spill_sweep.txt.)
//
// A restart is spilled in one of two states (the lane form's state at the top of its loop):
//   SP_FIRST   seeded, not evaluated yet
//   SP_TRIAL   a line-search trial point waiting for its evaluation (x = x0 + alpha s already formed)
// (a lane whose last direction was not a descent direction -- reset B and search again, 5e-5 of the trips -- keeps
// its wave in the lane form for one more trip: the spill waits until no lane is in that state).
//
// Protocol (all counters in the launch's queue block, zero before the launch and put back by its selection kernel):
//   count   entries RESERVED in the list (a spilling wave adds its lane count, then writes data and entries)
//   list    the slot of each entry; SPILL_NONE until written (store-release after the slot's planes), put back to
//           SPILL_NONE by the quad that consumes it
//   cursor  tickets handed out
//   done    waves that have left the lane form (no reservation can follow once it equals the grid size)
#pragma once

#include "ik_slsqp.hpp"
#include "ik_solve.hpp"

namespace optik {

template <int N>
struct SpillLayout {
    static constexpr int NL = N * (N + 1) / 2;
    // double planes
    static constexpr int X = 0, X0 = X + N, G = X0 + N, S = G + N, XB = S + N, XP = XB + N, L = XP + N, F0 = L + NL,
                         H3 = F0 + 1, AL = H3 + 1, FP = AL + 1, MF = FP + 1, ND = MF + 1;
    // int32 planes
    static constexpr int STATE = 0, LINE = 1, IRESET = 2, NEVALS = 3, NI = 4;
};
static_assert(SPILL_ND_MAX == SpillLayout<7>::ND && SPILL_NI == SpillLayout<7>::NI, "the pool is sized for n <= 7 (ik_solve.hpp)");
enum : int { SP_EMPTY = 0, SP_FIRST = 1, SP_TRIAL = 2 };
constexpr unsigned SPILL_NONE = 0xffffffffu;  // a list entry that has not been written (yet)

#define SPILL_D(P, plane, k) (P).d[(size_t)((plane) + (k)) * (P).C + slot]
#define SPILL_I(P, plane) (P).i32[(size_t)(plane) * (P).C + slot]

// The lane's restart into its slot (called by every lane of the wave that holds one, at the top of a trip).
template <int N>
OPTIK_DEV void spill_export(const SpillPool &P, size_t slot, bool first, const double (&x)[N], const double (&x0)[N],
                            const double (&g)[N], const double (&s)[N], const double (&xbest)[N], const double (&xprev)[N],
                            const double (&l)[N * (N + 1) / 2], double f0, double h3, double alpha, double fprev, double minf,
                            int line, int ireset, int nevals, unsigned long long item) {
    using E = SpillLayout<N>;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        SPILL_D(P, E::X, i) = x[i];
        SPILL_D(P, E::X0, i) = x0[i];
        SPILL_D(P, E::G, i) = g[i];
        SPILL_D(P, E::S, i) = s[i];
        SPILL_D(P, E::XB, i) = xbest[i];
        SPILL_D(P, E::XP, i) = xprev[i];
    }
#pragma unroll
    for (int i = 0; i < E::NL; ++i) SPILL_D(P, E::L, i) = l[i];
    SPILL_D(P, E::F0, 0) = f0;
    SPILL_D(P, E::H3, 0) = h3;
    SPILL_D(P, E::AL, 0) = alpha;
    SPILL_D(P, E::FP, 0) = fprev;
    SPILL_D(P, E::MF, 0) = minf;
    SPILL_I(P, E::STATE) = first ? SP_FIRST : SP_TRIAL;
    SPILL_I(P, E::LINE) = line;
    SPILL_I(P, E::IRESET) = ireset;
    SPILL_I(P, E::NEVALS) = nevals;
    P.item[slot] = item;
}

}  // namespace optik

#ifdef OPTIK_SPILL_TAIL  // (the quad solver's side: ik_quad_kernel.hip)
#include "ik_quad.hpp"

namespace optik {

// What quad_wave's Tail hook (ik_quad.hpp) is given: the pool, and the launch's work queue as the one "job" every
// spilled restart belongs to (targets, seeds, outputs, first-success words).
struct SpillTail {
    static constexpr bool on = true;
    SpillPool pool;
    const WorkQueue *wq;  // the launch's queue record (the kernel's LDS copy)
    unsigned long long *cursor;
    unsigned long long deadline;  // wall_clock64() ticks (absolute), 0 = none
    unsigned n_waves;             // waves of the launch: `done` reaches this when the last one has left the lane form

    OPTIK_DEV const WorkQueue &job(int) const { return *wq; }

    // Is the list entry of `ticket` there?  1 (slot = its slot; the entry is put back to SPILL_NONE), 0 not yet,
    // -1 it never will be: every wave has left the lane form and fewer entries were reserved.
    OPTIK_DEV int poll(unsigned long long ticket, unsigned &slot) const {
        // (done first: if it is complete, the count read after it is final)
        const unsigned long long dn = __hip_atomic_load(pool.done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long cnt = __hip_atomic_load(pool.count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket < cnt) {
            const unsigned v = __hip_atomic_load(pool.list + ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (v == SPILL_NONE) return 0;
            __hip_atomic_store(pool.list + ticket, SPILL_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            slot = v;
            return 1;
        }
        return dn >= (unsigned long long)n_waves ? -1 : 0;
    }
    OPTIK_DEV void idle() const { __builtin_amdgcn_s_sleep(16); }

    // The restart of `slot_u` into the quad's registers (lane q: joints / rows q, q + 4).  Returns whether
    // the slot held one.
    template <int N>
    OPTIK_DEV bool import(unsigned slot_u, int q, double (&x)[QuadDims<N>::NS], double (&x0)[QuadDims<N>::NS],
                          double (&g)[QuadDims<N>::NS], double (&sv)[QuadDims<N>::NS],
                          double (&Lr)[QuadDims<N>::NS][QuadDims<N>::NM], double (&dg)[QuadDims<N>::NS], double &pa,
                          double &pb, int &ia, int &ib, bool &first, bool &pending, int32_t &ret, double *xb,
                          double *xp) const {
        constexpr int NS = QuadDims<N>::NS, NM = QuadDims<N>::NM;
        using E = SpillLayout<N>;
        const SpillPool &P = pool;
        const size_t slot = slot_u;
        const int st = SPILL_I(P, E::STATE);
        if (st == SP_EMPTY) return false;
        const unsigned long long it = P.item[slot];
        const unsigned long long ts = it / wq->n_restarts;
        const unsigned long long rr = it - ts * wq->n_restarts;
        first = st == SP_FIRST;
        pending = false;
        ret = 0;
        const bool trial = !first;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int r = q + 4 * s;
            const bool val = r < N;
            const int jc = val ? r : N - 1;
            const double xv = SPILL_D(P, E::X, jc);
            x[s] = val ? xv : 0.0;
            xb[s * 64] = val ? SPILL_D(P, E::XB, jc) : 0.0;
            xp[s * 64] = val ? SPILL_D(P, E::XP, jc) : 0.0;
            x0[s] = x[s];
            g[s] = 0.0;
            sv[s] = 0.0;
            dg[s] = 1.0;
#pragma unroll
            for (int i = 0; i < NM; ++i) Lr[s][i] = 0.0;
            if (trial && val) {
                dg[s] = SPILL_D(P, E::L, lidx<N>(jc, jc));
#pragma unroll
                for (int i = 0; i < N - 1; ++i)
                    if (slot_has<N>(s, i) && i < jc) Lr[s][i] = SPILL_D(P, E::L, lidx<N>(i, jc));  // l(i, row): column i of L
                g[s] = SPILL_D(P, E::G, jc);
                x0[s] = SPILL_D(P, E::X0, jc);
                sv[s] = SPILL_D(P, E::S, jc);
            }
        }
        const double f0v = trial ? SPILL_D(P, E::F0, 0) : 0.0, h3v = trial ? SPILL_D(P, E::H3, 0) : 0.0;
        const double alv = trial ? SPILL_D(P, E::AL, 0) : 1.0;
        // the restart's scalars, one per lane (ik_quad.hpp): pa f0 | t0 | h3 | alpha, pb minf | fprev | f | --,
        // ia ireset | line | nevals | target slot, ib restart number low | high | slot | job
        pa = (q < 2) ? f0v : ((q == 2) ? h3v : alv);
        pb = (q == 0) ? SPILL_D(P, E::MF, 0) : ((q == 1) ? SPILL_D(P, E::FP, 0) : 0.0);
        ia = (q == 0) ? SPILL_I(P, E::IRESET) : ((q == 1) ? SPILL_I(P, E::LINE) : ((q == 2) ? SPILL_I(P, E::NEVALS) : (int)(unsigned)ts));
        ib = (q == 0) ? (int)(unsigned)(rr & 0xffffffffull)
                      : ((q == 1) ? (int)(unsigned)(rr >> 32) : ((q == 2) ? (int)slot_u : 0));
        return true;
    }

    // the slot gives its restart up (published by the caller): nothing to do -- a launch's spill list is consumed once
    template <int N>
    OPTIK_DEV void release(unsigned) const {}
};

}  // namespace optik
#endif  // OPTIK_SPILL_TAIL
