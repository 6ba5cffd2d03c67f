// ik_quad.hpp -- the restart solver with one restart per QUAD and its state SPREAD over the quad.
//
// (Round 2's leader-lane solver gave a restart four lanes too, but kept the whole SLSQP state in the
// quad's leader: 512 registers + 428 B of scratch, one wave per SIMD, three lanes idle outside the
// NNLS and the Jacobian columns.)  Here lane q of a quad owns joints / rows / columns q and q + 4:
//
//   by joint (2 doubles per lane each)   x, x0, g, s, x_best, x_prev, lb, ub, f (LSQ right-hand side)
//   row j of the packed LDL' factor      l(i, j), i < j, and the diagonal l(j, j)       (11 doubles, n = 7)
//   column j of E = D^1/2 L'             E[i][j], i < j, and E[j][j]                     (same shape)
//   rows q, q + 4 of E^-1                = NNLS columns q, q + 4 (lower bounds), n + q, n + q + 4 (upper)
//   Jacobian columns q, q + 4            -> gradient components q, q + 4
//
// and every scalar of Kraft's / NLopt's state machine is the same in the four lanes: inside a region
// of a trip it is replicated, ACROSS the regions each scalar is kept by one lane (pa / pb / ia / ib in
// quad_wave) and fetched where it is used.  Values move between the lanes of a quad with DPP
// quad_perm moves (ik_lane.hpp), never through LDS; the LDS of the solver is the 1 KB block per quad
// that holds the matrix of the bounded dual problem while ik_nnls_quad.hpp solves it -- and, while
// the quad evaluates, what the evaluation does not touch -- plus x_best / x_prev of every lane.
// The restarts come from the launch's work queue.
//
// Bit-exactness (the contract of DESIGN.md section 2): every sum the reference / oracle forms
// sequentially is formed here in the SAME ORDER from the same products -- either inside one lane
// (column j of E holds all the terms of f_j's recurrence, row j of L all the terms of (L' v)_j) or,
// where the terms live in different lanes, by fetching them one at a time in index order into an
// accumulator every lane carries (quad_dot).  What is spread over the lanes is work whose pieces
// are independent: sin / cos and local frames of different joints, Jacobian columns, rows of E^-1,
// the row updates of a rank-one LDL' modification, the three quotients of one of its pivots.
//
// Restates, per quad: /root/reference/crates/optik/src/lib.rs:301-391 (the restart closure) with
// NLopt's SLSQP (un-vendored; oracle/optik_oracle.c is the CPU statement of the same arithmetic).
#pragma once

#include "ik_lane.hpp"
#include "ik_solve.hpp"
#include "ik_nnls_quad.hpp"

namespace optik {

template <int N>
struct QuadDims {
    static constexpr int NS = (N > 4) ? 2 : 1;       // joints (rows, columns) a lane owns: q and q + 4
    static constexpr int NM = (N > 1) ? N - 1 : 1;   // entries below the diagonal a row can have
};

// can a row of slot s (rows 4 s .. 4 s + 3) have an entry in column i (i.e. some row > i, row < N)
template <int N>
OPTIK_DEV constexpr bool slot_has(int s, int i) { return i < N - 1 && (s == 1 || i < 3); }

OPTIK_DEV unsigned long long quad_get_u64(unsigned long long v, int k) {
    const int lo = quad_get((int)(unsigned)(v & 0xffffffffull), k);
    const int hi = quad_get((int)(unsigned)(v >> 32), k);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
}

// sum_k a_k b_k over the joints in index order, starting from +0.0 (the oracle's `acc = 0; acc += ...`):
// products where the operands live, one fetch per term.  Same value in every lane of the quad.
template <int N, int NS>
OPTIK_DEV double quad_dot(const double (&a)[NS], const double (&b)[NS]) {
    double p[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) p[s] = a[s] * b[s];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) acc += quad_get(p[k >> 2], k);
    return acc;
}

// Random configuration of restart `index` (ik_solve.hpp:restart_seed; lib.rs:86-91, 366-370) with the
// ChaCha8 block spread over the quad: lane q holds column q of the 4 x 4 state, a column round is the
// lane's own quarter round, a diagonal round the same after rotating rows 1, 2, 3 by one, two, three
// lanes (DPP) -- a quarter of the integer work of the block per lane.  Joint k's 64 bits are words
// 2k, 2k + 1: row k / 2, lanes 2 (k & 1) and 2 (k & 1) + 1.  Writes the lane's joints' components.
template <int N>
OPTIK_DEV void restart_seed_quad(const uint32_t (&key)[8], const double *lb, const double *scale, uint64_t index,
                                 double (&xq)[QuadDims<N>::NS]) {
    static_assert(N <= 8, "one ChaCha block per restart");
    constexpr int NS = QuadDims<N>::NS;
    const int q = quad_lane_now();
    const uint32_t cst = (q == 0) ? 0x61707865u : ((q == 1) ? 0x3320646eu : ((q == 2) ? 0x79622d32u : 0x6b206574u));
    const uint32_t s1 = key[q], s2 = key[4 + q];
    // words 12-13: block counter (0), 14-15: stream id
    const uint32_t s3 = (q == 2) ? (uint32_t)index : ((q == 3) ? (uint32_t)(index >> 32) : 0u);
    uint32_t a = cst, b = s1, c = s2, d = s3;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        OPTIK_QR(a, b, c, d)
        b = quad_rot(b, 1); c = quad_rot(c, 2); d = quad_rot(d, 3);
        OPTIK_QR(a, b, c, d)
        b = quad_rot(b, 3); c = quad_rot(c, 2); d = quad_rot(d, 1);
    }
    const uint32_t row[4] = {a + cst, b + s1, c + s2, d + s3};
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const uint32_t lo = (uint32_t)quad_get((int)row[k >> 1], 2 * (k & 1));
        const uint32_t hi = (uint32_t)quad_get((int)row[k >> 1], 2 * (k & 1) + 1);
        const uint64_t bits = (uint64_t)lo | ((uint64_t)hi << 32);
        const double v = uniform_inclusive(lb[k], scale[k], bits);
        xq[k >> 2] = (q == (k & 3)) ? v : xq[k >> 2];
    }
}

OPTIK_DEV Pose pose_sel(bool c, const Pose a, const Pose b) {
    Pose o;
    o.t = V3{c ? a.t.x : b.t.x, c ? a.t.y : b.t.y, c ? a.t.z : b.t.z};
    o.q = Q4{c ? a.q.i : b.q.i, c ? a.q.j : b.q.j, c ? a.q.k : b.q.k, c ? a.q.w : b.q.w};
    return o;
}

// ---- objective + gradient (ik_eval.hpp:eval_fg_stream, spread over the quad) ---------------------
// x by joint in, gradient by joint out; f in every lane.  sin / cos and origin * local of a joint
// are computed by its owner (kinematics.rs:142-158 forms `joint.origin * local_transform(q)` before
// multiplying it onto the chain: independent per joint); the chain product itself is sequential
// and every lane walks it, keeping the frames of its own joints; the error terms are formed by all
// four lanes alike; each lane then does the Jacobian columns of its joints.
// (makes the compiler forget what it knows about a pointer: a later load through it is issued again
// instead of keeping the first load's registers alive in between)
template <class T>
OPTIK_DEV const T *reload_barrier(const T *p) {
#ifndef OPTIK_NO_LAUNDER
    asm volatile("" : OPTIK_REG_INOUT(p));
#endif
    return p;
}

// The same for a pointer INTO LDS: only the offset is laundered, the address space is kept (ik_platform.hpp).
template <class T>
OPTIK_DEV const T *reload_barrier_lds(const T *p) {
#ifndef OPTIK_NO_LAUNDER
    return launder_lds(p);
#else
    return p;
#endif
}

// (the same for an integer: what is computed from the result is computed where it is used, not hoisted out
// of the solver loop as an invariant -- and then spilled for the whole loop)
OPTIK_DEV int opaque_int(int v) {
#ifndef OPTIK_NO_LAUNDER
    asm volatile("" : OPTIK_REG_INOUT(v));
#endif
    return v;
}

template <int N, bool TIP>
OPTIK_DEV double eval_quad(const ChainDev &ch, const EvalParams &ep, const double *target7,
                           const double (&x)[QuadDims<N>::NS], double (&gout)[QuadDims<N>::NS],
                           double *park /* the lane's parking doubles in the quad's block: park[4 i], i < 12 */) {
    constexpr int NS = QuadDims<N>::NS;
    const int q = quad_lane_now();
    Q4 jq[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int j = (q + 4 * s < N) ? q + 4 * s : N - 1;
        double sn, cs;
        sincos_dev(x[s] / 2.0, sn, cs);  // UnitQuaternion::from_axis_angle
        const Q4 local{ch.axis[j][0] * sn, ch.axis[j][1] * sn, ch.axis[j][2] * sn, cs};
        jq[s] = qmul(Q4{ch.origin[j][3], ch.origin[j][4], ch.origin[j][5], ch.origin[j][6]}, local);
        OPTIK_SCHED_FENCE();
    }
    Pose tf[NS];  // T_w_j after joint j's own rotation, for the lane's joints (kinematics.rs:153-156)
#pragma unroll
    for (int s = 0; s < NS; ++s) { tf[s].t = V3{0, 0, 0}; tf[s].q = Q4{0, 0, 0, 1}; }
    Pose state;
    state.t = V3{0, 0, 0};
    state.q = Q4{0, 0, 0, 1};
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int sk = k >> 2;
        Pose jt;  // joint.origin * local_transform(q): the translation part is exact
        jt.t = V3{ch.origin[k][0], ch.origin[k][1], ch.origin[k][2]};
        jt.q = Q4{quad_get(jq[sk].i, k), quad_get(jq[sk].j, k), quad_get(jq[sk].k, k), quad_get(jq[sk].w, k)};
        state = (k == 0) ? jt : pose_mul(state, jt);  // identity * jt is exact
        tf[sk] = pose_sel(q == (k & 3), state, tf[sk]);
        OPTIK_SCHED_FENCE();
    }
    if (TIP) state = pose_mul(state, load_pose(ch.origin[N]));
    const Pose ee = ep.has_ee_offset ? pose_mul(state, load_pose(ep.ee_offset)) : state;  // kinematics.rs:163

    // the lane's joints: body-frame Jacobian column (kinematics.rs:173-184) -- formed here, while the
    // joint frames are at hand: six doubles per joint stay, the frames (fourteen) go
    double lin[NS][3], ang[NS][3];
    {
        const Q4 eeqc = qconj(ee.q);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = (q + 4 * s < N) ? q + 4 * s : N - 1;
            const V3 ax{ch.axis[k][0], ch.axis[k][1], ch.axis[k][2]};
            const V3 angular = qrot(tf[s].q, ax);
            const V3 d{ee.t.x - tf[s].t.x, ee.t.y - tf[s].t.y, ee.t.z - tf[s].t.z};
            const V3 linear = cross(angular, d);
            const V3 al = qrot(eeqc, angular);
            const V3 ll = qrot(eeqc, linear);
            lin[s][0] = ll.x; lin[s][1] = ll.y; lin[s][2] = ll.z;
            ang[s][0] = al.x; ang[s][1] = al.y; ang[s][2] = al.z;
            OPTIK_SCHED_FENCE();
        }
    }

    // X = T_target^-1 T_ee  (objective.rs:69-70)
    const Pose X = pose_inv_mul(load_pose(target7), ee);
    // (the columns' geometry waits in LDS while the error terms -- the register peak of the kernel -- are formed)
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c) { park[4 * (6 * s + c)] = lin[s][c]; park[4 * (6 * s + 3 + c)] = ang[s][c]; }
    OPTIK_SCHED_FENCE();
    const V3 w = so3_log(X.q);
    const RotTerms rt = rot_terms(w);
    const M3 Jr = so3_right_jacobian(rt);          // math.rs:195
    const M3 Qm = se3_q_matrix(rt, X.t, Jr);       // math.rs:196 (E = Jr, math.rs:167)
    const V3 elin = se3_log_linear(rt, X.t);       // math.rs:120-122
    OPTIK_SCHED_FENCE();

    // weighted error for the value (objective.rs:52) and for the gradient (:104)
    const double *tq7 = reload_barrier(target7);
    const Q4 tq{tq7[3], tq7[4], tq7[5], tq7[6]};
    V3 fl = elin, fa = w;
    if (!ep.skip_lin) fl = weight_block(tq, elin, ep.w_lin);
    if (!ep.skip_ang) fa = weight_block(tq, w, ep.w_ang);
    V3 gl = fl, ga = fa;
    if (!ep.grad_same_as_value) {
        gl = elin; ga = w;
        if (!ep.skip_lin2) gl = weight_block(tq, elin, ep.w_lin2);
        if (!ep.skip_ang2) ga = weight_block(tq, w, ep.w_ang2);
    }
    const double e2[6] = {2.0 * gl.x, 2.0 * gl.y, 2.0 * gl.z, 2.0 * ga.x, 2.0 * ga.y, 2.0 * ga.z};
    // f = ||e||^2 (objective.rs:56)
    const double ef[6] = {fl.x, fl.y, fl.z, fa.x, fa.y, fa.z};
    double f = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) f += ef[i] * ef[i];
    OPTIK_SCHED_FENCE();

    // Jtask = Jlog6 * J (objective.rs:81) and g = (2 e') Jtask (objective.rs:106-109) for the lane's columns
    const double *pk = reload_barrier_lds((const double *)park);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { lin[s][c] = pk[4 * (6 * s + c)]; ang[s][c] = pk[4 * (6 * s + 3 + c)]; }
        double jt[6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m) acc += Jr.m[r][m] * lin[s][m];
#pragma unroll
            for (int m = 0; m < 3; ++m) acc += Qm.m[r][m] * ang[s][m];
            jt[r] = acc;
            double acc2 = 0.0;  // lower-left block of Jlog6 is zero
#pragma unroll
            for (int m = 0; m < 3; ++m) acc2 += Jr.m[r][m] * ang[s][m];
            jt[r + 3] = acc2;
        }
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) acc += e2[r] * jt[r];
        gout[s] = acc;
        OPTIK_SCHED_FENCE();
    }
    return f;
}

// nlopt_stop_x (ik_solve.hpp:stop_x) on by-joint vectors: the per-joint tests, then "all of them" over the quad
template <int N>
OPTIK_DEV bool stop_x_quad(const SolveParams &sp, const double (&x)[QuadDims<N>::NS],
                           const double (&oldx)[QuadDims<N>::NS]) {
    constexpr int NS = QuadDims<N>::NS;
    const int q = quad_lane();
    bool zero = true, allx = true;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bool val = q + 4 * s < N;
        zero = zero && (!val || x[s] == oldx[s]);
        allx = allx && (!val || !(__builtin_fabs(x[s] - oldx[s]) >= sp.xtol_abs));
    }
    const bool zero_q = quad_all(zero), allx_q = quad_all(allx);
    return (sp.stop_x_zero != 0 && zero_q) || allx_q;
}

// ---- Fletcher-Powell composite-t update LDL' += sigma z z' (ik_slsqp.hpp:ldl_update) -----------
// Row j of the factor and z_j live with joint j's owner.  Per pivot i: z_i and l(i,i) are fetched
// from the owner, the pivot's scalars are formed by every lane, the three quotients tp/t, delta/tp,
// t/tp by lanes 0, 1, 2 at once (one division's time), and every lane updates its own rows j > i.
// `live`: the quad really performs the update (the others run the same instructions on whatever
// they hold and keep their state).
template <int N>
OPTIK_DEV void ldl_quad(bool live, double (&Lr)[QuadDims<N>::NS][QuadDims<N>::NM], double (&dg)[QuadDims<N>::NS],
                        double (&z)[QuadDims<N>::NS], double sigma) {
    constexpr int NS = QuadDims<N>::NS;
    const int q = quad_lane();
    live = live && sigma != 0.0;
    const bool neg = sigma < 0.0;
    double w[NS], dl[NS];
    double t = 1.0 / sigma;
    const bool any_neg = wave_any(live && neg), any_pos = wave_any(live && !neg);
    if (any_neg) {
        // forward substitution w = L^-1 z first (no quotient in it) ...
#pragma unroll
        for (int s = 0; s < NS; ++s) w[s] = z[s];
#pragma unroll
        for (int i = 0; i < N - 1; ++i) {
            const double v = quad_get(w[i >> 2], i);
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (slot_has<N>(s, i)) w[s] = (q + 4 * s > i) ? w[s] - v * Lr[s][i] : w[s];
            OPTIK_SCHED_FENCE();
        }
        // ... then the quotients w_j^2 / l(j,j) and w_j / l(j,j), each by its owner, all at once: they feed
        // the sum t + sum_j w_j^2 / l(j,j) (in j order), t_j = t_(j+1) - w_j^2 / l(j,j), and -- the main
        // loop below repeats this very substitution on z, so its v is w_j -- that loop's delta = v / l(j,j)
        double c[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { c[s] = w[s] * w[s] / dg[s]; dl[s] = w[s] / dg[s]; }
        double tn = t;
#pragma unroll
        for (int i = 0; i < N; ++i) tn += quad_get(c[i >> 2], i);
        if (tn >= 0.0) tn = EPMACH / sigma;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = N - 1 - i;
            w[j >> 2] = (q == (j & 3)) ? tn : w[j >> 2];
            tn -= quad_get(c[j >> 2], j);
        }
        t = neg ? tn : t;
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) { w[s] = 0.0; dl[s] = 0.0; }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int si = i >> 2;
        const double v = quad_get(z[si], i);
        const double aii = quad_get(dg[si], i);
        const double wi = quad_get(w[si], i);
        double delta = quad_get(dl[si], i);  // (sigma < 0: formed above)
        if (any_pos) delta = neg ? delta : v / aii;
        const double tp = neg ? wi : t + delta * v;
        // alpha = tp / t (lane 0), beta = delta / tp (lane 1), gamma = t / tp (lane 2)
        const double num = (q == 0) ? tp : ((q == 1) ? delta : t);
        const double den = (q == 0) ? t : tp;
        const double quo = num / den;
        const double alpha = quad_get(quo, 0);
        dg[si] = (live && q == (i & 3)) ? alpha * aii : dg[si];
        if (i < N - 1) {
            const double beta = quad_get(quo, 1), gamma = quad_get(quo, 2);
            const bool big = alpha > 4.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (!slot_has<N>(s, i)) continue;
                const bool below = live && (q + 4 * s > i) && (q + 4 * s < N);
                const double u = Lr[s][i];
                const double zn = z[s] - v * u;
                const double a_big = gamma * u + beta * z[s];
                const double a_small = u + beta * zn;
                Lr[s][i] = below ? (big ? a_big : a_small) : u;
                z[s] = below ? zn : z[s];
            }
            t = tp;
        }
        OPTIK_SCHED_FENCE();
    }
}

// BFGS update of the LDL' factor with Powell damping (ik_slsqp.hpp:bfgs_update); u = g_new - g_old on
// entry (destroyed), s = the accepted step, both by joint.
template <int N>
OPTIK_DEV void bfgs_quad(bool live, double (&Lr)[QuadDims<N>::NS][QuadDims<N>::NM], double (&dg)[QuadDims<N>::NS],
                         const double (&sv)[QuadDims<N>::NS], double (&u)[QuadDims<N>::NS]) {
    constexpr int NS = QuadDims<N>::NS;
    const int q = quad_lane();
    double v[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) v[s] = 0.0;
    // v = L' s: v_i = s_i + sum_{j > i} l(i,j) s_j -- the terms of one sum live with the owners of rows j
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double p[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) p[s] = slot_has<N>(s, i) ? Lr[s][i] * sv[s] : 0.0;
        double h = 0.0;
#pragma unroll
        for (int j = i + 1; j < N; ++j) h += quad_get(p[j >> 2], j);
        const double vi = sv[i >> 2] + h;
        v[i >> 2] = (q == (i & 3)) ? vi : v[i >> 2];
    }
    // v = D v
#pragma unroll
    for (int s = 0; s < NS; ++s) v[s] = dg[s] * v[s];
    // v = L v: v_i += sum_{j < i} l(j,i) v_j with the v_j of before this pass -- row i holds every factor
    {
        double vb[N];
#pragma unroll
        for (int j = 0; j < N; ++j) vb[j] = quad_get(v[j >> 2], j);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            double h = 0.0;
#pragma unroll
            for (int j = 0; j < N - 1; ++j)
                if (slot_has<N>(s, j)) h = (j < q + 4 * s) ? h + Lr[s][j] * vb[j] : h;
            v[s] += h;
        }
    }
    double h1 = quad_dot<N, NS>(sv, u);
    const double h2 = quad_dot<N, NS>(sv, v);
    const double h3 = h2 * 0.2;
    {
        const bool damp = h1 < h3;
        const double h4 = (h2 - h3) / (h2 - h1);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double ud = u[s] * h4 + (1.0 - h4) * v[s];
            u[s] = damp ? ud : u[s];
        }
        h1 = damp ? h3 : h1;
    }
    OPTIK_SCHED_FENCE();
    ldl_quad<N>(live, Lr, dg, u, 1.0 / h1);
    OPTIK_SCHED_FENCE();
    ldl_quad<N>(live, Lr, dg, v, -1.0 / h2);
    OPTIK_SCHED_FENCE();
}

// ---- Kraft LSQ pieces (ik_slsqp.hpp: lsq_factor, lsq_bound_rows, ldp tail, lsq_finish) ------------

// E = D^1/2 L' by column (column j with joint j's owner: Ec[s][i] = E[i][row], i < row; Ed = E[row][row]),
// f = -E^-T g by joint, then Kraft's LSI Householder pass.  Returns 1, or 5 (E numerically singular);
// the same value in every lane of the quad.
template <int N>
OPTIK_DEV int lsq_factor_quad(const double (&Lr)[QuadDims<N>::NS][QuadDims<N>::NM], const double (&dg)[QuadDims<N>::NS],
                              const double (&g)[QuadDims<N>::NS], double (&Ec)[QuadDims<N>::NS][QuadDims<N>::NM],
                              double (&Ed)[QuadDims<N>::NS], double (&fv)[QuadDims<N>::NS]) {
    constexpr int NS = QuadDims<N>::NS;
    const int q = quad_lane();
    double sd[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) sd[s] = __builtin_sqrt(dg[s]);
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
        const double sdi = quad_get(sd[i >> 2], i);
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (slot_has<N>(s, i)) Ec[s][i] = Lr[s][i] * sdi;
    }
    // f_i = (g_i - sum_{k < i} E[k][i] f_k) / E[i][i]: column i holds every factor of its sum
    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { Ed[s] = sd[s]; acc[s] = 0.0; }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int si = i >> 2;
        const double fi = quad_get((g[si] - acc[si]) / sd[si], i);
        fv[si] = (q == (i & 3)) ? fi : fv[si];
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (slot_has<N>(s, i)) acc[s] = (q + 4 * s > i) ? acc[s] + Ec[s][i] * fi : acc[s];
        OPTIK_SCHED_FENCE();
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) fv[s] = -fv[s];

    // LSI: "QR" of the already-triangular E: row i's reflection depends on E[i][i] only, so every
    // owner prepares the reflections of its own rows (rows 0 .. N-2 have one) ...
    double up[NS], binv[NS];
    int act[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int r = q + 4 * s;
        const double p = Ed[s];
        double cl = __builtin_fabs(p);
        const bool nz = !(cl <= 0.0) && r < N - 1;
        const double clinv = 1.0 / cl;
        const double d = p * clinv;
        const double sm0 = d * d;
        cl *= __builtin_sqrt(sm0);
        if (p > 0.0) cl = -cl;
        up[s] = p - cl;
        Ed[s] = nz ? cl : Ed[s];
        const double b = up[s] * cl;
        const bool a = nz && !(b >= 0.0);
        act[s] = a ? 1 : 0;
        binv[s] = 1.0 / b;
        const double sm = fv[s] * up[s];
        if (a && sm != 0.0) fv[s] += (sm * binv[s]) * up[s];
    }
    // ... and row i's reflection is applied to E[i][j], j > i, by the owners of the columns j
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
        const double upi = quad_get(up[i >> 2], i), bi = quad_get(binv[i >> 2], i);
        const bool ai = quad_get(act[i >> 2], i) != 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (!slot_has<N>(s, i)) continue;
            const double sm = Ec[s][i] * upi;
            const bool on = ai && (q + 4 * s > i) && sm != 0.0;
            Ec[s][i] = on ? Ec[s][i] + (sm * bi) * upi : Ec[s][i];
        }
    }
    bool singular = false;
#pragma unroll
    for (int s = 0; s < NS; ++s) singular = singular || (q + 4 * s < N && !(__builtin_fabs(Ed[s]) >= EPMACH));
    return quad_any(singular) ? 5 : 1;
}

// Rows q, q + 4 of E^-1 (row[s][j], j >= the row; zero before it) and the two bound rows each gives
// (ik_slsqp.hpp:lsq_bound_rows).  The rows are independent recurrences: one or two per lane, all at
// once; E's entries come from their column owners.  Returns whether some h > 0 in the quad.
template <int N>
OPTIK_DEV bool bound_rows_quad(const double (&Ec)[QuadDims<N>::NS][QuadDims<N>::NM], const double (&Ed)[QuadDims<N>::NS],
                               const double (&fv)[QuadDims<N>::NS], const double (&lo)[QuadDims<N>::NS],
                               const double (&hi)[QuadDims<N>::NS], double (&row)[QuadDims<N>::NS][N],
                               double (&h_lo)[QuadDims<N>::NS], double (&h_hi)[QuadDims<N>::NS]) {
    constexpr int NS = QuadDims<N>::NS;
    const int q = quad_lane();
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double ejj = quad_get(Ed[j >> 2], j);
        double acc[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = 0.0;
#pragma unroll
        for (int k = 0; k < j; ++k) {
            const double ekj = quad_get(Ec[j >> 2][k], j);  // E[k][j]: entry k of column j
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s == 1 && k < 4) continue;  // rows 4.. start at column >= 4
                acc[s] = (k >= q + 4 * s) ? acc[s] + row[s][k] * ekj : acc[s];
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s == 1 && j < 4) { row[s][j] = 0.0; continue; }
            const int r = q + 4 * s;
            const double v = (((j == r) ? 1.0 : 0.0) - acc[s]) / ejj;
            row[s][j] = (j >= r) ? v : 0.0;
        }
        OPTIK_SCHED_FENCE();
    }
    bool need = false;
    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double fj = quad_get(fv[j >> 2], j);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s == 1 && j < 4) continue;
            acc[s] = (j >= q + 4 * s) ? acc[s] + row[s][j] * fj : acc[s];
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        h_lo[s] = lo[s] - acc[s];
        h_hi[s] = (-hi[s]) - (-acc[s]);
        need = need || (q + 4 * s < N && (h_lo[s] > 0.0 || h_hi[s] > 0.0));
    }
    return quad_any(need);
}

// LDP tail from the NNLS multipliers (oracle: lsq_dual): the transformed-space step,
// by joint.  ylo / yhi: multipliers of the lane's lower- / upper-bound columns.  Returns the mode.
template <int N>
OPTIK_DEV int ldp_quad(int mode, double rnorm, const double (&row)[QuadDims<N>::NS][N], const double (&h_lo)[QuadDims<N>::NS],
                       const double (&h_hi)[QuadDims<N>::NS], const double (&ylo)[QuadDims<N>::NS],
                       const double (&yhi)[QuadDims<N>::NS], double (&sv)[QuadDims<N>::NS]) {
    constexpr int NS = QuadDims<N>::NS;
    const int q = quad_lane();
    if (mode == 1 && rnorm <= 0.0) mode = 4;
    double plo[NS], phi[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { plo[s] = h_lo[s] * ylo[s]; phi[s] = h_hi[s] * yhi[s]; }
    double hy = 0.0;
#pragma unroll
    for (int r = 0; r < N; ++r) hy += quad_get(plo[r >> 2], r);
#pragma unroll
    for (int r = 0; r < N; ++r) hy += quad_get(phi[r >> 2], r);
    double fac = 1.0 - hy;
    const double d1 = 1.0 + fac;
    if (mode == 1 && d1 - 1.0 <= 0.0) mode = 4;
    fac = 1.0 / fac;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double a[NS], b[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { a[s] = row[s][j] * ylo[s]; b[s] = (-row[s][j]) * yhi[s]; }
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r <= j; ++r) acc += quad_get(a[r >> 2], r);
#pragma unroll
        for (int r = 0; r <= j; ++r) acc += quad_get(b[r >> 2], r);
        const double sj = fac * acc;
        sv[j >> 2] = (q == (j & 3)) ? sj : sv[j >> 2];
        OPTIK_SCHED_FENCE();
    }
    return mode;
}

// s (transformed space) -> s = E^-1 (s + f), clipped into [lo, hi] (ik_slsqp.hpp:lsq_finish).
template <int N>
OPTIK_DEV void lsq_finish_quad(const double (&Ec)[QuadDims<N>::NS][QuadDims<N>::NM], const double (&Ed)[QuadDims<N>::NS],
                               const double (&fv)[QuadDims<N>::NS], const double (&lo)[QuadDims<N>::NS],
                               const double (&hi)[QuadDims<N>::NS], double (&sv)[QuadDims<N>::NS]) {
    constexpr int NS = QuadDims<N>::NS;
    const int q = quad_lane();
#pragma unroll
    for (int s = 0; s < NS; ++s) sv[s] += fv[s];
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double p[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) p[s] = slot_has<N>(s, i) ? Ec[s][i] * sv[s] : 0.0;  // E[i][row] s_row
        double acc = 0.0;
#pragma unroll
        for (int j = i + 1; j < N; ++j) acc += quad_get(p[j >> 2], j);
        const int si = i >> 2;
        const double t = (sv[si] - acc) / Ed[si];
        sv[si] = (q == (i & 3)) ? t : sv[si];
        OPTIK_SCHED_FENCE();
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (sv[s] < lo[s]) sv[s] = lo[s];
        else if (sv[s] > hi[s]) sv[s] = hi[s];
    }
}

// ---- the wave: 16 quads, each solving restarts until the queue is dry ----------------------------

constexpr int QUADS_PER_WAVE = 64 / QUAD;

// doubles of LDS per wave: the NNLS blocks of its quads and the column of zeros (ik_nnls_quad.hpp) ...
template <int N>
constexpr int quad_wave_lds() { return nnls_quad_wave_lds<N>(); }
// ... and the best point so far / the previous iterate of every lane's joints ([slot][lane]: they are
// written when f improves / a line search ends and read when a restart is published -- not worth
// eight registers for the whole life of the kernel)
constexpr int quad_lane_lds() { return 4 * 64; }

// Early exit inside a trip (ik_nnls_quad.hpp: Stop): the restart's first-success word is looked at once per
// direction pass and once per loop trip of the NNLS as well -- a trip with a direction reset or a long active-set
// search takes several times an evaluation, and a launch with early exit ends when its slowest abandoned restart
// has noticed.  Every look uses the word asked for at the previous one (no wait) and asks again.  The reference
// abandons a restart at its next objective evaluation (lib.rs:308); abandoning it sooner changes nothing it returns.
struct QuadStop {
    static constexpr bool on = true;
    const WorkQueue *wq;       // the launch's queue record (LDS)
    const int &ia, &ib;        // the quad's keepers of the target slot and the restart number (ik_quad.hpp: quad_wave)
    bool enabled;              // the launch has first-success words (wave-uniform)
    bool hit;                  // the quad's restart has been overtaken
    unsigned long long seen;   // the word as last read
    OPTIK_DEV bool poll(bool running) {
        if (!enabled) return false;
        const WorkQueue &q = *launder_lds(wq);
        const unsigned long long index =
            q.restart_begin + (((unsigned long long)(unsigned)quad_get(ib, 1) << 32) | (unsigned)quad_get(ib, 0));
        const unsigned long long below = q.find_any ? ~0ull : index;
        const unsigned ts = (unsigned)quad_get(ia, 3);
        // (the quad's leader decides, as for the look before an evaluation: four lanes that saw different values
        // of the word would split the quad over two restarts)
        const bool now = quad_get((int)(running && seen < below), 0) != 0;
        hit = hit || now;
        if (running && !now) seen = __hip_atomic_load(q.first_success + ts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return now;
    }
};

template <int N, bool TIP>
OPTIK_DEV void quad_wave(const ChainDev &ch, const EvalParams &ep_in, const SolveParams &sp_in, const uint32_t (&key)[8],
                         const double (&scale)[MAX_DOF], const WorkQueue &wq_in,
                         double *nnls_lds /* quad_wave_lds<N>() doubles, the last 16 zero */,
                         double *lane_lds /* quad_lane_lds() doubles: x_best, x_prev of every lane */) {
    constexpr int NS = QuadDims<N>::NS, NM = QuadDims<N>::NM;
    constexpr int CPL = 4;
    const bool member = (int)((threadIdx.x & 63u) / QUAD) < wq_in.lanes;  // wq.lanes = restarts (quads) a wave holds at a time
    const double alfmin = 0.1;

    // xb[s * 64]: best point so far, xp[s * 64]: iterate of the last completed line search (the addresses
    // are formed where they are used, see wave_lane_now)
#define xb (lane_lds + wave_lane_now())
#define xp (lane_lds + 128 + wave_lane_now())
    // the quad's block of LDS: the NNLS matrix during a direction search, and during an evaluation the
    // parking place of what the evaluation does not touch (lane ql's i-th double at [4 i + ql])
#define blk (nnls_lds + (unsigned)(wave_lane_now() >> 2) * NnlsQuadGeom<N>::STRIDE)

    // SLSQP state of the quad's restart: by joint, by row, and the replicated scalars (names as in the oracle)
    double x[NS], x0[NS], g[NS], sv[NS];
    double Lr[NS][NM], dg[NS];
    // The replicated scalars of the restart are the same in the four lanes of the quad, so each is kept
    // by ONE of them and fetched (a DPP move) by the region of a trip that uses it -- four registers
    // across the evaluation and the direction search instead of twenty-two:
    //   pa: f0 | t0 | h3 | alpha      pb: minf | fprev | f | --
    //   ia: ireset | line | nevals | target slot      ib: restart number within the target, low | high | -- | --
    double pa = 0.0, pb = 0.0;
    int ia = 0, ib = 0;
    bool first = true;
    bool active = false, want = member;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        x[s] = 0.0; x0[s] = 0.0; g[s] = 0.0; sv[s] = 0.0; xb[s * 64] = 0.0; xp[s * 64] = 0.0; dg[s] = 1.0;
#pragma unroll
        for (int i = 0; i < NM; ++i) Lr[s][i] = 0.0;
    }
    OPTIK_PROF_DECL;  // (-DOPTIK_PROFILE: slots 0 refill, 1 eval, 4 bookkeeping + BFGS, 5 direction, 6 NNLS of it, 3 publish, 7 trips)

    QuadStop st{&wq_in, ia, ib, false, false, ~0ull};
    for (;;) {
        OPTIK_PROF_BEGIN();
        int32_t ret = 0;
        // ---- refill: quads without a restart pull the next work item (the leader fetches) ----------
        if (wave_any(want)) {
            // (the launch parameters live in LDS: every region of a trip re-reads what it needs through a
            // laundered pointer, so that none of them is carried -- and spilled -- across the other regions)
            const WorkQueue &wq = *reload_barrier_lds(&wq_in);
            unsigned long long it = fetch_items(wq.next_item, want && quad_lane_now() == 0);
            it = quad_get_u64(it, 0);
            // (the seed of the item's restart index, by every quad alike: the block's rounds move values
            // between the lanes of a quad, so they sit outside the per-quad branch)
            double seedq[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) seedq[s] = 0.0;
            // (target, restart) of the item: one division, in 32 bits whenever the launch has fewer than 2^32
            // items (a 64-bit division is ~150 instructions on this ISA)
            unsigned long long rq = 0;  // the item's restart number within its target
            unsigned tq = 0;            // ... and its target
            {
                const unsigned long long div = wq.restart_major ? wq.n_targets : wq.n_restarts;
                const unsigned long long itc = it < wq.total_items ? it : 0ull;
                unsigned long long quo;
                if (wq.total_items <= 0xffffffffull) quo = (unsigned long long)((unsigned)itc / (unsigned)div);
                else quo = itc / div;
                const unsigned long long rem = itc - quo * div;
                rq = wq.restart_major ? quo : rem;
                tq = (unsigned)(wq.restart_major ? rem : quo);
            }
            restart_seed_quad<N>(key, ch.lb, scale, wq.restart_begin + rq, seedq);
            if (want) {
                want = false;
                if (it < wq.total_items) {
                    const int qr = quad_lane_now();
                    const unsigned long long index = wq.restart_begin + rq;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        // lib.rs:366-370: restart 0 starts from the caller's seed
                        double v = seedq[s];
                        const int jc = (qr + 4 * s < N) ? qr + 4 * s : N - 1;
                        if (index == 0) v = wq.x0[(size_t)tq * N + jc];
                        x[s] = v; xb[s * 64] = v; xp[s * 64] = v; x0[s] = v; sv[s] = 0.0; g[s] = 0.0;
                    }
                    // f = f0 = t0 = h3 = 0, alpha = 1, minf = fprev = inf, ireset = line = nevals = 0
                    pa = (qr == 3) ? 1.0 : 0.0;
                    pb = (qr < 2) ? __builtin_huge_val() : 0.0;
                    ia = (qr == 3) ? (int)tq : 0;
                    ib = (qr == 0) ? (int)(unsigned)(rq & 0xffffffffull) : ((qr == 1) ? (int)(unsigned)(rq >> 32) : 0);
                    first = true;
                    active = true;
                }
            }
        }
        OPTIK_PROF_END(0);
        if (!wave_any(active || want)) break;
        OPTIK_PROF_COUNT(7, 1);

        const unsigned tslot = (unsigned)quad_get(ia, 3);
        {
        const unsigned rlo = (unsigned)quad_get(ib, 0), rhi = (unsigned)quad_get(ib, 1);
        if (active) {
            const WorkQueue &wq = *reload_barrier_lds(&wq_in);
            const unsigned long long index = wq.restart_begin + (((unsigned long long)rhi << 32) | rlo);
            // lib.rs:308: abandon when timed out or another restart of the target succeeded
            bool stop = false;
            if (wq.first_success) {
                const unsigned long long fs = __hip_atomic_load(wq.first_success + tslot, __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT);
                stop = wq.find_any ? (fs != ~0ull) : (fs < index);
                st.seen = fs;
            }
            if (wq.deadline && (unsigned long long)wall_clock64() > wq.deadline) stop = true;
            if (stop) ret = RES_FORCED_STOP;
        }
        st.enabled = (*reload_barrier_lds(&wq_in)).first_success != nullptr;
        st.hit = false;
        }
        // (the four lanes may have read first_success / the clock at different moments: the leader decides)
        ret = quad_get(ret, 0);
        const bool stepping = active && ret == 0;
        const bool do_eval = stepping;
        double gn[NS];
        double fn = 0.0;
        OPTIK_SCHED_FENCE();
        OPTIK_PROF_BEGIN();
        {
            // what the evaluation does not touch waits in the quad's (idle) block of LDS: the evaluation is
            // the register peak of the loop (~200 VGPRs on its own)
            int pi = 0;
            double *const bp = blk + quad_lane_now();  // the lane's i-th parked double: bp[4 i]
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                bp[4 * pi++] = x0[s]; bp[4 * pi++] = g[s]; bp[4 * pi++] = sv[s]; bp[4 * pi++] = dg[s];
#pragma unroll
                for (int i = 0; i < NM; ++i)
                    if (slot_has<N>(s, i)) bp[4 * pi++] = Lr[s][i];
            }
            OPTIK_SCHED_FENCE();
            // (the target pose is re-read for every evaluation: seven L1 / L2 hits instead of 14 registers)
            const double *target7 = (*reload_barrier_lds(&wq_in)).targets + (size_t)tslot * 7;
            const EvalParams &ep = *reload_barrier_lds(&ep_in);
            fn = eval_quad<N, TIP>(ch, ep, target7, x, gn, bp + 4 * pi);
            OPTIK_SCHED_FENCE();
            // (through a pointer the compiler knows nothing about: otherwise it forwards the stored values to
            // these loads, i.e. keeps them in registers across the evaluation after all)
            const double *pk = reload_barrier_lds((const double *)bp);
            pi = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                x0[s] = pk[4 * pi++]; g[s] = pk[4 * pi++]; sv[s] = pk[4 * pi++]; dg[s] = pk[4 * pi++];
#pragma unroll
                for (int i = 0; i < NM; ++i)
                    if (slot_has<N>(s, i)) Lr[s][i] = pk[4 * pi++];
            }
        }
        OPTIK_PROF_END(1);
        OPTIK_SCHED_FENCE();

        // ---- NLopt bookkeeping and Kraft's line search (labels 100 / 220), replicated scalars ------
        OPTIK_PROF_BEGIN();
        bool need_dir = false, reset = false, do_bfgs = false;
        const SolveParams &sp = *reload_barrier_lds(&sp_in);
        double u[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) u[s] = 0.0;
        bool sx_prev = false;
        if (xprev_live(sp) && wave_any(do_eval && !first)) {
            double xpv[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) xpv[s] = xp[s * 64];
            sx_prev = stop_x_quad<N>(sp, x, xpv);
        }
        // (this region's scalars, fetched from their keepers)
        double f = quad_get(pb, 2), minf = quad_get(pb, 0), fprev = quad_get(pb, 1);
        double alpha = quad_get(pa, 3);
        const double t0 = quad_get(pa, 1), h3 = quad_get(pa, 2);
        int nevals = quad_get(ia, 2);
        const int line = quad_get(ia, 1);
        if (do_eval) {
            f = fn;
            ++nevals;
            // NLopt: update best point so far; stopval is tested after every evaluation
            if (f < minf) {
                minf = f;
#pragma unroll
                for (int s = 0; s < NS; ++s) xb[s * 64] = x[s];
            }
            if (minf < sp.stopval) {
                ret = RES_STOPVAL_REACHED;
            } else if (nevals >= MAX_EVALS_CAP) {
                ret = RES_ITER_CAP;
            } else if (first) {
                // SLSQPB label 100/110: initialise, reset the BFGS matrix
                first = false;
#pragma unroll
                for (int s = 0; s < NS; ++s) g[s] = gn[s];
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
                        else if (xprev_live(sp) && sx_prev) ret = RES_XTOL_REACHED;
                    }
                    fprev = f;
#pragma unroll
                    for (int s = 0; s < NS; ++s) xp[s * 64] = x[s];
                    if (ret == 0 && nevals >= MAX_EVALS_CAP) ret = RES_ITER_CAP;
                    if (ret == 0) {
                        // label 260: BFGS update with u = g_new - g_old
#pragma unroll
                        for (int s = 0; s < NS; ++s) { u[s] = gn[s] - g[s]; g[s] = gn[s]; }
                        do_bfgs = true;
                        need_dir = true;
                    }
                }
            }
        }
        {
            const int qr = quad_lane_now();
            pb = (qr == 0) ? minf : ((qr == 1) ? fprev : ((qr == 2) ? f : pb));
            pa = (qr == 3) ? alpha : pa;
            ia = (qr == 2) ? nevals : ia;
        }
        OPTIK_SCHED_FENCE();
        if (wave_any(do_bfgs)) bfgs_quad<N>(do_bfgs, Lr, dg, sv, u);
        OPTIK_SCHED_FENCE();
        OPTIK_PROF_END(4);

        // ---- labels 110/130: (reset,) search direction, descent test: a wave-uniform loop, every
        // quad that needs a direction takes part in every step of a round
        OPTIK_PROF_BEGIN();
        while (wave_any(need_dir)) {
            OPTIK_PROF_COUNT(2, 1);  // (direction passes: more than one per trip when some quad has to reset and search again)
            const SolveParams &sp = *reload_barrier_lds(&sp_in);
            if (st.poll(need_dir)) { ret = RES_FORCED_STOP; need_dir = false; }
            if (!wave_any(need_dir)) break;
            bool pass = need_dir;
            const bool sx0 = stop_x_quad<N>(sp, x, x0);
            const int qd = quad_lane_now();
            if (wave_any(pass && reset)) {
                const double fd = quad_get(pb, 2), f0 = quad_get(pa, 0);
                int ireset = quad_get(ia, 0);
                if (pass && reset) ++ireset;
                ia = (qd == 0) ? ireset : ia;
                if (pass && reset) {
                    if (ireset > 5) {
                        // label 255 with acc = 0 -> mode 8; NLopt's relaxed test vs (f0, x0)
                        ret = RES_ROUNDOFF_LIMITED;
                        if (__builtin_fabs(fd - f0) < sp.ftol_abs && !__builtin_isinf(f0)) ret = RES_FTOL_REACHED;
                        else if (sx0) ret = RES_XTOL_REACHED;
                        need_dir = false;
                        pass = false;
                    } else {
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            dg[s] = 1.0;
#pragma unroll
                            for (int i = 0; i < NM; ++i) Lr[s][i] = 0.0;
                        }
                    }
                }
            }
            double Ec[NS][NM], Ed[NS], fv[NS], lo[NS], hi[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int jc = (qd + 4 * s < N) ? qd + 4 * s : N - 1;
                lo[s] = ch.lb[jc] - x[s];
                hi[s] = ch.ub[jc] - x[s];
                fv[s] = 0.0;
                Ed[s] = 1.0;
#pragma unroll
                for (int i = 0; i < NM; ++i) Ec[s][i] = 0.0;
            }
            OPTIK_SCHED_FENCE();
            int lmode = lsq_factor_quad<N>(Lr, dg, g, Ec, Ed, fv);
            OPTIK_SCHED_FENCE();
            double row[NS][N], h_lo[NS], h_hi[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int j = 0; j < N; ++j) row[s][j] = 0.0;
            bool need_nnls = bound_rows_quad<N>(Ec, Ed, fv, lo, hi, row, h_lo, h_hi);
            need_nnls = need_nnls && pass && lmode == 1;
            OPTIK_SCHED_FENCE();
            // ---- the bounded dual problems of this round, one per quad, all 64 lanes -----------
            double ylo[NS], yhi[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) { ylo[s] = 0.0; yhi[s] = 0.0; }
            int nmode = 1;
            double rnorm = 1.0;
            if (wave_any(need_nnls)) {
#ifdef OPTIK_PROFILE
                const unsigned long long t_nn = __builtin_readcyclecounter();
#endif
                // the lane's columns go to the quad's block: row r = q + 4 s of E^-1 is column r + 1 (lower
                // bound, h_lo below it) and, negated, column N + r + 1 (upper bound, h_hi below it)
                // (the lane number is made opaque here: the column ids and the dozen LDS addresses
                // derived from them are loop invariants the compiler would otherwise hoist and spill)
                const int qn = quad_lane_now();
                double *const bk = blk;
                int ids[CPL];
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    const int s = k & 1;
                    const bool neg = k >= 2;
                    ids[k] = 0x7fff;
                    if (s < NS) {
                        const int r = qn + 4 * s;
                        ids[k] = (r < N) ? (neg ? N : 0) + r + 1 : 0x7fff;
                        if (need_nnls && r < N) {
                            double *c = bk + NnlsQuadGeom<N>::CS * (ids[k] - 1);
#pragma unroll
                            for (int j = 0; j < N; ++j) {
                                const double v = row[s][j];
                                c[j] = neg ? ((j >= r) ? -v : 0.0) : v;
                            }
                            c[N] = neg ? h_hi[s] : h_lo[s];
                        }
                    }
                }
                int iters;
                double xv[CPL];
                nnls_quad<N, NoPipe, QuadStop>(need_nnls, ids, bk, nnls_lds + QUADS_PER_WAVE * NnlsQuadGeom<N>::STRIDE, xv,
                                               nmode, rnorm, iters, nullptr, &st);
#pragma unroll
                for (int s = 0; s < NS; ++s) { ylo[s] = xv[s]; yhi[s] = xv[2 + s]; }
#ifdef OPTIK_PROFILE
                OPTIK_PROF_COUNT(6, __builtin_readcyclecounter() - t_nn);
#endif
            }
            // (a restart overtaken while its sub-problem was being solved: abandoned here, its multipliers unused)
            if (st.hit && pass) { ret = RES_FORCED_STOP; need_dir = false; pass = false; }
            OPTIK_SCHED_FENCE();
            double sn[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) sn[s] = 0.0;
            if (wave_any(need_nnls)) {
                const int m2 = ldp_quad<N>(nmode, rnorm, row, h_lo, h_hi, ylo, yhi, sn);
                if (need_nnls) lmode = m2;
                else {
#pragma unroll
                    for (int s = 0; s < NS; ++s) sn[s] = 0.0;
                }
            }
            lsq_finish_quad<N>(Ec, Ed, fv, lo, hi, sn);
            OPTIK_SCHED_FENCE();
            const double gs = quad_dot<N, NS>(g, sn);
            const double fq = quad_get(pb, 2);
            const int qe = quad_lane_now();
            if (pass) {
                if (lmode != 1) {
                    // NLopt: modes 5,6,7 -> ROUNDOFF_LIMITED; 3,4,9 -> FAILURE
                    ret = (lmode == 5 || lmode == 6 || lmode == 7) ? RES_ROUNDOFF_LIMITED : RES_FAILURE;
                    need_dir = false;
                } else {
                    // (g is also Kraft's v: the gradient at the start of the line search)
#pragma unroll
                    for (int s = 0; s < NS; ++s) { sv[s] = sn[s]; x0[s] = x[s]; }
                    // f0 = t0 = f; h3 = gs - h1 * h4 with h1 = 0 (no constraints)
                    pa = (qe < 2) ? fq : ((qe == 2) ? gs : pa);
                    if (gs >= 0.0) {
                        reset = true;  // not a descent direction: reset B and repeat
                    } else {
                        ia = (qe == 1) ? 0 : ia;       // line = 0
                        pa = (qe == 3) ? 1.0 : pa;     // alpha = 1
                        need_dir = false;
                    }
                }
            }
        }
        OPTIK_PROF_END(5);
        OPTIK_PROF_BEGIN();
        const int qf = quad_lane_now();
        const double alpha_t = quad_get(pa, 3);
        if (stepping && ret == 0) {
            // label 190: next trial point x = x0 + alpha * s, clipped (NLopt)
            ia = (qf == 1) ? ia + 1 : ia;           // ++line
            pa = (qf == 2) ? alpha_t * pa : pa;     // h3 = alpha * h3
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int jc = (qf + 4 * s < N) ? qf + 4 * s : N - 1;
                sv[s] *= alpha_t;
                double xi = x0[s];
                xi += sv[s];
                const double lbs = ch.lb[jc], ubs = ch.ub[jc];
                if (xi < lbs) xi = lbs;
                else if (xi > ubs) xi = ubs;
                x[s] = xi;
            }
        }
        // ---- a restart ended: classify (lib.rs:376-379), publish, free the quad -------------------
        const bool ended = active && ret != 0;
        if (wave_any(ended)) {
            const SolveParams &sp = *reload_barrier_lds(&sp_in);
            const double minf = quad_get(pb, 0);
            const int nevals = quad_get(ia, 2);
            const unsigned long long rr = ((unsigned long long)(unsigned)quad_get(ib, 1) << 32) | (unsigned)quad_get(ib, 0);
            // where the restart's results go: the launch's outputs
            const WorkQueue &wq = *reload_barrier_lds(&wq_in);
            const double *const o_x0 = wq.x0;
            double *const o_x = wq.out_x, *const o_f = wq.out_f, *const o_key = wq.out_key;
            int32_t *const o_status = wq.out_status, *const o_evals = wq.out_evals;
            unsigned long long *const o_fs = wq.first_success;
            const unsigned long long o_restarts = wq.n_restarts, o_begin = wq.restart_begin, o_stride = wq.total_items;
            const bool o_quality = wq.quality != 0;
            const unsigned long long item = (unsigned long long)tslot * o_restarts + rr;  // output column
            const unsigned long long index = o_begin + rr;
            bool val[NS];
            int jc[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                val[s] = qf + 4 * s < N;
                jc[s] = val[s] ? qf + 4 * s : N - 1;
            }
            const bool success = (sp.ok_stopval && ret == RES_STOPVAL_REACHED)
                                 || (sp.ok_ftol && ret == RES_FTOL_REACHED)
                                 || (sp.ok_xtol && ret == RES_XTOL_REACHED);
            // selection key (lib.rs:402-407): Quality = ||x - x0||_2, Speed = index
            double kq = 0.0;
            if (wave_any(ended && o_quality)) {
                double d2[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double d = xb[s * 64] - ((ended && o_quality && val[s]) ? o_x0[(size_t)tslot * N + jc[s]] : 0.0);
                    d2[s] = d * d;
                }
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) acc += quad_get(d2[i >> 2], i);
                kq = __builtin_sqrt(acc);
            }
            if (ended) {
                if (o_x) {
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        if (val[s]) o_x[(size_t)jc[s] * o_stride + item] = xb[s * 64];
                }
                int first_in = 0;  // (leader) this restart is the first of its target to succeed
                if (qf == 0) {
                    if (o_f) o_f[item] = minf;
                    if (o_status) o_status[item] = ret;
                    if (o_evals) o_evals[item] = nevals;
                    double k = __builtin_huge_val();
                    if (success) {
                        if (o_quality) k = kq;
                        else {
                            k = (double)index;
                            if (o_fs) first_in = atomicMin(o_fs + tslot, index) == ~0ull ? 1 : 0;
                        }
                    }
                    if (o_key) o_key[item] = k;
                }
                {
                // a single call under the first-success rule: the answer goes to the host right away (WorkQueue::claim)
                const WorkQueue &wqc = *reload_barrier_lds(&wq_in);
                if (wqc.claim) {
                    first_in = quad_get(first_in, 0);
                    if (first_in) {
                        double *cx = reinterpret_cast<double *>(wqc.claim) + 3;
#pragma unroll
                        for (int s = 0; s < NS; ++s)
                            if (val[s]) cx[jc[s]] = xb[s * 64];
                        if (qf == 0) {
                            wqc.claim[1] = index;
                            reinterpret_cast<double *>(wqc.claim)[2] = minf;
                        }
                        __threadfence_system();
                        if (qf == 0)
                            __hip_atomic_store(wqc.claim, wqc.claim_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
                }
                active = false;
                want = true;
            }
        }
        OPTIK_PROF_END(3);
    }
    OPTIK_PROF_FLUSH(wq_in.prof);
#undef xb
#undef xp
#undef blk
}

}  // namespace optik
