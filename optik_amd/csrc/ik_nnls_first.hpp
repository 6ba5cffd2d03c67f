// ik_nnls_first.hpp -- the FIRST pass of Lawson-Hanson's NNLS on a lane's own bounded sub-problem, per lane.
//
// Of the bounded sub-problems the kernels solve, 45 % end after one solve pass with one active bound
// (nnls_pass_hist.py (a rounds 3-5 tool: git history): the column with the largest dual enters, its multiplier comes out positive, no other
// dual is positive afterwards).  The quad NNLS (ik_nnls_quad.hpp) spends two loop trips of a four-lane quad on each of
// them -- a third of its quad-trips -- at one wave per SIMD and sixteen unrelated problems per instruction.  Here
// every lane of the lane-per-restart form runs that first pass on its OWN problem, 64 problems per instruction,
// straight from its packed record in LDS, and says whether the problem is thereby solved; only the others go through
// the quads.
//
// This is nnls_quad specialised to nsetp = 0, npp1 = 1, b = e_m -- statement by statement (steps two .. five, six ..
// ten, then steps two .. four again), on whole columns with their structural zeros as explicit +0.0, so that every
// number formed is the number the quad NNLS forms: mode 1, one pass, the same multiplier, the same rnorm.  Whenever
// anything else would happen (the column is rejected, its multiplier is not positive, another dual is positive
// afterwards) the function answers "not solved" and the problem takes the usual way from the start.
#pragma once
// (included by ik_lane64.hpp after Lane64Geom, the layout of a lane's record)

namespace optik {

// Column j (1-based id) of the lane's problem enters an EMPTY active set: Householder construction with pivot row 1
// (step five), b := Q e_m, z(1) = b(1) / A(1, j) (steps six .. ten with alpha = 1).  Returns false when nnls_quad would
// not simply accept the column with a positive multiplier; otherwise w = the transformation's weights (up on the pivot
// row), b = Q e_m, up / ulp / hb / apply_live as nnls_quad names them, yv = the multiplier.
template <int N>
OPTIK_DEV bool first_pass_column(const double *rp, int j, double (&w)[N + 1], double (&b)[N + 1], double &up, double &ulp,
                                 double &hb, bool &apply_live, double &yv) {
    typedef Lane64Geom<N> G;
    constexpr int m = N + 1;
    const double factor = 0.01;
    const int jr = (j > N) ? j - N - 1 : j - 1;  // the row of E^-1 the column is
    const bool jneg = j > N;
    {
        const int tri = jr * N - (jr * (jr - 1)) / 2 - jr;  // G::g(jr, i) - i
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double e = rp[64 * (i >= jr ? tri + i : 0)];
            const double v = (i >= jr) ? e : 0.0;
            w[i] = jneg ? ((i >= jr) ? -v : 0.0) : v;
        }
        w[N] = jneg ? rp[64 * (G::NG + N + jr)] : rp[64 * (G::NG + jr)];
    }
    const double asave = w[0];
    w[0] = 0.0;  // (r > npp1 keeps, the pivot row and above are zeroed)
    double cl = __builtin_fabs(asave);
#pragma unroll
    for (int r = 0; r < m; ++r) {
        const double sm = __builtin_fabs(w[r]);
        cl = (sm > cl) ? sm : cl;
    }
    const bool pivot = (1 < m) && !(cl <= 0.0);
    up = 0.0;
    ulp = asave;
    {
        const double clinv = 1.0 / cl;
        double d = asave * clinv;
        double sm = d * d;
#pragma unroll
        for (int r = 0; r < m; ++r) {
            d = w[r] * clinv;
            sm += d * d;
        }
        double c2 = cl * __builtin_sqrt(sm);
        c2 = (asave > 0.0) ? -c2 : c2;
        up = pivot ? asave - c2 : up;
        ulp = pivot ? c2 : ulp;
    }
    // (no rows above the pivot: unorm = 0 and diff(unorm + t, unorm) > 0 is t > 0 on either of nnls_quad's routes)
    const double t = factor * __builtin_fabs(ulp);
    const bool ok1 = t > 0.0;
    const double hprod = up * ulp;
    apply_live = (1 < m) && !(__builtin_fabs(ulp) <= 0.0) && !(hprod >= 0.0);
    hb = apply_live ? 1.0 / hprod : 0.0;
    w[0] = up;
    // b := Q b with b = e_m
#pragma unroll
    for (int r = 0; r < m; ++r) b[r] = (r == m - 1) ? 1.0 : 0.0;
    double smb = 0.0;
#pragma unroll
    for (int r = 0; r < m; ++r) {
        const double pr = b[r] * w[r];
        smb = (r == 0) ? pr : smb + pr;
    }
    const bool actb = apply_live && ok1 && smb != 0.0;
    const double smhb = actb ? smb * hb : 0.0;
    const double bpiv = b[0];
    const double ztp = actb ? bpiv + smhb * up : bpiv;
    const double aztp = __builtin_fabs(ztp), aulp = __builtin_fabs(ulp);
    const bool tame = aztp >= 0x1p-500 && aztp <= 0x1p500 && aulp >= 0x1p-500 && aulp <= 0x1p500;
    const bool quo_pos = tame ? ((ztp > 0.0) == (ulp > 0.0)) : (ztp / ulp > 0.0);
    const bool found = ok1 && quo_pos;
    yv = 0.0;
    if (!(found && actb)) return false;  // (rejected, or b untouched: not the common case -- the quads take it from the start)
#pragma unroll
    for (int r = 0; r < m; ++r) b[r] = b[r] + smhb * w[r];
    // steps six .. ten with nsetp = 1: z(1) = b(1) / A(1, j), A(1, j) = ulp
    const double zi = b[0] / ulp;
    if (!(zi > 0.0)) return false;  // (a step length below one and a removal would follow)
    yv = (1.0 - 1.0) * 0.0 + 1.0 * zi;  // x(j) = (1 - alpha) x(j) + alpha z, alpha = 1
    return true;
}

// Column c (1-based id, not the one that entered) of the problem after that transformation, as nnls_quad leaves it in
// its block: cv + (cv . w) hb w when the transformation touches it, the column itself otherwise (structural zeros +0.0).
template <int N>
OPTIK_DEV void first_pass_other_column(const double *rp, int c, const double (&w)[N + 1], double hb, bool apply_live,
                                       double (&nv)[N + 1]) {
    typedef Lane64Geom<N> G;
    constexpr int m = N + 1;
    const int rr = (c > N) ? c - N - 1 : c - 1;
    const bool neg = c > N;
    const int tri = rr * N - (rr * (rr - 1)) / 2 - rr;
    double cv[m];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double e = rp[64 * (i >= rr ? tri + i : 0)];
        const double v = (i >= rr) ? e : 0.0;
        cv[i] = neg ? ((i >= rr) ? -v : 0.0) : v;
    }
    cv[N] = neg ? rp[64 * (G::NG + N + rr)] : rp[64 * (G::NG + rr)];
    double sm = 0.0;
#pragma unroll
    for (int r = 0; r < m; ++r) {
        const double pr = cv[r] * w[r];
        sm = (r == 0) ? pr : sm + pr;
    }
    const bool act = apply_live && sm != 0.0;
    const double smh = act ? sm * hb : 0.0;
#pragma unroll
    for (int r = 0; r < m; ++r) nv[r] = act ? cv[r] + smh * w[r] : cv[r];
}

// rp = rec_lds + lane (the lane's record: value v at rp[64 v]).  Returns FIRST_SOLVED when the problem ends after this
// pass: then y_id (1-based column id) has the only non-zero multiplier y_val, and rnorm is the residual norm;
// FIRST_WARM when column y_id entered with a positive multiplier but another dual is positive afterwards (a quad can
// start from there: first_pass_column / first_pass_other_column give it the state); FIRST_COLD otherwise.
enum : int { FIRST_SOLVED = 0, FIRST_WARM = 1, FIRST_COLD = 2 };
template <int N>
OPTIK_DEV int nnls_first_pass(const double *rp, int &y_id, double &y_val, double &rnorm) {
    typedef Lane64Geom<N> G;
    constexpr int m = N + 1, n = 2 * N;
    y_id = 1;
    y_val = 0.0;
    rnorm = 1.0;

    // ---- step two with b = e_m: the dual of column c is sum_r A(r, c) b(r), started at +0.0 -- its last row (h), or
    // NaN when the column holds a non-finite entry; step three: the largest positive one, ties to the lowest id
    double bw = 0.0;
    int j = 0;
    double zrow[N];  // +0.0, or NaN when row rr of E^-1 holds a non-finite entry (the negated column: the same)
#pragma unroll
    for (int rr = 0; rr < N; ++rr) {
        double z = 0.0;
#pragma unroll
        for (int i = rr; i < N; ++i) z += rp[64 * G::g(rr, i)] * 0.0;
        zrow[rr] = z;
        const double wlo = z + rp[64 * G::hlo(rr)] * 1.0;
        const bool bl = (wlo > 0.0) && (wlo > bw);  // (ids rise: a tie keeps the earlier one)
        bw = bl ? wlo : bw;
        j = bl ? rr + 1 : j;
    }
#pragma unroll
    for (int rr = 0; rr < N; ++rr) {
        const double whi = zrow[rr] + rp[64 * G::hhi(rr)] * 1.0;
        const bool bh = (whi > 0.0) && (whi > bw);
        bw = bh ? whi : bw;
        j = bh ? N + rr + 1 : j;
    }
    if (j == 0) return FIRST_COLD;  // (no positive dual: the caller does not send such a problem here; the quads settle it)

    // ---- step five, b := Q b, steps six .. ten (first_pass_column: shared with the quads' warm start)
    double w[m], b[m];
    double up, ulp, hb, yv;
    bool apply_live;
    if (!first_pass_column<N>(rp, j, w, b, up, ulp, hb, apply_live, yv)) return FIRST_COLD;

    // (What is left out of the sums below are products with a column's structural zeros: signed zeros, which change a sum
    // only when it is a zero itself -- and a zero sum decides the same way whatever its sign: `sm != 0`, `sdot > 0`.
    // An untouched column -- sm == 0 -- gets smh = 0 and cv + 0 w = cv up to the sign of a zero.)
    bool more = false;
    int npos = 0;  // duals that are positive after the pass: how many more columns are likely to enter
#pragma unroll
    for (int c = 1; c <= n; ++c) {
        const int rr = (c > N) ? c - N - 1 : c - 1;
        const bool neg = c > N;
        double cv[m];
#pragma unroll
        for (int i = rr; i < N; ++i) {
            const double e = rp[64 * G::g(rr, i)];
            cv[i] = neg ? -e : e;
        }
        cv[N] = neg ? rp[64 * G::hhi(rr)] : rp[64 * G::hlo(rr)];
        double sm = 0.0;
#pragma unroll
        for (int r = rr; r < m; ++r) {
            const double pr = cv[r] * w[r];
            sm = (r == rr) ? pr : sm + pr;
        }
        const bool act = apply_live && sm != 0.0;
        const double smh = act ? sm * hb : 0.0;
        double sdot = 0.0;
#pragma unroll
        for (int r = 0; r < m; ++r) {
            const double nv = (r >= rr) ? cv[r] + smh * w[r] : smh * w[r];
            sdot += nv * ((r >= 1) ? b[r] : 0.0);
        }
        more = more || (c != j && sdot > 0.0);
        npos += (c != j && sdot > 0.0) ? 1 : 0;
        OPTIK_SCHED_FENCE();  // (one column at a time: interleaved, the fourteen of them keep ~100 doubles live)
    }
    y_id = j;
    if (more) { y_val = (double)npos; return FIRST_WARM; }  // (FIRST_WARM: y_val carries that count, for the ranking)

    // ---- rnorm = ||b(2 .. m)||, as residual_norm forms it
    double xmax = 0.0;
#pragma unroll
    for (int r = 1; r < m; ++r) {
        const double av = __builtin_fabs(b[r]);
        if (av > xmax) xmax = av;
    }
    double rn = 0.0;
    if (xmax != 0.0) {
        const double scale = 1.0 / xmax;
        double sum = 0.0;
#pragma unroll
        for (int r = 1; r < m; ++r) {
            const double xs = scale * b[r];
            sum += xs * xs;
        }
        rn = xmax * __builtin_sqrt(sum);
    }
    y_val = yv;
    rnorm = rn;
    return FIRST_SOLVED;
}

}  // namespace optik
