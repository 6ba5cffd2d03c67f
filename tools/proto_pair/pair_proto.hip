// pair_proto.hip -- PROTOTYPES (not product code): the two heaviest per-lane phases of the lane-per-restart solver
// restated for TWO lanes per restart, to price a "second wave per SIMD" form from compiled code (VERDICT r5 item 1;
// DESIGN.md section 5.1, the table "two lanes per restart").
//
// A pair = lanes 2k (p = 0) and 2k + 1 (p = 1) of a wave; joint / row j belongs to lane j & 1 (slot j >> 1: four
// slots, the odd lane's fourth is padding for n = 7).  What can be split is split (the joint quaternions, the Jacobian
// columns / gradient components, the rows of the packed factor and of every vector); what is a chain through all joints
// or a sum whose ORDER is part of the bit-exactness contract is replicated in both lanes -- the same rule the quad solver
// follows with four lanes.  The kernels below compute the same bits as the product's per-lane functions (checked on the
// GPU by tools/pair_prototype.py --run); their registers / instruction counts are what the table quotes.
//
//   eval_pair_kernel<N, TIP>     eval_fg_stream's arithmetic (ik_eval.hpp), 32 configurations per wave
//   eval_lane_kernel<N, TIP>     the product's eval_fg_stream, 64 per wave (the like-for-like reference in this object)
//   bfgs_pair_kernel<N>          bfgs_update's arithmetic (ik_slsqp.hpp), state split by rows
//   bfgs_lane_kernel<N>          the product's bfgs_update
#include <hip/hip_runtime.h>

#include "ik_launch.hpp"
#include "ik_lane.hpp"
#include "ik_host_params.hpp"

namespace optik {

// value of the pair's even / odd lane, in both lanes (DPP quad_perm [0,0,2,2] / [1,1,3,3])
OPTIK_DEV double pair_even(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0xA0, 0xf, 0xf, true); }
OPTIK_DEV double pair_odd(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0xF5, 0xf, 0xf, true); }
OPTIK_DEV double pair_of(double v, int owner) { return owner ? pair_odd(v) : pair_even(v); }  // owner: a constant after unrolling
OPTIK_DEV Q4 pair_of(const Q4 q, int owner) { return Q4{pair_of(q.i, owner), pair_of(q.j, owner), pair_of(q.k, owner), pair_of(q.w, owner)}; }

template <int N>
struct PairGeom {
    static constexpr int NH = (N + 1) / 2;  // slots per lane
};

// ---- evaluation ------------------------------------------------------------------------------------------------------
// qown[s] = q[p + 2 s].  gsink(s, v): gradient component p + 2 s (the odd lane's last slot is padding when N is odd).
template <int N, bool TIP, class GSink>
OPTIK_DEV double eval_fg_pair(const ChainDev &ch, const EvalParams &ep, const Pose target, const double (&qown)[PairGeom<N>::NH],
                              int p, GSink &&gsink) {
    constexpr int NH = PairGeom<N>::NH;
    // Per slot: the own joint's quaternion origin * from_axis_angle (kinematics.rs:245-248) -- split: one sincos per lane
    // for two joints -- then the chain product through both joints of the slot -- replicated: it runs through every joint
    // in order.  Each lane keeps the frames of its own joints (orientation and position: the product's register diet, which
    // walks the chain again for the positions, needs the orientations of ALL joints).
    Q4 tfq[NH];
    V3 tft[NH];
    Pose state, ee;
    {
#pragma unroll
        for (int s = 0; s < NH; ++s) {
            const int jo = (p + 2 * s < N) ? p + 2 * s : N - 1;  // (padding slot: any valid joint)
            double sn, cs;
            sincos_dev(qown[s] / 2.0, sn, cs);
            const Q4 local{ch.axis[jo][0] * sn, ch.axis[jo][1] * sn, ch.axis[jo][2] * sn, cs};
            const Q4 jq = qmul(Q4{ch.origin[jo][3], ch.origin[jo][4], ch.origin[jo][5], ch.origin[jo][6]}, local);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = 2 * s + h;
                if (j >= N) continue;
                Pose jt;
                jt.t = V3{ch.origin[j][0], ch.origin[j][1], ch.origin[j][2]};
                jt.q = pair_of(jq, h);
                state = (j == 0) ? jt : pose_mul(state, jt);
                if (h == 0) { tfq[s] = state.q; tft[s] = state.t; }  // (the even joint first; the odd one replaces it in the odd lane)
                else {
                    const bool mine = p == 1;
                    tfq[s] = Q4{mine ? state.q.i : tfq[s].i, mine ? state.q.j : tfq[s].j, mine ? state.q.k : tfq[s].k,
                                mine ? state.q.w : tfq[s].w};
                    tft[s] = V3{mine ? state.t.x : tft[s].x, mine ? state.t.y : tft[s].y, mine ? state.t.z : tft[s].z};
                }
            }
            OPTIK_SCHED_FENCE_EVAL();
        }
        if (TIP) state = pose_mul(state, load_pose(ch.origin[N]));
        ee = ep.has_ee_offset ? pose_mul(state, load_pose(ep.ee_offset)) : state;
    }
    // error terms: replicated (a serial computation on 7 numbers)
    const Pose X = pose_inv_mul(target, ee);
    const V3 w = so3_log(X.q);
    const RotTerms rt = rot_terms(w);
    const M3 Jr = so3_right_jacobian(rt);
    const M3 Qm = se3_q_matrix(rt, X.t, Jr);
    const V3 elin = se3_log_linear(rt, X.t);
    V3 fl = elin, fa = w;
    if (!ep.skip_lin) fl = weight_block(target.q, elin, ep.w_lin);
    if (!ep.skip_ang) fa = weight_block(target.q, w, ep.w_ang);
    V3 gl = fl, ga = fa;
    if (!ep.grad_same_as_value) {
        gl = elin; ga = w;
        if (!ep.skip_lin2) gl = weight_block(target.q, elin, ep.w_lin2);
        if (!ep.skip_ang2) ga = weight_block(target.q, w, ep.w_ang2);
    }
    const double e2[6] = {2.0 * gl.x, 2.0 * gl.y, 2.0 * gl.z, 2.0 * ga.x, 2.0 * ga.y, 2.0 * ga.z};
    const double ef[6] = {fl.x, fl.y, fl.z, fa.x, fa.y, fa.z};
    double f = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) f += ef[i] * ef[i];
    OPTIK_SCHED_FENCE_EVAL();
    // Jacobian columns and gradient components of the own joints: split
    const Q4 eeqc = qconj(ee.q);
#pragma unroll
    for (int s = 0; s < NH; ++s) {
        const int k = (p + 2 * s < N) ? p + 2 * s : N - 1;
        const V3 ax{ch.axis[k][0], ch.axis[k][1], ch.axis[k][2]};
        const V3 angular = qrot(tfq[s], ax);
        const V3 d{ee.t.x - tft[s].x, ee.t.y - tft[s].y, ee.t.z - tft[s].z};
        const V3 linear = cross(angular, d);
        const V3 al = qrot(eeqc, angular);
        const V3 ll = qrot(eeqc, linear);
        const double lin[3] = {ll.x, ll.y, ll.z};
        const double ang[3] = {al.x, al.y, al.z};
        double jt[6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m) acc += Jr.m[r][m] * lin[m];
#pragma unroll
            for (int m = 0; m < 3; ++m) acc += Qm.m[r][m] * ang[m];
            jt[r] = acc;
            double acc2 = 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m) acc2 += Jr.m[r][m] * ang[m];
            jt[r + 3] = acc2;
        }
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) acc += e2[r] * jt[r];
        gsink(s, acc);
        OPTIK_SCHED_FENCE_EVAL();
    }
    return f;
}

// Occupancy as in the solvers: a static LDS block pins the workgroups per CU -- 40 KB per single-wave workgroup = one wave
// per SIMD (where the lane-per-restart solver runs), 20 KB = two (where a pair form would).  keep_lds() keeps it allocated.
template <int BYTES>
OPTIK_DEV void keep_lds(const void *flag) {
    __shared__ double pad[BYTES / 8];
    if (flag == (const void *)1) { pad[threadIdx.x] = 1.0; asm volatile("" :: "v"(pad[threadIdx.x ^ 1])); }
}
constexpr int LDS_ONE_WAVE = 39 * 1024, LDS_TWO_WAVES = 19 * 1024;

struct ProtoEval {
    const ChainDev *chain;
    EvalParams ep;
    double target[7];
    const double *q;  // [n][B]
    long long B;
    double *f;        // [B]
    double *g;        // [n][B]
};

// (two waves per SIMD is what the pair form is for: at most 256 registers)
template <int N, bool TIP>
__global__ __launch_bounds__(64, 2) void eval_pair_kernel(const ProtoEval a) {
    keep_lds<LDS_TWO_WAVES>(a.q);
    __shared__ ChainDev sch;
    stage_chain(sch, a.chain);
    constexpr int NH = PairGeom<N>::NH;
    const Pose target = load_pose(a.target);
    const int p = (int)(threadIdx.x & 1u);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < 2 * a.B; t += (long long)gridDim.x * blockDim.x) {
        const long long b = t >> 1;
        double qown[NH];
#pragma unroll
        for (int s = 0; s < NH; ++s) qown[s] = (p + 2 * s < N) ? a.q[(size_t)(p + 2 * s) * a.B + b] : 0.0;
        const double f = eval_fg_pair<N, TIP>(sch, a.ep, target, qown, p, [&](int s, double v) {
            if (p + 2 * s < N) a.g[(size_t)(p + 2 * s) * a.B + b] = v;
        });
        if (p == 0) a.f[b] = f;
    }
}

template <int N, bool TIP>
__global__ __launch_bounds__(64, 1) void eval_lane_kernel(const ProtoEval a) {
    keep_lds<LDS_ONE_WAVE>(a.q);
    __shared__ ChainDev sch;
    stage_chain(sch, a.chain);
    const Pose target = load_pose(a.target);
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.B; b += (long long)gridDim.x * blockDim.x) {
        double q[N];
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = a.q[(size_t)i * a.B + b];
        const double f = eval_fg_stream<N, TIP>(sch, a.ep, target, q, [&](int k, double v) { a.g[(size_t)k * a.B + b] = v; });
        a.f[b] = f;
    }
}

// ---- BFGS update of the packed LDL' factor -----------------------------------------------------------------------------
// Entry (row j, column i), j >= i, of the packed factor belongs to the lane that owns ROW j (lane j & 1), as do the
// components j of s, u, v, z, w: the rank-one sweeps over a column's rows are local.  The diagonal entry and the pivot
// component of step i go to both lanes (two broadcasts), the scalar recurrences (t, tp, alpha, beta, gamma: three divisions
// per step) are replicated, and so is every ordered sum over all rows (h1, h2, the columns of L' s): their products are
// formed by the owners, the additions run in both lanes in the oracle's order.
//
// Storage per lane: lown[c][s] = l(row p + 2 s, column c) for rows >= c (unused entries are never read), so[s] = s[p + 2 s] ...
template <int N>
struct PairL {
    static constexpr int NH = PairGeom<N>::NH;
    double v[N][NH];  // v[c][s]: entry (row p + 2 s, column c); meaningful when p + 2 s >= c
};

// x[j] for a row j that is a constant after unrolling, from the lane that owns it
template <int N>
OPTIK_DEV double row_of(const double (&x)[PairGeom<N>::NH], int j) { return pair_of(x[j >> 1], j & 1); }

template <int N>
OPTIK_DEV void ldl_update_pair(PairL<N> &a, double (&z)[PairGeom<N>::NH], double sigma, int p) {
    constexpr int NH = PairGeom<N>::NH;
    if (sigma == 0.0) return;
    double w[NH];
#pragma unroll
    for (int s = 0; s < NH; ++s) w[s] = 0.0;
    double t = 1.0 / sigma;
    if (sigma < 0.0) {
#pragma unroll
        for (int s = 0; s < NH; ++s) w[s] = z[s];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double v = row_of<N>(w, i);
            const double aii = pair_of(a.v[i][i >> 1], i & 1);
            t += v * v / aii;
#pragma unroll
            for (int s = 0; s < NH; ++s) {
                const bool below = p + 2 * s > i;   // rows j > i
                const double nw = w[s] - v * a.v[i][s];
                w[s] = below ? nw : w[s];
            }
            OPTIK_SCHED_FENCE_SLSQP();
        }
        if (t >= 0.0) t = EPMACH / sigma;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = N - 1 - i;
            const double u = row_of<N>(w, j);
            const double ajj = pair_of(a.v[j][j >> 1], j & 1);
            // w[j] = t in the owner
            {
                const bool own = (j & 1) == p;
                w[j >> 1] = own ? t : w[j >> 1];
            }
            t -= u * u / ajj;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double v = row_of<N>(z, i);
        const double aii = pair_of(a.v[i][i >> 1], i & 1);
        const double delta = v / aii;
        const double wi = row_of<N>(w, i);
        const double tp = (sigma < 0.0) ? wi : t + delta * v;
        const double alpha = tp / t;
        {
            const bool own = (i & 1) == p;
            const double na = alpha * aii;
            a.v[i][i >> 1] = own ? na : a.v[i][i >> 1];
        }
        if (i < N - 1) {
            const double beta = delta / tp;
            if (alpha > 4.0) {
                const double gamma = t / tp;
#pragma unroll
                for (int s = 0; s < NH; ++s) {
                    const bool below = p + 2 * s > i;
                    const double u = a.v[i][s];
                    const double na = gamma * u + beta * z[s];
                    const double nz = z[s] - v * u;
                    a.v[i][s] = below ? na : a.v[i][s];
                    z[s] = below ? nz : z[s];
                }
            } else {
#pragma unroll
                for (int s = 0; s < NH; ++s) {
                    const bool below = p + 2 * s > i;
                    const double nz = z[s] - v * a.v[i][s];
                    const double na = a.v[i][s] + beta * nz;
                    z[s] = below ? nz : z[s];
                    a.v[i][s] = below ? na : a.v[i][s];
                }
            }
            t = tp;
        }
        OPTIK_SCHED_FENCE_SLSQP();
    }
}

template <int N>
OPTIK_DEV void bfgs_update_pair(PairL<N> &l, const double (&s)[PairGeom<N>::NH], double (&u)[PairGeom<N>::NH], int p) {
    constexpr int NH = PairGeom<N>::NH;
    double v[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) v[k] = 0.0;
    // v = L D L' s.  Row i of L' s: s[i] + sum_{j > i} l(j, i) s[j], the products by the owners of the rows j, the sum in order
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double pr[NH];
#pragma unroll
        for (int k = 0; k < NH; ++k) pr[k] = l.v[i][k] * s[k];  // (row p + 2 k, column i)
        double h = 0.0;
#pragma unroll
        for (int j = i + 1; j < N; ++j) h += row_of<N>(pr, j);
        const double vi = row_of<N>(s, i) + h;
        const bool own = (i & 1) == p;
        v[i >> 1] = own ? vi : v[i >> 1];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool own = (i & 1) == p;
        const double d = l.v[i][i >> 1] * v[i >> 1];
        v[i >> 1] = own ? d : v[i >> 1];
    }
    // v[i] += sum_{j < i} l(i, j) v[j]: the entries of row i are the owner's, the v[j] come from both lanes
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double h = 0.0;
#pragma unroll
        for (int j = 0; j < i; ++j) h += l.v[j][i >> 1] * row_of<N>(v, j);
        const bool own = (i & 1) == p;
        const double nv = v[i >> 1] + h;
        v[i >> 1] = own ? nv : v[i >> 1];
    }
    double h1 = 0.0, h2 = 0.0;
    {
        double p1[NH], p2[NH];
#pragma unroll
        for (int k = 0; k < NH; ++k) { p1[k] = s[k] * u[k]; p2[k] = s[k] * v[k]; }
#pragma unroll
        for (int i = 0; i < N; ++i) h1 += row_of<N>(p1, i);
#pragma unroll
        for (int i = 0; i < N; ++i) h2 += row_of<N>(p2, i);
    }
    const double h3 = h2 * 0.2;
    if (h1 < h3) {
        const double h4 = (h2 - h3) / (h2 - h1);
        h1 = h3;
#pragma unroll
        for (int k = 0; k < NH; ++k) u[k] *= h4;
#pragma unroll
        for (int k = 0; k < NH; ++k) u[k] += (1.0 - h4) * v[k];
    }
    OPTIK_SCHED_FENCE_SLSQP();
    ldl_update_pair<N>(l, u, 1.0 / h1, p);
    OPTIK_SCHED_FENCE_SLSQP();
    ldl_update_pair<N>(l, v, -1.0 / h2, p);
    OPTIK_SCHED_FENCE_SLSQP();
}

struct ProtoBfgs {
    const double *l;  // [NL][B]  packed factor, column-packed as the product's (lidx)
    const double *s;  // [n][B]
    const double *u;  // [n][B]
    long long B;
    double *lout;     // [NL][B]
};

template <int N>
__global__ __launch_bounds__(64, 2) void bfgs_pair_kernel(const ProtoBfgs a) {
    keep_lds<LDS_TWO_WAVES>(a.l);
    constexpr int NH = PairGeom<N>::NH;
    const int p = (int)(threadIdx.x & 1u);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < 2 * a.B; t += (long long)gridDim.x * blockDim.x) {
        const long long b = t >> 1;
        PairL<N> l;
        double s[NH], u[NH];
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            const int j = p + 2 * k;
            s[k] = (j < N) ? a.s[(size_t)j * a.B + b] : 0.0;
            u[k] = (j < N) ? a.u[(size_t)j * a.B + b] : 0.0;
#pragma unroll
            for (int c = 0; c < N; ++c) l.v[c][k] = (j < N && j >= c) ? a.l[(size_t)lidx<N>(c, j) * a.B + b] : 0.0;
        }
        bfgs_update_pair<N>(l, s, u, p);
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            const int j = p + 2 * k;
#pragma unroll
            for (int c = 0; c < N; ++c)
                if (j < N && j >= c) a.lout[(size_t)lidx<N>(c, j) * a.B + b] = l.v[c][k];
        }
    }
}

template <int N>
__global__ __launch_bounds__(64, 1) void bfgs_lane_kernel(const ProtoBfgs a) {
    keep_lds<LDS_ONE_WAVE>(a.l);
    constexpr int NL = N * (N + 1) / 2;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.B; b += (long long)gridDim.x * blockDim.x) {
        double l[NL], s[N], u[N];
#pragma unroll
        for (int i = 0; i < NL; ++i) l[i] = a.l[(size_t)i * a.B + b];
#pragma unroll
        for (int i = 0; i < N; ++i) { s[i] = a.s[(size_t)i * a.B + b]; u[i] = a.u[(size_t)i * a.B + b]; }
        bfgs_update<N>(l, s, u);
#pragma unroll
        for (int i = 0; i < NL; ++i) a.lout[(size_t)i * a.B + b] = l[i];
    }
}

}  // namespace optik

using namespace optik;

// C entry points for tools/pair_prototype.py --run: HOST buffers in and out, the kernel timed over `reps` launches.
namespace {
template <class T>
struct DevBuf {
    T *p = nullptr;
    DevBuf(const T *host, size_t n) { (void)hipMalloc(&p, n * sizeof(T)); if (host) (void)hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice); }
    ~DevBuf() { if (p) (void)hipFree(p); }
};
template <class F>
float timed_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / (float)reps;
}
}  // namespace

extern "C" int proto_eval(int pair, int tip, const ChainDev *chain_host, const double *target7, const double *q,
                          long long B, double *f, double *g, int reps, float *ms) {
    EvalParams ep_;
    const double one[3] = {1.0, 1.0, 1.0};
    optik::hostparams::make_eval_params(one, one, nullptr, ep_);
    const EvalParams *ep = &ep_;
    DevBuf<ChainDev> dch(chain_host, 1);
    DevBuf<double> dq(q, (size_t)7 * B), df(nullptr, (size_t)B), dg(nullptr, (size_t)7 * B);
    ProtoEval a;
    a.chain = dch.p;
    a.ep = *ep;
    for (int i = 0; i < 7; ++i) a.target[i] = target7[i];
    a.q = dq.p; a.B = B; a.f = df.p; a.g = dg.p;
    const int grid = 1024 * 8;
    *ms = timed_ms([&] {
        if (pair) {
            if (tip) hipLaunchKernelGGL((eval_pair_kernel<7, true>), dim3(grid), dim3(64), 0, nullptr, a);
            else hipLaunchKernelGGL((eval_pair_kernel<7, false>), dim3(grid), dim3(64), 0, nullptr, a);
        } else {
            if (tip) hipLaunchKernelGGL((eval_lane_kernel<7, true>), dim3(grid), dim3(64), 0, nullptr, a);
            else hipLaunchKernelGGL((eval_lane_kernel<7, false>), dim3(grid), dim3(64), 0, nullptr, a);
        }
    }, reps);
    (void)hipMemcpy(f, df.p, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost);
    (void)hipMemcpy(g, dg.p, sizeof(double) * (size_t)7 * B, hipMemcpyDeviceToHost);
    return (int)hipDeviceSynchronize();
}

extern "C" int proto_bfgs(int pair, const double *l, const double *s, const double *u, long long B, double *lout, int reps, float *ms) {
    DevBuf<double> dl(l, (size_t)28 * B), ds(s, (size_t)7 * B), du(u, (size_t)7 * B), dout(nullptr, (size_t)28 * B);
    ProtoBfgs a{dl.p, ds.p, du.p, B, dout.p};
    const int grid = 1024 * 8;
    *ms = timed_ms([&] {
        if (pair) hipLaunchKernelGGL((bfgs_pair_kernel<7>), dim3(grid), dim3(64), 0, nullptr, a);
        else hipLaunchKernelGGL((bfgs_lane_kernel<7>), dim3(grid), dim3(64), 0, nullptr, a);
    }, reps);
    (void)hipMemcpy(lout, dout.p, sizeof(double) * (size_t)28 * B, hipMemcpyDeviceToHost);
    return (int)hipDeviceSynchronize();
}
