// ik_kernels.hip -- gfx950 kernels and the C ABI of include/optik_hip.h.
//
// Kernels (all f64, one configuration / restart per lane, 64-lane workgroups):
//   ik_solve_kernel   the hot path: seed -> SLSQP restart -> status/x/f, plus the
//                     per-wave argmin of the selection key (wavefront shuffles)
//   ik_select_kernel  reduces the per-tile winners of each target (lib.rs:397-413)
//   eval_batch_kernel objective + gradient for a batch of configurations
//   fk_batch_kernel   end-effector pose (+ body Jacobian) for a batch
//   seed_batch_kernel ChaCha8 restart seeds
//   probe_kernel      elementary functions (test hook)
// No CPU fallback exists: every entry point fails loudly without a device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/optik_hip.h"
#include "ik_solve.hpp"

using namespace optik;

// ---------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------

namespace {

constexpr int WAVE = 64;

struct TileRec {
    unsigned long long idx;  // winning restart index in the tile, ~0 if none
    double key;
};

struct SolveLaunch {
    const ChainDev *chain;
    EvalParams ep;
    SolveParams sp;
    uint32_t key[8];               // ChaCha key = seed_from_u64(42)
    double scale[MAX_DOF];         // rand UniformFloat scale per joint
    const double *targets;         // [T][7]
    const double *x0;              // [T][n]
    unsigned long long restart_begin;
    unsigned long long n_restarts;  // per target
    int tiles_per_target;
    int n_tiles;
    int quality;                   // selection key: 1 = ||x - x0||, 0 = index
    int pad;
    double *out_x;                 // [n][T*R]
    double *out_f;
    int32_t *out_status;
    int32_t *out_evals;
    TileRec *tile_recs;            // [n_tiles]
    unsigned long long *first_success;  // [T] or nullptr
    unsigned long long deadline_ticks;  // relative, 0 = none
};

__device__ __forceinline__ void stage_chain(ChainDev &dst, const ChainDev *src) {
    constexpr int ND = (int)(sizeof(ChainDev) / sizeof(double));
    static_assert(sizeof(ChainDev) % sizeof(double) == 0, "ChainDev is a whole number of doubles");
    const double *s = reinterpret_cast<const double *>(src);
    double *d = reinterpret_cast<double *>(&dst);
    for (int i = threadIdx.x; i < ND; i += blockDim.x) d[i] = s[i];
    __syncthreads();
}

// (valid, key, idx) argmin across the wave: smaller key wins, ties -> smaller idx.
__device__ __forceinline__ void wave_argmin(double &key, unsigned long long &idx) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double okey = __shfl_xor(key, off, WAVE);
        const unsigned long long oidx = __shfl_xor(idx, off, WAVE);
        const bool take = (oidx != ~0ull) && (idx == ~0ull || okey < key || (okey == key && oidx < idx));
        if (take) { key = okey; idx = oidx; }
    }
}

template <int N, bool TIP>
__global__ __launch_bounds__(WAVE) void ik_solve_kernel(const SolveLaunch a) {
    __shared__ ChainDev sch;
    __shared__ double nnls_lds[NnlsLayout<N>::SLOTS * WAVE];
    stage_chain(sch, a.chain);
    const int lane = threadIdx.x;
    const NnlsWs<N> ws{nnls_lds + lane};
    const unsigned long long t_start = a.deadline_ticks ? wall_clock64() : 0ull;
    AbortCtl ctl;
    ctl.first_success = a.first_success;
    ctl.deadline = a.deadline_ticks ? t_start + a.deadline_ticks : 0ull;

    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int t = tile / a.tiles_per_target;
        const int chunk = tile - t * a.tiles_per_target;
        const unsigned long long local = (unsigned long long)chunk * WAVE + (unsigned long long)lane;
        const bool active = local < a.n_restarts;
        const unsigned long long index = a.restart_begin + local;
        const Pose target = load_pose(a.targets + (size_t)t * 7);
        const double *x0p = a.x0 + (size_t)t * N;

        // lib.rs:366-370: restart 0 starts from the caller's seed
        double x[N];
        restart_seed<N>(a.key, sch.lb, a.scale, index, x);
        if (index == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = x0p[i];
        }

        RestartOut<N> r;
        solve_restart<N, TIP>(sch, a.ep, a.sp, target, ws, active, x, index, ctl, (unsigned)t, r);

        const size_t col = (size_t)t * a.n_restarts + local;
        const size_t ld = (size_t)a.n_restarts * (size_t)(a.n_tiles / a.tiles_per_target);
        if (active) {
            if (a.out_x) {
#pragma unroll
                for (int i = 0; i < N; ++i) a.out_x[(size_t)i * ld + col] = r.x[i];
            }
            if (a.out_f) a.out_f[col] = r.f;
            if (a.out_status) a.out_status[col] = r.result;
            if (a.out_evals) a.out_evals[col] = r.n_evals;
        }
        // selection key (lib.rs:402-407): Quality = ||x - x0||_2, Speed = index
        double key = 0.0;
        unsigned long long idx = ~0ull;
        if (active && r.success) {
            idx = index;
            if (a.quality) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) { const double d = r.x[i] - x0p[i]; acc += d * d; }
                key = __builtin_sqrt(acc);
            } else {
                key = (double)index;
            }
            if (a.first_success && !a.quality) atomicMin(a.first_success + t, index);
        }
        wave_argmin(key, idx);
        if (lane == 0) { a.tile_recs[tile].idx = idx; a.tile_recs[tile].key = key; }
    }
}

struct SelectLaunch {
    const TileRec *tile_recs;
    int tiles_per_target;
    int n;
    unsigned long long restart_begin;
    unsigned long long n_restarts;
    size_t ld;             // T * R
    const double *out_x;   // per-restart [n][ld] (may be null)
    const double *out_f;
    double *win_x;         // [T][n]
    double *win_f;
    unsigned long long *win_idx;
    double *win_key;
};

// One 64-lane block per target: argmin over the target's tile records.
__global__ __launch_bounds__(WAVE) void ik_select_kernel(const SelectLaunch a) {
    const int t = blockIdx.x;
    double key = 0.0;
    unsigned long long idx = ~0ull;
    for (int i = threadIdx.x; i < a.tiles_per_target; i += WAVE) {
        const TileRec r = a.tile_recs[(size_t)t * a.tiles_per_target + i];
        const bool take = (r.idx != ~0ull) && (idx == ~0ull || r.key < key || (r.key == key && r.idx < idx));
        if (take) { key = r.key; idx = r.idx; }
    }
    wave_argmin(key, idx);
    if (threadIdx.x == 0) {
        if (a.win_idx) a.win_idx[t] = idx;
        if (a.win_key) a.win_key[t] = key;
        const bool found = idx != ~0ull;
        const size_t col = (size_t)t * a.n_restarts + (found ? (size_t)(idx - a.restart_begin) : 0);
        if (a.win_f) a.win_f[t] = (found && a.out_f) ? a.out_f[col] : __builtin_nan("");
        if (a.win_x) {
            for (int i = 0; i < a.n; ++i)
                a.win_x[(size_t)t * a.n + i] =
                    (found && a.out_x) ? a.out_x[(size_t)i * a.ld + col] : __builtin_nan("");
        }
    }
}

struct EvalLaunch {
    const ChainDev *chain;
    EvalParams ep;
    double target[7];
    const double *q;  // [n][B]
    long long B;
    double *f;        // [B]
    double *g;        // [n][B] or null
};

template <int N, bool TIP>
__global__ __launch_bounds__(256) void eval_batch_kernel(const EvalLaunch a) {
    __shared__ ChainDev sch;
    stage_chain(sch, a.chain);
    const Pose target = load_pose(a.target);
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.B;
         b += (long long)gridDim.x * blockDim.x) {
        double q[N], g[N];
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = a.q[(size_t)i * a.B + b];
        const double f = eval_fg<N, TIP>(sch, a.ep, target, q, g);
        a.f[b] = f;
        if (a.g) {
#pragma unroll
            for (int i = 0; i < N; ++i) a.g[(size_t)i * a.B + b] = g[i];
        }
    }
}

struct FkLaunch {
    const ChainDev *chain;
    EvalParams ep;  // only the ee_offset part is used
    const double *q;
    long long B;
    double *pose;  // [7][B]
    double *jac;   // [6n][B] or null
};

template <int N, bool TIP>
__global__ __launch_bounds__(256) void fk_batch_kernel(const FkLaunch a) {
    __shared__ ChainDev sch;
    stage_chain(sch, a.chain);
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.B;
         b += (long long)gridDim.x * blockDim.x) {
        double q[N];
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = a.q[(size_t)i * a.B + b];
        Kin<N, TIP> kin;
        forward_kinematics<N, TIP>(sch, a.ep, q, kin);
        const double p[7] = {kin.ee.t.x, kin.ee.t.y, kin.ee.t.z, kin.ee.q.i, kin.ee.q.j, kin.ee.q.k, kin.ee.q.w};
#pragma unroll
        for (int i = 0; i < 7; ++i) a.pose[(size_t)i * a.B + b] = p[i];
        if (a.jac) {
            // joint_jacobian, kinematics.rs:166-196
            const Q4 eeqc = qconj(kin.ee.q);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const V3 ax{sch.axis[k][0], sch.axis[k][1], sch.axis[k][2]};
                const V3 angular = qrot(kin.tf[k].q, ax);
                const V3 d{kin.ee.t.x - kin.tf[k].t.x, kin.ee.t.y - kin.tf[k].t.y, kin.ee.t.z - kin.tf[k].t.z};
                const V3 linear = cross(angular, d);
                const V3 al = qrot(eeqc, angular);
                const V3 ll = qrot(eeqc, linear);
                const double c6[6] = {ll.x, ll.y, ll.z, al.x, al.y, al.z};
#pragma unroll
                for (int r = 0; r < 6; ++r) a.jac[(size_t)(k * 6 + r) * a.B + b] = c6[r];
            }
        }
    }
}

struct SeedLaunch {
    uint32_t key[8];
    double lb[MAX_DOF];
    double scale[MAX_DOF];
    unsigned long long first;
    long long count;
    double *q;  // [n][count]
};

template <int N>
__global__ __launch_bounds__(256) void seed_batch_kernel(const SeedLaunch a) {
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.count;
         b += (long long)gridDim.x * blockDim.x) {
        double q[N];
        restart_seed<N>(a.key, a.lb, a.scale, a.first + (unsigned long long)b, q);
#pragma unroll
        for (int i = 0; i < N; ++i) a.q[(size_t)i * a.count + b] = q[i];
    }
}

__global__ void probe_kernel(int op, const double *a, const double *b, long long count, double *out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (long long)gridDim.x * blockDim.x) {
        double r = 0.0, s, c;
        switch (op) {
        case 0: r = a[i] / b[i]; break;
        case 1: r = __builtin_sqrt(a[i]); break;
        case 2: sincos_dev(a[i], s, c); r = s; break;
        case 3: sincos_dev(a[i], s, c); r = c; break;
        default: r = atan2_q1(a[i], b[i]); break;
        }
        out[i] = r;
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// host side: C ABI
// ---------------------------------------------------------------------------

struct optik_hip_chain {
    ChainDev host;
    ChainDev *dev = nullptr;
    int n = 0;
    bool tip = false;
    uint32_t key[8];
    double scale[MAX_DOF];
    // launch workspace (grown on demand; one in-flight ik call per chain handle)
    std::mutex mu;
    std::mutex host_mu;  // serialises optik_hip_ik_host calls (they share the workspace below)
    TileRec *tile_recs = nullptr;
    size_t tile_cap = 0;
    unsigned long long *first_success = nullptr;
    size_t fs_cap = 0;
    // scratch per-restart buffers when the caller does not provide them
    double *tmp_x = nullptr, *tmp_f = nullptr;
    size_t tmp_cols = 0;
    // timing
    int timing = 0;
    static constexpr int EV_POOL = 256;  // event pairs recorded round-robin around the solve kernel
    hipEvent_t ev0[EV_POOL] = {}, ev1[EV_POOL] = {};
    int ev_count = 0;                    // launches recorded since the last reset
    optik_hip_launch_info last{};
    int num_cus = 0;
    int wall_clock_khz = 0;
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(OPTIK_HIP_ENODEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// rand_core 0.9 SeedableRng::seed_from_u64 (PCG32 expansion).
void seed_from_u64(uint64_t state, uint32_t key[8]) {
    const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
    for (int i = 0; i < 8; ++i) {
        state = state * MUL + INC;
        const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
}

// rand 0.9 UniformFloat<f64>::new_inclusive: scale with the 1-ulp decrease loop.
double uniform_scale(double low, double high) {
    const double max_rand = 1.0 - 2.220446049250313e-16;
    double scale = (high - low) / max_rand;
    while (scale * max_rand + low > high) {
        uint64_t u;
        std::memcpy(&u, &scale, 8);
        u -= 1;
        std::memcpy(&scale, &u, 8);
    }
    return scale;
}

// approx::relative_eq!(a, b, epsilon = eps), default max_relative = f64::EPSILON.
bool relative_eq(double a, double b, double eps) {
    if (a == b) return true;
    if (std::isinf(a) || std::isinf(b)) return false;
    const double d = std::fabs(a - b);
    if (d <= eps) return true;
    const double largest = std::fmax(std::fabs(a), std::fabs(b));
    return d <= largest * 2.220446049250313e-16;
}

// nalgebra is_identity on a 3-vector (objective.rs:13,25; quirk Q2).
bool vec3_is_identity(const double w[3]) {
    const double eps = 1e-20;
    return relative_eq(w[0], 1.0, eps) && relative_eq(w[1], 0.0, eps) && relative_eq(w[2], 0.0, eps);
}

void make_eval_params(const double wl[3], const double wa[3], const double *ee_offset7, EvalParams &ep) {
    std::memset(&ep, 0, sizeof ep);
    for (int i = 0; i < 3; ++i) {
        ep.w_lin[i] = wl[i];
        ep.w_ang[i] = wa[i];
        ep.w_lin2[i] = wl[i] * wl[i];  // objective.rs:102-103
        ep.w_ang2[i] = wa[i] * wa[i];
    }
    ep.skip_lin = vec3_is_identity(ep.w_lin);
    ep.skip_ang = vec3_is_identity(ep.w_ang);
    ep.skip_lin2 = vec3_is_identity(ep.w_lin2);
    ep.skip_ang2 = vec3_is_identity(ep.w_ang2);
    ep.grad_same_as_value = std::memcmp(ep.w_lin, ep.w_lin2, sizeof ep.w_lin) == 0
                            && std::memcmp(ep.w_ang, ep.w_ang2, sizeof ep.w_ang) == 0;
    const double ident[7] = {0, 0, 0, 0, 0, 0, 1};
    ep.has_ee_offset = ee_offset7 && std::memcmp(ee_offset7, ident, sizeof ident) != 0;
    std::memcpy(ep.ee_offset, ee_offset7 ? ee_offset7 : ident, sizeof ident);
}

int ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(OPTIK_HIP_ENODEVICE, std::string("no HIP device available: ")
                                             + (e == hipSuccess ? "device count is 0" : hipGetErrorString(e)));
    return 0;
}

// Dispatch on (n, trailing fixed joint).
#define OPTIK_DISPATCH(CH, CALL)                                                       \
    do {                                                                               \
        const int n_ = (CH)->n;                                                        \
        const bool tip_ = (CH)->tip;                                                   \
        if (n_ == 6 && !tip_) { CALL(6, false); }                                      \
        else if (n_ == 6 && tip_) { CALL(6, true); }                                   \
        else if (n_ == 7 && !tip_) { CALL(7, false); }                                 \
        else if (n_ == 7 && tip_) { CALL(7, true); }                                   \
        else return fail(OPTIK_HIP_EUNSUPPORTED, "kernels are built for n in {6,7}");  \
    } while (0)

int grid_for(const optik_hip_chain *ch, long long work, int block, int per_cu) {
    long long blocks = (work + block - 1) / block;
    const long long cap = (long long)(ch->num_cus > 0 ? ch->num_cus : 256) * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

extern "C" {

int optik_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *optik_hip_last_error(void) { return g_err.c_str(); }

int optik_hip_chain_create(const double *origins, const double *axes, const int32_t *types,
                           int32_t n_joints, const double *lb, const double *ub, int32_t n,
                           optik_hip_chain **out) {
    if (!origins || !axes || !types || !lb || !ub || !out) return fail(OPTIK_HIP_EINVAL, "null argument");
    if (n < 1 || n > MAX_DOF) return fail(OPTIK_HIP_EUNSUPPORTED, "num_positions must be in 1..8");
    if (n_joints != n && n_joints != n + 1)
        return fail(OPTIK_HIP_EUNSUPPORTED, "chain must be n revolute joints plus an optional trailing fixed joint");
    for (int j = 0; j < n; ++j)
        if (types[j] != OPTIK_JOINT_REVOLUTE)
            return fail(OPTIK_HIP_EUNSUPPORTED,
                        "only revolute joints are supported (the reference's Jacobian panics on prismatic, kinematics.rs:185)");
    if (n_joints == n + 1 && types[n] != OPTIK_JOINT_FIXED)
        return fail(OPTIK_HIP_EUNSUPPORTED, "joint after the last revolute joint must be fixed");
    if (int rc = ensure_device()) return rc;

    auto *ch = new optik_hip_chain();
    std::memset(&ch->host, 0, sizeof ch->host);
    ch->n = n;
    ch->tip = (n_joints == n + 1);
    ch->host.n_pos = n;
    ch->host.has_tip = ch->tip;
    for (int j = 0; j < n_joints; ++j)
        for (int k = 0; k < 7; ++k) ch->host.origin[j][k] = origins[j * 7 + k];
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < 3; ++k) ch->host.axis[j][k] = axes[j * 3 + k];
    for (int k = 0; k < n; ++k) {
        ch->host.lb[k] = lb[k];
        ch->host.ub[k] = ub[k];
        // infinite limits (continuous joints) make random_range panic in the
        // reference (quirk Q5); restarts > 0 are refused at launch time instead.
        ch->scale[k] = (std::isfinite(lb[k]) && std::isfinite(ub[k])) ? uniform_scale(lb[k], ub[k]) : NAN;
    }
    seed_from_u64(42, ch->key);  // RNG_SEED, lib.rs:360
    hipError_t e = hipMalloc(&ch->dev, sizeof(ChainDev));
    if (e == hipSuccess) e = hipMemcpy(ch->dev, &ch->host, sizeof(ChainDev), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        delete ch;
        return fail(OPTIK_HIP_ENODEVICE, std::string("chain upload: ") + hipGetErrorString(e));
    }
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&ch->num_cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&ch->wall_clock_khz, hipDeviceAttributeWallClockRate, dev);
    *out = ch;
    return 0;
}

void optik_hip_chain_destroy(optik_hip_chain *ch) {
    if (!ch) return;
    if (ch->dev) hipFree(ch->dev);
    if (ch->tile_recs) hipFree(ch->tile_recs);
    if (ch->first_success) hipFree(ch->first_success);
    if (ch->tmp_x) hipFree(ch->tmp_x);
    if (ch->tmp_f) hipFree(ch->tmp_f);
    for (int i = 0; i < optik_hip_chain::EV_POOL; ++i) {
        if (ch->ev0[i]) hipEventDestroy(ch->ev0[i]);
        if (ch->ev1[i]) hipEventDestroy(ch->ev1[i]);
    }
    delete ch;
}

int32_t optik_hip_chain_num_positions(const optik_hip_chain *ch) { return ch ? ch->n : 0; }

int optik_hip_eval_batch(const optik_hip_chain *ch, const optik_solver_config *cfg, const double *target7,
                         const double *ee_offset7, const double *d_q, int64_t B, double *d_f, double *d_g,
                         void *stream) {
    if (!ch || !cfg || !target7 || !d_q || !d_f || B < 0) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (B == 0) return 0;
    EvalLaunch a;
    a.chain = ch->dev;
    make_eval_params(cfg->linear_weight, cfg->angular_weight, ee_offset7, a.ep);
    std::memcpy(a.target, target7, sizeof a.target);
    a.q = d_q; a.B = B; a.f = d_f; a.g = d_g;
    const int grid = grid_for(ch, B, 256, 8);
#define CALL(NN, TT) hipLaunchKernelGGL((eval_batch_kernel<NN, TT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a)
    OPTIK_DISPATCH(ch, CALL);
#undef CALL
    HIP_TRY(hipGetLastError());
    return 0;
}

int optik_hip_fk_batch(const optik_hip_chain *ch, const double *ee_offset7, const double *d_q, int64_t B,
                       double *d_pose, double *d_jac, void *stream) {
    if (!ch || !d_q || !d_pose || B < 0) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (B == 0) return 0;
    FkLaunch a;
    a.chain = ch->dev;
    const double one[3] = {1, 1, 1};
    make_eval_params(one, one, ee_offset7, a.ep);
    a.q = d_q; a.B = B; a.pose = d_pose; a.jac = d_jac;
    const int grid = grid_for(ch, B, 256, 8);
#define CALL(NN, TT) hipLaunchKernelGGL((fk_batch_kernel<NN, TT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a)
    OPTIK_DISPATCH(ch, CALL);
#undef CALL
    HIP_TRY(hipGetLastError());
    return 0;
}

int optik_hip_seed_batch(const optik_hip_chain *ch, uint64_t first, int64_t count, double *d_q, void *stream) {
    if (!ch || !d_q || count < 0) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (count == 0) return 0;
    for (int k = 0; k < ch->n; ++k)
        if (std::isnan(ch->scale[k]))
            return fail(OPTIK_HIP_EINVAL, "random restarts need finite joint limits (reference: random_range panics)");
    SeedLaunch a;
    std::memcpy(a.key, ch->key, sizeof a.key);
    std::memcpy(a.lb, ch->host.lb, sizeof a.lb);
    std::memcpy(a.scale, ch->scale, sizeof a.scale);
    a.first = first; a.count = count; a.q = d_q;
    const int grid = grid_for(ch, count, 256, 8);
#define CALL(NN, TT) hipLaunchKernelGGL((seed_batch_kernel<NN>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a)
    OPTIK_DISPATCH(ch, CALL);
#undef CALL
    HIP_TRY(hipGetLastError());
    return 0;
}

int optik_hip_ik_batch(optik_hip_chain *ch, const optik_solver_config *cfg, const double *d_targets,
                       const double *d_x0, int32_t T, const double *ee_offset7, uint64_t restart_begin,
                       uint64_t restart_end, uint32_t flags, double deadline_s, const optik_hip_ik_outputs *out,
                       void *stream_v) {
    if (!ch || !cfg || !d_targets || !d_x0 || !out || T < 1) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (restart_end <= restart_begin) return fail(OPTIK_HIP_EINVAL, "empty restart range");
    if (cfg->solution_mode != 1 && cfg->solution_mode != 2)
        return fail(OPTIK_HIP_EINVAL, "solution_mode must be 1 (Quality) or 2 (Speed)");
    const uint64_t R = restart_end - restart_begin;
    if (restart_end > 1 || restart_begin > 0)
        for (int k = 0; k < ch->n; ++k)
            if (std::isnan(ch->scale[k]))
                return fail(OPTIK_HIP_EINVAL, "random restarts need finite joint limits (reference: random_range panics)");
    hipStream_t stream = (hipStream_t)stream_v;
    std::lock_guard<std::mutex> lock(ch->mu);

    const uint64_t tiles_per_target = (R + WAVE - 1) / WAVE;
    const uint64_t n_tiles64 = tiles_per_target * (uint64_t)T;
    if (n_tiles64 > 0x7fffffffull) return fail(OPTIK_HIP_EINVAL, "too many restarts in one launch");
    const int n_tiles = (int)n_tiles64;
    const size_t cols = (size_t)T * (size_t)R;

    if ((size_t)n_tiles > ch->tile_cap) {
        if (ch->tile_recs) HIP_TRY(hipFree(ch->tile_recs));
        ch->tile_recs = nullptr;
        HIP_TRY(hipMalloc(&ch->tile_recs, sizeof(TileRec) * (size_t)n_tiles));
        ch->tile_cap = (size_t)n_tiles;
    }
    const bool early = (flags & OPTIK_HIP_IK_EARLY_EXIT) && cfg->solution_mode == 2;
    if (early) {
        if ((size_t)T > ch->fs_cap) {
            if (ch->first_success) HIP_TRY(hipFree(ch->first_success));
            ch->first_success = nullptr;
            HIP_TRY(hipMalloc(&ch->first_success, sizeof(unsigned long long) * (size_t)T));
            ch->fs_cap = (size_t)T;
        }
        HIP_TRY(hipMemsetAsync(ch->first_success, 0xff, sizeof(unsigned long long) * (size_t)T, stream));
    }
    // the selection needs the per-restart x / f: use scratch if the caller skips them
    double *px = out->d_x, *pf = out->d_f;
    const bool want_win = out->d_win_x || out->d_win_f;
    if (want_win && (!px || !pf)) {
        if (cols > ch->tmp_cols) {
            if (ch->tmp_x) HIP_TRY(hipFree(ch->tmp_x));
            if (ch->tmp_f) HIP_TRY(hipFree(ch->tmp_f));
            ch->tmp_x = ch->tmp_f = nullptr;
            HIP_TRY(hipMalloc(&ch->tmp_x, sizeof(double) * cols * (size_t)ch->n));
            HIP_TRY(hipMalloc(&ch->tmp_f, sizeof(double) * cols));
            ch->tmp_cols = cols;
        }
        if (!px) px = ch->tmp_x;
        if (!pf) pf = ch->tmp_f;
    }

    SolveLaunch a;
    std::memset(&a, 0, sizeof a);
    a.chain = ch->dev;
    make_eval_params(cfg->linear_weight, cfg->angular_weight, ee_offset7, a.ep);
    a.sp.stopval = cfg->tol_f;
    a.sp.ftol_abs = (cfg->tol_df > 0.0) ? cfg->tol_df : 1e-3 * cfg->tol_f;  // lib.rs:283-293
    a.sp.xtol_abs = cfg->tol_dx;
    a.sp.ok_stopval = cfg->tol_f >= 0.0;
    a.sp.ok_ftol = cfg->tol_df >= 0.0;
    a.sp.ok_xtol = cfg->tol_dx >= 0.0;
    std::memcpy(a.key, ch->key, sizeof a.key);
    std::memcpy(a.scale, ch->scale, sizeof a.scale);
    a.targets = d_targets;
    a.x0 = d_x0;
    a.restart_begin = restart_begin;
    a.n_restarts = R;
    a.tiles_per_target = (int)tiles_per_target;
    a.n_tiles = n_tiles;
    a.quality = (cfg->solution_mode == 1);
    a.out_x = px;
    a.out_f = pf;
    a.out_status = out->d_status;
    a.out_evals = out->d_evals;
    a.tile_recs = ch->tile_recs;
    a.first_success = early ? ch->first_success : nullptr;
    a.deadline_ticks = 0;
    if (deadline_s > 0.0) {
        const double khz = ch->wall_clock_khz > 0 ? (double)ch->wall_clock_khz : 100000.0;
        a.deadline_ticks = (unsigned long long)(deadline_s * khz * 1e3);
        if (a.deadline_ticks == 0) a.deadline_ticks = 1;
    }

    // One 64-lane workgroup per tile; 2 workgroups fit a CU (LDS-bound), so cap
    // the grid at a few waves per CU slot and let workgroups stride over tiles.
    const int cus = ch->num_cus > 0 ? ch->num_cus : 256;
    int grid = n_tiles;
    const int cap = cus * 2 * 4;
    if (grid > cap) grid = cap;

    const int ev_slot = ch->ev_count % optik_hip_chain::EV_POOL;
    if (ch->timing) {
        if (!ch->ev0[ev_slot]) { HIP_TRY(hipEventCreate(&ch->ev0[ev_slot])); HIP_TRY(hipEventCreate(&ch->ev1[ev_slot])); }
        HIP_TRY(hipEventRecord(ch->ev0[ev_slot], stream));
    }
    int lds = 0;
#define CALL(NN, TT)                                                                                 \
    lds = (int)(sizeof(ChainDev) + sizeof(double) * NnlsLayout<NN>::SLOTS * WAVE);                   \
    hipLaunchKernelGGL((ik_solve_kernel<NN, TT>), dim3(grid), dim3(WAVE), 0, stream, a)
    OPTIK_DISPATCH(ch, CALL);
#undef CALL
    HIP_TRY(hipGetLastError());
    if (ch->timing) { HIP_TRY(hipEventRecord(ch->ev1[ev_slot], stream)); ch->ev_count += 1; }
    ch->last.grid = grid; ch->last.block = WAVE; ch->last.lds_bytes = lds; ch->last.tiles = n_tiles;

    if (out->d_win_x || out->d_win_f || out->d_win_idx || out->d_win_key) {
        SelectLaunch s;
        s.tile_recs = ch->tile_recs;
        s.tiles_per_target = (int)tiles_per_target;
        s.n = ch->n;
        s.restart_begin = restart_begin;
        s.n_restarts = R;
        s.ld = cols;
        s.out_x = px; s.out_f = pf;
        s.win_x = out->d_win_x; s.win_f = out->d_win_f;
        s.win_idx = (unsigned long long *)out->d_win_idx; s.win_key = out->d_win_key;
        hipLaunchKernelGGL(ik_select_kernel, dim3(T), dim3(WAVE), 0, stream, s);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

int optik_hip_ik_host(optik_hip_chain *ch, const optik_solver_config *cfg, const double *targets,
                      const double *x0, int32_t T, const double *ee_offset7, uint64_t restart_begin,
                      uint64_t restart_end, uint32_t flags, double deadline_s, double *win_x, double *win_f,
                      uint64_t *win_idx, double *win_key) {
    if (!ch || !targets || !x0 || T < 1) return fail(OPTIK_HIP_EINVAL, "bad argument");
    // the launch workspace of the chain is in use until the copies below are done
    std::lock_guard<std::mutex> host_lock(ch->host_mu);
    const int n = ch->n;
    double *d_t = nullptr, *d_x0 = nullptr, *d_wx = nullptr, *d_wf = nullptr, *d_wk = nullptr;
    uint64_t *d_wi = nullptr;
    int rc = 0;
    auto cleanup = [&]() {
        if (d_t) hipFree(d_t);
        if (d_x0) hipFree(d_x0);
        if (d_wx) hipFree(d_wx);
        if (d_wf) hipFree(d_wf);
        if (d_wk) hipFree(d_wk);
        if (d_wi) hipFree(d_wi);
    };
#define TRY_CLEAN(expr)                                                                           \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            cleanup();                                                                            \
            return fail(OPTIK_HIP_ENODEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                         \
    } while (0)
    TRY_CLEAN(hipMalloc(&d_t, sizeof(double) * 7 * (size_t)T));
    TRY_CLEAN(hipMalloc(&d_x0, sizeof(double) * (size_t)n * (size_t)T));
    TRY_CLEAN(hipMalloc(&d_wx, sizeof(double) * (size_t)n * (size_t)T));
    TRY_CLEAN(hipMalloc(&d_wf, sizeof(double) * (size_t)T));
    TRY_CLEAN(hipMalloc(&d_wk, sizeof(double) * (size_t)T));
    TRY_CLEAN(hipMalloc(&d_wi, sizeof(uint64_t) * (size_t)T));
    TRY_CLEAN(hipMemcpy(d_t, targets, sizeof(double) * 7 * (size_t)T, hipMemcpyHostToDevice));
    TRY_CLEAN(hipMemcpy(d_x0, x0, sizeof(double) * (size_t)n * (size_t)T, hipMemcpyHostToDevice));
    optik_hip_ik_outputs o;
    std::memset(&o, 0, sizeof o);
    o.d_win_x = d_wx; o.d_win_f = d_wf; o.d_win_idx = d_wi; o.d_win_key = d_wk;
    rc = optik_hip_ik_batch(ch, cfg, d_t, d_x0, T, ee_offset7, restart_begin, restart_end, flags, deadline_s,
                            &o, nullptr);
    if (rc) { cleanup(); return rc; }
    TRY_CLEAN(hipDeviceSynchronize());
    if (win_x) TRY_CLEAN(hipMemcpy(win_x, d_wx, sizeof(double) * (size_t)n * (size_t)T, hipMemcpyDeviceToHost));
    if (win_f) TRY_CLEAN(hipMemcpy(win_f, d_wf, sizeof(double) * (size_t)T, hipMemcpyDeviceToHost));
    if (win_idx) TRY_CLEAN(hipMemcpy(win_idx, d_wi, sizeof(uint64_t) * (size_t)T, hipMemcpyDeviceToHost));
    if (win_key) TRY_CLEAN(hipMemcpy(win_key, d_wk, sizeof(double) * (size_t)T, hipMemcpyDeviceToHost));
#undef TRY_CLEAN
    cleanup();
    return 0;
}

int optik_hip_probe(int32_t op, const double *a, const double *b, int64_t count, double *out) {
    if (!a || !out || count < 0) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (int rc = ensure_device()) return rc;
    if (count == 0) return 0;
    double *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
    const size_t bytes = sizeof(double) * (size_t)count;
    HIP_TRY(hipMalloc(&d_a, bytes));
    HIP_TRY(hipMalloc(&d_b, bytes));
    HIP_TRY(hipMalloc(&d_o, bytes));
    HIP_TRY(hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_b, b ? b : a, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_kernel, dim3(1024), dim3(256), 0, nullptr, op, d_a, d_b, (long long)count, d_o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost));
    hipFree(d_a); hipFree(d_b); hipFree(d_o);
    return 0;
}

void optik_hip_set_timing(optik_hip_chain *ch, int32_t enabled) {
    if (!ch) return;
    std::lock_guard<std::mutex> lock(ch->mu);
    ch->timing = enabled;
    ch->ev_count = 0;
}

int optik_hip_timing_mean(optik_hip_chain *ch, double *mean_ms, int32_t *count) {
    if (!ch || !mean_ms || !count) return fail(OPTIK_HIP_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lock(ch->mu);
    const int n = ch->ev_count < optik_hip_chain::EV_POOL ? ch->ev_count : optik_hip_chain::EV_POOL;
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        HIP_TRY(hipEventSynchronize(ch->ev1[i]));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ch->ev0[i], ch->ev1[i]));
        sum += ms;
    }
    *mean_ms = n ? sum / n : 0.0;
    *count = n;
    return 0;
}

int optik_hip_last_launch(const optik_hip_chain *ch, optik_hip_launch_info *info) {
    if (!ch || !info) return fail(OPTIK_HIP_EINVAL, "bad argument");
    *info = ch->last;
    info->kernel_ms = 0.0f;
    if (ch->timing && ch->ev_count > 0) {
        const int slot = (ch->ev_count - 1) % optik_hip_chain::EV_POOL;
        HIP_TRY(hipEventSynchronize(ch->ev1[slot]));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ch->ev0[slot], ch->ev1[slot]));
        info->kernel_ms = ms;
    }
    return 0;
}

}  // extern "C"
