// ik_kernels.hip -- the C ABI of include/optik_hip.h and the kernels that are not solvers.
//
//   single-launch solvers   launched from here, defined in their own translation units: the lane-per-restart form
//                           (ik_lane_kernel.hip), the quad solver (ik_quad_kernel.hip), the general run-time-n solver
//                           (ik_wide_kernel.hip); optik_hip_ik_batch picks one by launch size and joint count
//   eng_*_kernel            the streaming engine's phase kernels (ik_engine.hpp, ik_nnls_coop.hpp)
//   ik_tile_argmin_kernel, ik_select_kernel, ik_select_small_kernel
//                           the selection of lib.rs:397-413 over the per-restart keys
//   eval_batch_kernel       objective + gradient for a batch of configurations
//   fk_batch_kernel         end-effector pose (+ body Jacobian) for a batch
//   seed_batch_kernel       ChaCha8 restart seeds
//   probe_kernel            elementary functions (test hook)
// All f64.  No CPU fallback exists: every entry point fails loudly without a device.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/optik_hip.h"
#include "device_scope.hpp"
#include "ik_host_params.hpp"
#include "ik_launch.hpp"
#include "ik_wide_launch.hpp"
#include "ik_engine.hpp"

using namespace optik;
using namespace optik::hostparams;

// ---------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------

namespace {

constexpr int WAVE = 64;

struct TileRec {
    unsigned long long idx;  // winning restart index in the tile, ~0 if none
    double key;
};

// (key, idx) argmin across the wave: smaller key wins, ties -> smaller idx; idx ~0 = none.
__device__ __forceinline__ void wave_argmin(double &key, unsigned long long &idx) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double okey = __shfl_xor(key, off, WAVE);
        const unsigned long long oidx = __shfl_xor(idx, off, WAVE);
        const bool take = (oidx != ~0ull) && (idx == ~0ull || okey < key || (okey == key && oidx < idx));
        if (take) { key = okey; idx = oidx; }
    }
}

constexpr int QUADS_PER_WAVE_HOST = 16;  // restarts a wave of the quad solver holds


struct SelectLaunch {
    const double *out_key;   // [T*R] selection key, +inf unless the restart succeeded
    const double *out_x;     // [n][T*R]
    const double *out_f;     // [T*R]
    TileRec *tile_recs;      // [T][tiles_per_target]
    int tiles_per_target;
    int tile;                // restarts per tile
    int n;
    int pad;
    unsigned long long restart_begin;
    unsigned long long n_restarts;
    size_t ld;               // T * R
    double *win_x;           // [T][n]
    double *win_f;
    unsigned long long *win_idx;
    double *win_key;
    // the launch's work-item counter and first-success words, put back to their initial values by the last
    // kernel of the launch so that the next launch needs no fill commands in front of it (null: leave them)
    unsigned long long *reset_queue;
    unsigned long long *reset_fs;  // [T]
};

// Stage 1 of the selection (lib.rs:397-413): per-block argmin of the keys of one
// tile of one target -- wavefront shuffles, then one 16-byte record per block.
__global__ __launch_bounds__(256) void ik_tile_argmin_kernel(const SelectLaunch a) {
    __shared__ double s_key[4];
    __shared__ unsigned long long s_idx[4];
    // (one-dimensional grid over target-major tiles: grid.y would cap T at 65 535)
    const int t = (int)(blockIdx.x / (unsigned)a.tiles_per_target);
    const int tile = (int)(blockIdx.x % (unsigned)a.tiles_per_target);
    const unsigned long long lo = (unsigned long long)tile * (unsigned long long)a.tile;
    unsigned long long hi = lo + (unsigned long long)a.tile;
    if (hi > a.n_restarts) hi = a.n_restarts;
    double key = 0.0;
    unsigned long long idx = ~0ull;
    for (unsigned long long r = lo + threadIdx.x; r < hi; r += blockDim.x) {
        const double k = a.out_key[(size_t)t * a.n_restarts + r];
        const unsigned long long i = a.restart_begin + r;
        const bool ok = k < __builtin_huge_val();
        if (ok && (idx == ~0ull || k < key || (k == key && i < idx))) { key = k; idx = i; }
    }
    wave_argmin(key, idx);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_key[wave] = key; s_idx[wave] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            const bool take = (s_idx[w] != ~0ull)
                              && (idx == ~0ull || s_key[w] < key || (s_key[w] == key && s_idx[w] < idx));
            if (take) { key = s_key[w]; idx = s_idx[w]; }
        }
        TileRec rec;
        rec.idx = idx;
        rec.key = key;
        a.tile_recs[(size_t)t * a.tiles_per_target + tile] = rec;
    }
}

// The winner of target t goes out (one thread), and the launch's queue / first-success words go back to their
// initial values for the next launch.
__device__ void select_publish(const SelectLaunch &a, int t, double key, unsigned long long idx) {
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
    if (a.reset_fs) a.reset_fs[t] = ~0ull;
    if (a.reset_queue && t == 0) *a.reset_queue = 0ull;
}

// Stage 2: one 64-lane block per target reduces the tile records and gathers the winner.
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
    if (threadIdx.x == 0) select_publish(a, t, key, idx);
}

// Both stages in one kernel for a launch of at most one tile of restarts per target (a single ik() call's first
// launches, a Speed batch's rounds): one 256-thread block per target.
__global__ __launch_bounds__(256) void ik_select_small_kernel(const SelectLaunch a) {
    __shared__ double s_key[4];
    __shared__ unsigned long long s_idx[4];
    const int t = blockIdx.x;
    double key = 0.0;
    unsigned long long idx = ~0ull;
    for (unsigned long long r = threadIdx.x; r < a.n_restarts; r += blockDim.x) {
        const double k = a.out_key[(size_t)t * a.n_restarts + r];
        const unsigned long long i = a.restart_begin + r;
        const bool ok = k < __builtin_huge_val();
        if (ok && (idx == ~0ull || k < key || (k == key && i < idx))) { key = k; idx = i; }
    }
    wave_argmin(key, idx);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_key[wave] = key; s_idx[wave] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            const bool take = (s_idx[w] != ~0ull)
                              && (idx == ~0ull || s_key[w] < key || (s_key[w] == key && s_idx[w] < idx));
            if (take) { key = s_key[w]; idx = s_idx[w]; }
        }
        select_publish(a, t, key, idx);
    }
}

// ---- streaming engine kernels (ik_engine.hpp) ---------------------------------
// minimum waves per SIMD the register allocator must leave room for (tuned on MI355X)
#ifndef OPTIK_ENG_EVAL_WAVES
#define OPTIK_ENG_EVAL_WAVES 2
#endif
#ifndef OPTIK_ENG_UPD_WAVES
#define OPTIK_ENG_UPD_WAVES 2
#endif
// threads per workgroup of the per-slot kernels (eval / update / finish)
#ifndef OPTIK_ENG_SLOT_BLOCK
#define OPTIK_ENG_SLOT_BLOCK 128
#endif

// ... of the update and finish kernels: one wave.  A workgroup keeps its LDS (the update
// kernel's record window: 20 KB per wave) and its CU slots until its slowest wave is done;
// with single-wave workgroups every wave gives them back as soon as it ends (measured
// +7 % restarts/s against 256-thread workgroups).
#ifndef OPTIK_ENG_UPD_BLOCK
#define OPTIK_ENG_UPD_BLOCK 64
#endif

template <int N, bool TIP>
__global__ __launch_bounds__(OPTIK_ENG_SLOT_BLOCK, OPTIK_ENG_EVAL_WAVES) void eng_eval_kernel(const EngArgs a) {
    __shared__ ChainDev sch;
    extern __shared__ __attribute__((aligned(16))) unsigned char eval_dyn_lds[];  // n_jobs job records
    EngJob *sjobs = reinterpret_cast<EngJob *>(eval_dyn_lds);
    {
        static_assert(sizeof(EngJob) % sizeof(double) == 0, "EngJob is a whole number of doubles");
        const double *src = reinterpret_cast<const double *>(a.jobs);
        double *dst = reinterpret_cast<double *>(sjobs);
        const int nd = a.n_jobs * (int)(sizeof(EngJob) / sizeof(double));
        for (int i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = src[i];
    }
    stage_chain(sch, a.chain);
    const size_t local = (size_t)blockIdx.x * OPTIK_ENG_SLOT_BLOCK + threadIdx.x;
    const size_t slot = (size_t)a.slot_base + local;
    // first kernel of a trip: reset the other parity's list counters (consumed last trip; this
    // trip's suspended solves and finishing pass and the next trip's update append to it) and
    // the in-use count the update kernel accumulates -- no memset launches between trips
    if (blockIdx.x == 0) {
        if (threadIdx.x < (unsigned)NN_CLASSES) a.nn_class_count[a.parity ^ 1][threadIdx.x] = 0u;
        if (threadIdx.x == (unsigned)NN_CLASSES) *a.n_active = 0u;
        if (threadIdx.x == (unsigned)NN_CLASSES + 1u) *a.refill_count = 0u;
        if (a.cont_count && threadIdx.x >= 32u && threadIdx.x < 32u + (unsigned)NN_CONT_SHARDS) a.cont_count[threadIdx.x - 32u] = 0u;
    }
    const bool evaluated = local < a.n_slots && eng_eval_body<N, TIP>(a, sch, sjobs, slot);
    const unsigned long long em = __ballot(evaluated);
    if (a.exec_evals && em && (threadIdx.x & 63u) == 0)
        atomicAdd(a.exec_evals + (blockIdx.x % ENG_EXEC_SHARDS), (unsigned long long)__popcll(em));
}

template <int N>
__global__ __launch_bounds__(OPTIK_ENG_UPD_BLOCK, OPTIK_ENG_UPD_WAVES) void eng_update_kernel(const EngArgs a) {
    __shared__ ChainDev sch;
    stage_chain(sch, a.chain);
    const size_t local = (size_t)blockIdx.x * OPTIK_ENG_UPD_BLOCK + threadIdx.x;
    const size_t slot = (size_t)a.slot_base + local;
    __shared__ double rec_win[(OPTIK_ENG_UPD_BLOCK / 64) * RecIo<N>::WINDOW];
    eng_update_body<N>(a, sch, local < a.n_slots ? slot : (size_t)a.slot_base, local, local < a.n_slots,
                       rec_win + (threadIdx.x / 64u) * RecIo<N>::WINDOW);
}

// columns of a bounded sub-problem held per lane (16 / CPL lanes share a problem)
#ifndef OPTIK_ENG_CPL
#define OPTIK_ENG_CPL 4
#endif
#ifndef OPTIK_ENG_NNLS_WAVES
#define OPTIK_ENG_NNLS_WAVES 2
#endif
// One wave per workgroup: the waves of a workgroup hold their CU slots until the slowest
// of them (the largest iteration count among its problems) is done.
#ifndef OPTIK_ENG_NNLS_BLOCK
#define OPTIK_ENG_NNLS_BLOCK 64
#endif
__global__ __launch_bounds__(256) void eng_bucket_kernel(const EngArgs a) { eng_bucket_body(a); }

template <int N>
__global__ __launch_bounds__(OPTIK_ENG_NNLS_BLOCK, OPTIK_ENG_NNLS_WAVES) void eng_nnls_coop_kernel(const EngArgs a) {
    constexpr int WL = coop_wave_lds<OPTIK_ENG_CPL>();
    __shared__ __attribute__((aligned(16))) double win[(OPTIK_ENG_NNLS_BLOCK / 64) * WL];
    double *wave_lds = win + (threadIdx.x / 64u) * WL;
    if ((threadIdx.x & 63u) < 8) wave_lds[WL - 8 + (threadIdx.x & 63u)] = 0.0;
    eng_nnls_coop_body<N, OPTIK_ENG_CPL>(a, wave_lds);
}

template <int N>
__global__ __launch_bounds__(OPTIK_ENG_UPD_BLOCK, OPTIK_ENG_UPD_WAVES) void eng_finish_kernel(const EngArgs a) {
    __shared__ ChainDev sch;
    stage_chain(sch, a.chain);
    const size_t local = (size_t)blockIdx.x * OPTIK_ENG_UPD_BLOCK + threadIdx.x;
    const size_t slot = (size_t)a.slot_base + local;
    // the host reads the in-use count of a chunk from pinned memory (no copy kernel on the stream)
    if (a.host_in_use && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.host_in_use, *a.n_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    eng_finish_body<N>(a, sch, local < a.n_slots ? slot : (size_t)a.slot_base, local, local < a.n_slots);
}

__global__ __launch_bounds__(256) void eng_tail_list_kernel(const int32_t *state, unsigned long long n_slots,
                                                            unsigned int *count, unsigned int *list) {
    tail_list_body(state, n_slots, count, list);
}

__global__ __launch_bounds__(256) void eng_compact_scan_kernel(const CompactArgs c) { compact_scan_body(c); }
__global__ __launch_bounds__(256) void eng_compact_move_kernel(const CompactArgs c) { compact_move_body(c); }

__global__ void fill_f64_kernel(double *p, unsigned long long n, double v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// the run's deadline on the device clock (EngArgs::deadline_word)
__global__ void eng_deadline_kernel(unsigned long long *word, unsigned long long ticks) {
    *word = ticks ? (unsigned long long)wall_clock64() + ticks : 0ull;
}

__global__ void eng_init_kernel(int32_t *state, unsigned int *cls2, unsigned long long C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) { state[i] = ST_REFILL; cls2[i] = NN_NONE; cls2[C + i] = NN_NONE; }  // every slot wants a work item
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
        double q[N];
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = a.q[(size_t)i * a.B + b];
        // the gradient streams to its column as each component is known (as in eng_eval_kernel:
        // neither it nor the joint positions stay in registers across the evaluation)
        const double f = eval_fg_stream<N, TIP>(sch, a.ep, target, q, [&](int k, double v) {
            if (a.g) a.g[(size_t)k * a.B + b] = v;
        });
        a.f[b] = f;
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

// Chains with prismatic joints: the reference's FK handles them (kinematics.rs:243-255), its
// Jacobian -- and with it ik() -- does not (kinematics.rs:185: todo!()).  One generic kernel walks
// the joint table at run time: state = state * (origin_j * local_transform_j(q_j)).
struct FkGeneralLaunch {
    int32_t n_joints, n_pos;
    int32_t types[MAX_JOINTS];
    int32_t pad;
    double origin[MAX_JOINTS][7];
    double axis[MAX_JOINTS][3];
    double ee_offset[7];
    const double *q;  // [n][B]
    long long B;
    double *pose;     // [7][B]
};

__global__ __launch_bounds__(256) void fk_general_kernel(const FkGeneralLaunch a) {
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.B;
         b += (long long)gridDim.x * blockDim.x) {
        Pose state;
        state.t = V3{0.0, 0.0, 0.0};
        state.q = Q4{0.0, 0.0, 0.0, 1.0};
        int qi = 0;
        for (int j = 0; j < a.n_joints; ++j) {
            Pose local;
            local.t = V3{0.0, 0.0, 0.0};
            local.q = Q4{0.0, 0.0, 0.0, 1.0};
            if (a.types[j] == OPTIK_JOINT_REVOLUTE) {
                double s, c;
                sincos_dev(a.q[(size_t)qi * a.B + b] / 2.0, s, c);  // UnitQuaternion::from_axis_angle
                local.q = Q4{a.axis[j][0] * s, a.axis[j][1] * s, a.axis[j][2] * s, c};
                ++qi;
            } else if (a.types[j] == OPTIK_JOINT_PRISMATIC) {
                const double d = a.q[(size_t)qi * a.B + b];
                local.t = V3{a.axis[j][0] * d, a.axis[j][1] * d, a.axis[j][2] * d};
                ++qi;
            }
            const Pose jt = pose_mul(load_pose(a.origin[j]), local);
            state = pose_mul(state, jt);
        }
        const Pose ee = pose_mul(state, load_pose(a.ee_offset));
        const double p[7] = {ee.t.x, ee.t.y, ee.t.z, ee.q.i, ee.q.j, ee.q.k, ee.q.w};
#pragma unroll
        for (int i = 0; i < 7; ++i) a.pose[(size_t)i * a.B + b] = p[i];
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

// math.rs functions one at a time (test hook: compared with the reference's own golden vectors,
// /root/reference/crates/optik/tests/test_math.rs:14-61).  pose = t[3], quat[i,j,k,w]; matrices row-major.
// op 0 so3::log (3), 1 so3::right_jacobian(so3::log(q)) (9), 2 se3::log (6: V^-1 t, w),
// 3 se3::right_jacobian (36: [[J, Q], [0, J]]).
__global__ void probe_math_kernel(int op, const double *poses, long long count, double *out, int stride) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (long long)gridDim.x * blockDim.x) {
        const Pose X = load_pose(poses + i * 7);
        double *o = out + i * stride;
        const V3 w = so3_log(X.q);
        if (op == 0) { o[0] = w.x; o[1] = w.y; o[2] = w.z; continue; }
        const RotTerms rt = rot_terms(w);
        const M3 Jr = so3_right_jacobian(rt);
        if (op == 1) {
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[r * 3 + c] = Jr.m[r][c];
        } else if (op == 2) {
            const V3 lin = se3_log_linear(rt, X.t);
            o[0] = lin.x; o[1] = lin.y; o[2] = lin.z; o[3] = w.x; o[4] = w.y; o[5] = w.z;
        } else {
            const M3 Q = se3_q_matrix(rt, X.t, Jr);
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    o[r * 6 + c] = Jr.m[r][c];
                    o[r * 6 + 3 + c] = Q.m[r][c];
                    o[(r + 3) * 6 + c] = 0.0;
                    o[(r + 3) * 6 + 3 + c] = Jr.m[r][c];
                }
        }
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
    double scale[WIDE_MAX_DOF];
    int range_rule = 0;  // OPTIK_HIP_RANGE_*: how `scale` was formed
    // a chain with 9 .. 16 joint positions (ik_wide.hpp): its own table, the general kernels, no engine
    bool wide = false;
    WideChainDev whost;
    WideChainDev *wdev = nullptr;
    double *wide_ws = nullptr;  // restart workspace of the resident waves
    size_t wide_ws_waves = 0;
    int device_id = 0;   // the HIP device the chain lives on (the current device at creation)
    // a chain with prismatic joints: FK only (as in the reference); the joint table for fk_general_kernel
    bool prismatic = false;
    int n_joints = 0;
    int32_t types[MAX_JOINTS] = {};
    double axis_all[MAX_JOINTS][3] = {};
    // launch workspace (grown on demand; one in-flight ik call per chain handle)
    std::mutex mu;
    std::mutex eng_session_mu;  // optik_hip_engine_solve: submit + run of one job, not interleaved with another's
    std::mutex host_mu;  // serialises optik_hip_ik_host calls (they share the workspace below)
    TileRec *tile_recs = nullptr;
    size_t tile_cap = 0;
    unsigned long long *first_success = nullptr;
    size_t fs_cap = 0;
    // (what the last launch's selection kernel left behind: the work-item counter at 0, this many leading
    // first-success words at ~0 -- a launch that finds them so skips its fill commands)
    // (host-side knowledge that holds for launches ORDERED behind that selection kernel: the stream it ran on is kept
    // with it, a launch on any other stream fills the words itself)
    bool queue_clean = false;
    size_t fs_clean = 0;
    hipStream_t clean_stream = nullptr;
    // scratch per-restart buffers when the caller does not provide them
    double *tmp_x = nullptr, *tmp_f = nullptr, *tmp_key = nullptr;
    size_t tmp_cols = 0;
    unsigned long long *queue = nullptr;  // work-item counter of the in-flight launch
    unsigned long long *prof = nullptr;   // phase timers (OPTIK_PROFILE builds)
    // streaming engine (ik_engine.hpp)
    struct EngineJobHost {
        EngJob dev;
        int T;
        optik_hip_ik_outputs out;
        double *own_x = nullptr, *own_f = nullptr, *own_key = nullptr;  // scratch when the caller skips them
        unsigned long long *own_fs = nullptr;
        bool full_pool = false;  // OPTIK_HIP_IK_FULL_POOL
    };
    std::vector<EngineJobHost> eng_jobs;
    optik_solver_config eng_cfg{};
    double eng_ee[7] = {0, 0, 0, 0, 0, 0, 1};
    bool eng_has_ee = false;
    bool eng_pool_attached = false;  // counted among the users of its device's EnginePool
    size_t eng_C = 0;                // the eng_* slot buffers below are views of that pool (engine_reserve)
    double *eng_d = nullptr;
    int32_t *eng_i32 = nullptr;
    unsigned long long *eng_item = nullptr;
    EngJob *eng_djobs = nullptr;
    unsigned int *eng_counters = nullptr;  // per trip parity {list length, class sizes}, then n_active
    unsigned int *eng_order = nullptr;     // 2 x [NN_CLASSES][C]
    double *eng_carry = nullptr;           // [C][NN_CARRY]
    unsigned int *eng_list = nullptr;      // 2 x [C] class entries by slot, per trip parity
    unsigned int *eng_refill = nullptr;    // [C] slots wanting a work item, per sub-pool range
    unsigned int *eng_cont = nullptr;      // continuation lists: [C] entries (a sub-pool's shards share its range), then [ENG_MAX_POOLS][NN_CONT_SHARDS] counters
    double *hw_dev = nullptr, *hw_pin = nullptr;  // optik_hip_ik_host: device block and pinned staging
    size_t hw_cap = 0;                            // doubles
    // optik_hip_ik_host, one target under the first-success rule: the block the first successful restart writes its
    // answer to (WorkQueue::claim; pinned, host-coherent), the sequence number of the last launch that used it, the
    // request optik_hip_ik_host leaves for the launch it is about to make and whether that launch took it up
    unsigned long long *hw_claim = nullptr;
    hipEvent_t claim_done = nullptr;  // recorded behind such a launch: what the polling host also looks at
    unsigned long long claim_seq = 0;
    // such a launch may still be running on the null stream (set, under `mu`, in the critical section that queues it;
    // cleared by whoever has waited for the null stream)
    bool claim_pending = false;
    unsigned hw_flip = 0;  // which half of the pinned block the next zero-copy call uses
    unsigned long long *nnls_trace = nullptr;  // OPTIK_NNLS_TRACE builds
    unsigned int *eng_trip_log = nullptr;      // OPTIK_ENG_TRIP_LOG diagnostics
    unsigned int *eng_compact = nullptr;       // [ENG_MAX_POOLS][2] counters, then free list [C], move list [C]
    hipStream_t eng_streams[ENG_MAX_POOLS] = {};  // streams of sub-pools 1.. (sub-pool 0 runs on the caller's)
    hipEvent_t eng_pool_ev[ENG_MAX_POOLS][8] = {};
    hipEvent_t eng_fork_ev = nullptr, eng_join_ev[ENG_MAX_POOLS] = {};
    int eng_pools = 1;
    int eng_launches = 0;                      // NNLS launches of the last run, all sub-pools
    unsigned long long *eng_deadline = nullptr;  // device word: wall_clock64() value at which the run's max_time expires
    int eng_tail_restarts = 0;                 // restarts (upper bound) the tail kernel took over in the last run
    int eng_tail_solver = 0;                   // ... and on which solver: 0 none, 1 per-lane, 2 cooperative, 3 quad
    int eng_compactions = 0;
    double *eng_prob = nullptr;            // 2 x [C][2n][n+1]
    double *eng_y = nullptr;               // 2 x [C][2n]
    double *eng_meta = nullptr;            // 2 x [C][2]
    unsigned int *eng_pinned = nullptr;    // host-pinned read-back ring
    hipEvent_t eng_ev[8] = {};
    int eng_trips = 0;                     // trips of the last run
    // per-kernel HIP-event timing of sampled trips (eval, update, nnls, finish)
    static constexpr int ENG_EV = 1024;
    hipEvent_t eng_tev[4][ENG_EV][2] = {};
    int eng_tcount = 0;                    // sampled trips of the last run
    double eng_kernel_ms[4] = {0, 0, 0, 0};
    unsigned long long *eng_nn_total = nullptr;  // [0] problems solved by the NNLS kernel; [1 .. 64] executed evaluations (sharded)
    unsigned long long eng_nn_problems = 0;
    unsigned long long eng_exec_evals = 0;       // objective + gradient evaluations executed by the last run
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

// ---- options ---------------------------------------------------------------------------------------------
// Every tuning option of the kernel layer, in one place.  The defaults come from the environment ONCE, at the
// first use (the OPTIK_* names below); tests and tools change them through optik_hip_set_option (optik_hip.h).
// Nothing else in this library reads the environment (robot_host.cpp: OPTIK_HOST_THREADS, OPTIK_DEVICES).
enum : int { SK_AUTO = 0, SK_QUAD = 1, SK_LANE64 = 2, SK_GENERAL = 3 };
struct Options {
    int solve_kernel = SK_AUTO;      // OPTIK_SOLVE_KERNEL = quad | lane64 | general: which single-launch solver (auto: by size)
    long long engine_slots = 0;      // OPTIK_ENGINE_SLOTS: capacity of the engine's slot pool (0: 393 216, fewer for early-exit jobs)
    int engine_pools = 0;            // OPTIK_ENG_POOLS: sub-pools of an engine run (0: three, fewer for small pools)
    int engine_nnls_budget = 6;      // OPTIK_ENG_NNLS_BUDGET: solve passes per problem per NNLS launch
    int engine_nnls_slack = 1;       // OPTIK_ENG_NNLS_SLACK: ... and per problem: its predicted count + this
    long long engine_tail_max = -1;  // OPTIK_ENG_TAIL_MAX: restarts left at which the quad solver takes a run over (-1: total / 8; 0: never)
    int wide_form = 0;               // OPTIK_WIDE_FORM = lds | hbm: the general solver's form (9 .. 16 joints); 2: one-lane LDS form
    int range_rule = OPTIK_HIP_RANGE_SINGLE_INCLUSIVE;  // OPTIK_RANDOM_RANGE_RULE = new_inclusive: rand 0.9.2 reading of new chains
    int engine_compact = 1;          // (no environment name) drain compaction of an engine run's sub-pools
    int stop_x_legacy = 0;           // (no environment name) nlopt_stop_x of NLopt 2.5: no zero-step rule
};
int solve_kernel_from(const char *e) {
    if (!e) return SK_AUTO;
    if (!std::strcmp(e, "quad")) return SK_QUAD;
    if (!std::strcmp(e, "lane64")) return SK_LANE64;
    if (!std::strcmp(e, "general")) return SK_GENERAL;
    return SK_AUTO;
}
Options &opt() {
    static Options o = [] {
        Options v;
        v.solve_kernel = solve_kernel_from(std::getenv("OPTIK_SOLVE_KERNEL"));
        if (const char *e = std::getenv("OPTIK_ENGINE_SLOTS")) v.engine_slots = std::atoll(e);
        if (const char *e = std::getenv("OPTIK_ENG_POOLS")) v.engine_pools = std::atoi(e);
        if (const char *e = std::getenv("OPTIK_ENG_NNLS_BUDGET")) v.engine_nnls_budget = std::atoi(e) > 0 ? std::atoi(e) : 1;
        if (const char *e = std::getenv("OPTIK_ENG_NNLS_SLACK")) v.engine_nnls_slack = std::atoi(e) > 0 ? std::atoi(e) : 0;
        if (const char *e = std::getenv("OPTIK_ENG_TAIL_MAX")) v.engine_tail_max = std::atoll(e);
        if (const char *e = std::getenv("OPTIK_WIDE_FORM")) v.wide_form = std::strcmp(e, "hbm") == 0 ? 1 : 0;
        if (const char *e = std::getenv("OPTIK_RANDOM_RANGE_RULE"))
            if (std::strcmp(e, "new_inclusive") == 0 || std::strcmp(e, "1") == 0) v.range_rule = OPTIK_HIP_RANGE_NEW_INCLUSIVE;
        return v;
    }();
    return o;
}

// The slot pool of the streaming engine: ONE per device, shared by every chain (robot) of that
// device -- ~1 GB at the default capacity, sized for n = 7 (a superset of every smaller chain's
// planes and records).  An engine run owns it from its set-up to its last kernel (run_mu): runs
// on one GPU gain nothing from overlapping, and a slot's content never outlives a run
// (eng_init_kernel resets every slot).  Freed when the last chain that used it is destroyed.
struct EnginePool {
    std::mutex run_mu;
    size_t C = 0;
    double *d = nullptr;
    int32_t *i32 = nullptr;
    unsigned long long *item = nullptr;
    double *prob = nullptr, *y = nullptr, *meta = nullptr, *carry = nullptr;
    unsigned int *list = nullptr, *refill = nullptr, *cont = nullptr, *order = nullptr, *compact = nullptr;
    int users = 0;
};
constexpr int ENG_MAX_DEVICES = 64;
static EnginePool g_eng_pools[ENG_MAX_DEVICES];
static std::mutex g_eng_pools_mu;  // guards `users`
static EnginePool &engine_pool_of(const optik_hip_chain *ch) {
    const int d = ch->device_id >= 0 && ch->device_id < ENG_MAX_DEVICES ? ch->device_id : 0;
    return g_eng_pools[d];
}
static void engine_pool_free(EnginePool &P) {
    (void)hipFree(P.d); (void)hipFree(P.i32); (void)hipFree(P.item); (void)hipFree(P.prob); (void)hipFree(P.y);
    (void)hipFree(P.meta); (void)hipFree(P.carry); (void)hipFree(P.list); (void)hipFree(P.refill); (void)hipFree(P.cont);
    (void)hipFree(P.order); (void)hipFree(P.compact);
    P.d = nullptr; P.i32 = nullptr; P.item = nullptr; P.prob = P.y = P.meta = P.carry = nullptr;
    P.list = P.refill = P.cont = P.order = P.compact = nullptr;
    P.C = 0;
}



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

int default_range_rule() { return opt().range_rule; }

void set_chain_scales(optik_hip_chain *ch, int rule) {
    ch->range_rule = rule;
    for (int k = 0; k < ch->n; ++k) {
        const double lb = ch->wide ? ch->whost.lb[k] : ch->host.lb[k], ub = ch->wide ? ch->whost.ub[k] : ch->host.ub[k];
        // infinite limits (continuous joints) make random_range panic in the
        // reference (quirk Q5); restarts > 0 are refused at launch time instead.
        ch->scale[k] = (std::isfinite(lb) && std::isfinite(ub)) ? uniform_scale(lb, ub, rule) : NAN;
        if (ch->wide) ch->whost.scale[k] = ch->scale[k];
    }
}

int ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(OPTIK_HIP_ENODEVICE, std::string("no HIP device available: ")
                                             + (e == hipSuccess ? "device count is 0" : hipGetErrorString(e)));
    return 0;
}

// A chain lives on the device that was current when it was created; its entry points make that
// device current for the calling thread for the duration of the call and restore the caller's
// device on every exit path (device_scope.hpp) -- a host that also drives torch / RCCL on the
// thread finds its own device current again.
#define BIND_DEVICE(CH)                                                                 \
    optik::DeviceScope dev_scope_((CH)->device_id);                                     \
    if (!dev_scope_.ok()) return fail(OPTIK_HIP_ENODEVICE, "hipSetDevice(" + std::to_string((CH)->device_id) + ") failed")

// Dispatch on (n, trailing fixed joint).
// Kernels are instantiated for 1 <= n <= 8 revolute joints, each with and without a trailing
// fixed joint; the streaming engine for n <= 7 (n + 1 <= 8 rows fit its register-resident
// NNLS) -- an 8-DoF chain's engine jobs run on the single-kernel path, same results.
#define OPTIK_N_RANGE_MSG "this kernel is built for 1 <= n <= 8 revolute joints"
#define OPTIK_DISPATCH_ONE(NN, CALL)                                                   \
    if (!done_ && n_ == NN) {                                                          \
        if (tip_) { CALL(NN, true); } else { CALL(NN, false); }                        \
        done_ = true;                                                                  \
    }
#define OPTIK_DISPATCH(CH, CALL)                                                       \
    do {                                                                               \
        const int n_ = (CH)->n;                                                        \
        const bool tip_ = (CH)->tip;                                                   \
        bool done_ = false;                                                            \
        OPTIK_DISPATCH_ONE(1, CALL)                                                     \
        OPTIK_DISPATCH_ONE(2, CALL) OPTIK_DISPATCH_ONE(3, CALL) OPTIK_DISPATCH_ONE(4, CALL) \
        OPTIK_DISPATCH_ONE(5, CALL) OPTIK_DISPATCH_ONE(6, CALL) OPTIK_DISPATCH_ONE(7, CALL) \
        OPTIK_DISPATCH_ONE(8, CALL)                                                     \
        if (!done_) return fail(OPTIK_HIP_EUNSUPPORTED, OPTIK_N_RANGE_MSG);            \
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
    if (n < 1 || n > WIDE_MAX_DOF) return fail(OPTIK_HIP_EUNSUPPORTED, "num_positions must be in 1..16");
    if (n_joints != n && n_joints != n + 1)
        return fail(OPTIK_HIP_EUNSUPPORTED, "chain must be n revolute joints plus an optional trailing fixed joint");
    bool prismatic = false;
    for (int j = 0; j < n; ++j) {
        if (types[j] == OPTIK_JOINT_PRISMATIC) prismatic = true;
        else if (types[j] != OPTIK_JOINT_REVOLUTE)
            return fail(OPTIK_HIP_EUNSUPPORTED, "the first n joints of the chain must be revolute or prismatic");
    }
    if (n_joints == n + 1 && types[n] != OPTIK_JOINT_FIXED)
        return fail(OPTIK_HIP_EUNSUPPORTED, "joint after the last revolute joint must be fixed");
    if (prismatic && n > MAX_DOF)
        return fail(OPTIK_HIP_EUNSUPPORTED, "prismatic joints are supported for chains of at most 8 joint positions");
    if (int rc = ensure_device()) return rc;

    auto *ch = new optik_hip_chain();
    std::memset(&ch->host, 0, sizeof ch->host);
    std::memset(&ch->whost, 0, sizeof ch->whost);
    ch->n = n;
    if (n > MAX_DOF) {
        // 9 .. 16 joint positions: the general kernels of ik_wide.hpp (one table, joint count at run time)
        ch->wide = true;
        ch->n_joints = n_joints;
        ch->tip = (n_joints == n + 1);
        WideChainDev &w = ch->whost;
        w.n_pos = n;
        w.has_tip = ch->tip;
        for (int j = 0; j < n_joints; ++j)
            for (int k = 0; k < 7; ++k) w.origin[j][k] = origins[j * 7 + k];
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < 3; ++k) w.axis[j][k] = axes[j * 3 + k];
        for (int k = 0; k < n; ++k) { w.lb[k] = lb[k]; w.ub[k] = ub[k]; }
        set_chain_scales(ch, default_range_rule());
        seed_from_u64(42, ch->key);  // RNG_SEED, lib.rs:360
        hipError_t e = hipMalloc(&ch->wdev, sizeof(WideChainDev));
        if (e == hipSuccess) e = hipMemcpy(ch->wdev, &ch->whost, sizeof(WideChainDev), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (ch->wdev) (void)hipFree(ch->wdev);
            delete ch;
            return fail(OPTIK_HIP_ENODEVICE, std::string("chain upload: ") + hipGetErrorString(e));
        }
        int dev = 0;
        hipGetDevice(&dev);
        ch->device_id = dev;
        hipDeviceGetAttribute(&ch->num_cus, hipDeviceAttributeMultiprocessorCount, dev);
        hipDeviceGetAttribute(&ch->wall_clock_khz, hipDeviceAttributeWallClockRate, dev);
        *out = ch;
        return 0;
    }
    ch->prismatic = prismatic;
    ch->n_joints = n_joints;
    for (int j = 0; j < n_joints; ++j) {
        ch->types[j] = types[j];
        for (int k = 0; k < 3; ++k) ch->axis_all[j][k] = axes[j * 3 + k];
    }
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
    }
    set_chain_scales(ch, default_range_rule());
    seed_from_u64(42, ch->key);  // RNG_SEED, lib.rs:360
    hipError_t e = hipMalloc(&ch->dev, sizeof(ChainDev));
    if (e == hipSuccess) e = hipMemcpy(ch->dev, &ch->host, sizeof(ChainDev), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        delete ch;
        return fail(OPTIK_HIP_ENODEVICE, std::string("chain upload: ") + hipGetErrorString(e));
    }
    int dev = 0;
    hipGetDevice(&dev);
    ch->device_id = dev;
    hipDeviceGetAttribute(&ch->num_cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&ch->wall_clock_khz, hipDeviceAttributeWallClockRate, dev);
    *out = ch;
    return 0;
}

void optik_hip_chain_destroy(optik_hip_chain *ch) {
    if (!ch) return;
    optik::DeviceScope dev_scope(ch->device_id);  // frees and the pool's last sync run on the chain's device
    if (ch->claim_pending) (void)hipStreamSynchronize(nullptr);
    if (ch->dev) hipFree(ch->dev);
    if (ch->wdev) hipFree(ch->wdev);
    if (ch->wide_ws) hipFree(ch->wide_ws);
    if (ch->tile_recs) hipFree(ch->tile_recs);
    if (ch->first_success) hipFree(ch->first_success);
    if (ch->tmp_x) hipFree(ch->tmp_x);
    if (ch->tmp_f) hipFree(ch->tmp_f);
    if (ch->tmp_key) hipFree(ch->tmp_key);
    if (ch->queue) hipFree(ch->queue);
    if (ch->eng_pool_attached) {  // the slot pool belongs to the device: released with its last user
        EnginePool &P = engine_pool_of(ch);
        std::lock_guard<std::mutex> run(P.run_mu);
        std::lock_guard<std::mutex> g(g_eng_pools_mu);
        if (--P.users == 0) { (void)hipDeviceSynchronize(); engine_pool_free(P); }
    }
    if (ch->eng_djobs) hipFree(ch->eng_djobs);
    if (ch->eng_counters) hipFree(ch->eng_counters);
    if (ch->eng_trip_log) hipFree(ch->eng_trip_log);
    if (ch->eng_deadline) hipFree(ch->eng_deadline);
    if (ch->hw_dev) hipFree(ch->hw_dev);
    if (ch->hw_pin) hipHostFree(ch->hw_pin);
    if (ch->hw_claim) hipHostFree(ch->hw_claim);
    if (ch->claim_done) hipEventDestroy(ch->claim_done);
    if (ch->eng_pinned) hipHostFree(ch->eng_pinned);
    for (auto &pe : ch->eng_pool_ev) for (auto &e : pe) if (e) hipEventDestroy(e);
    if (ch->eng_fork_ev) hipEventDestroy(ch->eng_fork_ev);
    for (auto &e : ch->eng_join_ev) if (e) hipEventDestroy(e);
    for (auto &st : ch->eng_streams) if (st) hipStreamDestroy(st);
    for (auto &e : ch->eng_ev) if (e) hipEventDestroy(e);
    for (auto &k : ch->eng_tev) for (auto &p2 : k) for (auto &e : p2) if (e) hipEventDestroy(e);
    if (ch->eng_nn_total) hipFree(ch->eng_nn_total);
    for (int i = 0; i < optik_hip_chain::EV_POOL; ++i) {
        if (ch->ev0[i]) hipEventDestroy(ch->ev0[i]);
        if (ch->ev1[i]) hipEventDestroy(ch->ev1[i]);
    }
    delete ch;
}

int32_t optik_hip_chain_num_positions(const optik_hip_chain *ch) { return ch ? ch->n : 0; }

int optik_hip_chain_set_range_rule(optik_hip_chain *ch, int32_t rule) {
    if (!ch || (rule != OPTIK_HIP_RANGE_SINGLE_INCLUSIVE && rule != OPTIK_HIP_RANGE_NEW_INCLUSIVE))
        return fail(OPTIK_HIP_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lock(ch->mu);
    set_chain_scales(ch, rule);
    if (ch->wide) {  // (the scales of a wide chain are part of its device table)
        BIND_DEVICE(ch);
        HIP_TRY(hipMemcpy(ch->wdev, &ch->whost, sizeof(WideChainDev), hipMemcpyHostToDevice));
    }
    return 0;
}

int32_t optik_hip_chain_range_rule(const optik_hip_chain *ch) { return ch ? ch->range_rule : -1; }

int optik_hip_eval_batch(const optik_hip_chain *ch, const optik_solver_config *cfg, const double *target7,
                         const double *ee_offset7, const double *d_q, int64_t B, double *d_f, double *d_g,
                         void *stream) {
    if (!ch || !cfg || !target7 || !d_q || !d_f || B < 0) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (B == 0) return 0;
    if (ch->prismatic)
        return fail(OPTIK_HIP_EUNSUPPORTED,
                    "prismatic joints: only forward kinematics is available (the reference's Jacobian panics, kinematics.rs:185)");
    BIND_DEVICE(ch);
    if (ch->wide) {
        WideBatchLaunch w;
        std::memset(&w, 0, sizeof w);
        w.chain = ch->wdev;
        make_eval_params(cfg->linear_weight, cfg->angular_weight, ee_offset7, w.ep);
        std::memcpy(w.target, target7, sizeof w.target);
        w.q = d_q; w.B = B; w.f = d_f; w.g = d_g;
        HIP_TRY(wide_batch_launch(0, grid_for(ch, B, 256, 8), (hipStream_t)stream, w));
        return 0;
    }
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
    BIND_DEVICE(ch);
    if (ch->prismatic) {
        if (d_jac)
            return fail(OPTIK_HIP_EUNSUPPORTED,
                        "joint_jacobian: prismatic joints are not implemented (the reference panics: kinematics.rs:185 todo!())");
        FkGeneralLaunch g;
        std::memset(&g, 0, sizeof g);
        g.n_joints = ch->n_joints;
        g.n_pos = ch->n;
        std::memcpy(g.types, ch->types, sizeof g.types);
        std::memcpy(g.origin, ch->host.origin, sizeof g.origin);
        std::memcpy(g.axis, ch->axis_all, sizeof g.axis);
        const double ident[7] = {0, 0, 0, 0, 0, 0, 1};
        std::memcpy(g.ee_offset, ee_offset7 ? ee_offset7 : ident, sizeof ident);
        g.q = d_q; g.B = B; g.pose = d_pose;
        hipLaunchKernelGGL(fk_general_kernel, dim3(grid_for(ch, B, 256, 8)), dim3(256), 0, (hipStream_t)stream, g);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (ch->wide) {
        WideBatchLaunch w;
        std::memset(&w, 0, sizeof w);
        w.chain = ch->wdev;
        const double one_w[3] = {1, 1, 1};
        make_eval_params(one_w, one_w, ee_offset7, w.ep);
        w.q = d_q; w.B = B; w.pose = d_pose; w.jac = d_jac;
        HIP_TRY(wide_batch_launch(1, grid_for(ch, B, 256, 8), (hipStream_t)stream, w));
        return 0;
    }
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
    BIND_DEVICE(ch);
    if (ch->wide) {
        WideBatchLaunch w;
        std::memset(&w, 0, sizeof w);
        w.chain = ch->wdev;
        std::memcpy(w.key, ch->key, sizeof w.key);
        w.first = first; w.B = count; w.q_out = d_q;
        HIP_TRY(wide_batch_launch(2, grid_for(ch, count, 256, 8), (hipStream_t)stream, w));
        return 0;
    }
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

}  // extern "C"

// optik_hip_ik_batch with the chain's launch mutex already held.
static int ik_batch_locked(optik_hip_chain *ch, const optik_solver_config *cfg, const double *d_targets,
                           const double *d_x0, int32_t T, const double *ee_offset7, uint64_t restart_begin,
                           uint64_t restart_end, uint32_t flags, double deadline_s, const optik_hip_ik_outputs *out,
                           void *stream_v, bool claim_request = false, bool *claim_armed = nullptr);

extern "C" {

int optik_hip_ik_batch(optik_hip_chain *ch, const optik_solver_config *cfg, const double *d_targets,
                       const double *d_x0, int32_t T, const double *ee_offset7, uint64_t restart_begin,
                       uint64_t restart_end, uint32_t flags, double deadline_s, const optik_hip_ik_outputs *out,
                       void *stream_v) {
    if (!ch) return fail(OPTIK_HIP_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lock(ch->mu);
    return ik_batch_locked(ch, cfg, d_targets, d_x0, T, ee_offset7, restart_begin, restart_end, flags, deadline_s, out,
                           stream_v);
}

}  // extern "C"

static int ik_batch_locked(optik_hip_chain *ch, const optik_solver_config *cfg, const double *d_targets,
                           const double *d_x0, int32_t T, const double *ee_offset7, uint64_t restart_begin,
                           uint64_t restart_end, uint32_t flags, double deadline_s, const optik_hip_ik_outputs *out,
                           void *stream_v, bool claim_request, bool *claim_armed) {
    if (claim_armed) *claim_armed = false;
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
    if (ch->prismatic)
        return fail(OPTIK_HIP_EUNSUPPORTED,
                    "prismatic joints: only forward kinematics is available (the reference's Jacobian panics, kinematics.rs:185)");
    BIND_DEVICE(ch);
    // (a single call that returned on its first success may have left its launch running on the null stream: a
    // launch on another stream shares the chain's workspace with it and waits; on the null stream it queues behind)
    // (on the null stream the flag stays: optik_hip_ik_host's staged path still has to know)
    if (ch->claim_pending && stream != nullptr) { HIP_TRY(hipStreamSynchronize(nullptr)); ch->claim_pending = false; }

    // selection tiles: 4096 restarts per 256-thread block
    constexpr int SEL_TILE = 4096;
    const uint64_t tiles_per_target = (R + SEL_TILE - 1) / SEL_TILE;
    const uint64_t n_tiles64 = tiles_per_target * (uint64_t)T;
    // (HIP rejects a launch whose grid.x * block.x reaches 2^32: 256-thread tile blocks cap the tiles at 2^24 - 1)
    if (n_tiles64 * 256ull >= (1ull << 32))
        return fail(OPTIK_HIP_EINVAL, "too many restarts / targets in one launch (2^24 or more selection tiles)");
    const int n_tiles = (int)n_tiles64;
    const size_t cols = (size_t)T * (size_t)R;

    if ((size_t)n_tiles > ch->tile_cap) {
        if (ch->tile_recs) HIP_TRY(hipFree(ch->tile_recs));
        ch->tile_recs = nullptr;
        HIP_TRY(hipMalloc(&ch->tile_recs, sizeof(TileRec) * (size_t)n_tiles));
        ch->tile_cap = (size_t)n_tiles;
    }
    if (!ch->queue) { HIP_TRY(hipMalloc(&ch->queue, sizeof(unsigned long long))); ch->queue_clean = false; }
    // (the words are only known to be clean to a launch queued behind the selection kernel that cleaned them)
    if (stream != ch->clean_stream) { ch->queue_clean = false; ch->fs_clean = 0; }
    if (!ch->queue_clean) HIP_TRY(hipMemsetAsync(ch->queue, 0, sizeof(unsigned long long), stream));
    ch->queue_clean = false;  // (until this launch's selection kernel has put it back)
    const bool early = (flags & OPTIK_HIP_IK_EARLY_EXIT) && cfg->solution_mode == 2;
    size_t fs_clean_after = ch->fs_clean;  // (a launch without early exit leaves the words alone)
    if (early) {
        if ((size_t)T > ch->fs_cap) {
            if (ch->first_success) HIP_TRY(hipFree(ch->first_success));
            ch->first_success = nullptr;
            ch->fs_clean = 0;
            HIP_TRY(hipMalloc(&ch->first_success, sizeof(unsigned long long) * (size_t)T));
            ch->fs_cap = (size_t)T;
        }
        if (ch->fs_clean < (size_t)T)
            HIP_TRY(hipMemsetAsync(ch->first_success, 0xff, sizeof(unsigned long long) * (size_t)T, stream));
        fs_clean_after = std::max(ch->fs_clean, (size_t)T);  // once the selection kernel has put words [0, T) back
        ch->fs_clean = 0;
    }
    // the selection needs the per-restart x / f / key: scratch if the caller skips them
    const bool want_win = out->d_win_x || out->d_win_f || out->d_win_idx || out->d_win_key;
    double *px = out->d_x, *pf = out->d_f, *pk = nullptr;
    if (want_win) {
        const bool need_xf = (!px || !pf) && (out->d_win_x || out->d_win_f);
        if (cols > ch->tmp_cols) {
            if (ch->tmp_x) HIP_TRY(hipFree(ch->tmp_x));
            if (ch->tmp_f) HIP_TRY(hipFree(ch->tmp_f));
            if (ch->tmp_key) HIP_TRY(hipFree(ch->tmp_key));
            ch->tmp_x = ch->tmp_f = ch->tmp_key = nullptr;
            HIP_TRY(hipMalloc(&ch->tmp_key, sizeof(double) * cols));
            ch->tmp_cols = cols;
        }
        if (need_xf && !ch->tmp_x) {
            HIP_TRY(hipMalloc(&ch->tmp_x, sizeof(double) * ch->tmp_cols * (size_t)ch->n));
            HIP_TRY(hipMalloc(&ch->tmp_f, sizeof(double) * ch->tmp_cols));
        }
        pk = ch->tmp_key;
        if (!px && out->d_win_x) px = ch->tmp_x;
        if (!pf && out->d_win_f) pf = ch->tmp_f;
    }

    SolveLaunch a;
    std::memset(&a, 0, sizeof a);
    a.chain = ch->dev;  // (null for a wide chain: its launch takes ch->wdev)
    make_eval_params(cfg->linear_weight, cfg->angular_weight, ee_offset7, a.ep);
    fill_solve_params(cfg, a.sp, opt().stop_x_legacy != 0);
    std::memcpy(a.key, ch->key, sizeof a.key);
    std::memcpy(a.scale, ch->scale, sizeof a.scale);  // (n <= 8; a wide chain's scales are in its table)
    a.wq.next_item = ch->queue;
    a.wq.total_items = (unsigned long long)cols;
    a.wq.n_restarts = R;
    a.wq.restart_begin = restart_begin;
    a.wq.targets = d_targets;
    a.wq.x0 = d_x0;
    a.wq.first_success = early ? ch->first_success : nullptr;
    a.wq.find_any = (early && (flags & OPTIK_HIP_IK_FIND_ANY)) ? 1 : 0;
    a.wq.claim = nullptr;
    a.wq.claim_seq = 0;
    a.wq.restart_major = (flags & OPTIK_HIP_IK_RESTART_MAJOR) ? 1 : 0;
    a.wq.n_targets = (unsigned long long)T;
    a.wq.deadline = 0;
    a.wq.quality = (cfg->solution_mode == 1);
    a.wq.out_x = px;
    a.wq.out_f = pf;
    a.wq.out_key = pk;
    a.wq.out_status = out->d_status;
    a.wq.out_evals = out->d_evals;
    a.wq.prof = nullptr;
#ifdef OPTIK_PROFILE
    if (!ch->prof) HIP_TRY(hipMalloc(&ch->prof, 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(ch->prof, 0, 8 * sizeof(unsigned long long), stream));
    a.wq.prof = ch->prof;
#endif
    a.deadline_ticks = 0;
    if (deadline_s > 0.0) {
        const double khz = ch->wall_clock_khz > 0 ? (double)ch->wall_clock_khz : 100000.0;
        a.deadline_ticks = (unsigned long long)(deadline_s * khz * 1e3);
        if (a.deadline_ticks == 0) a.deadline_ticks = 1;
    }

    // Which solver (option solve_kernel; same results, bit for bit): the quad solver of ik_quad.hpp (a restart per
    // quad of lanes, its state spread over the quad, NNLS matrix in LDS; n <= 8), from one full load of the chip
    // on the lane-per-restart form of ik_lane64.hpp (n <= 7), or -- `general` -- the run-time-n solver of
    // ik_wide.hpp on a chain of at most 8 joints too: a third, independently written device solver for the parity
    // tests; chains of 9 .. 16 joints always run on it.
    const int sk = opt().solve_kernel;
    bool widek = ch->wide || sk == SK_GENERAL;
    if (widek && !ch->wide) {
        // the chain's table in the general kernels' layout (uploaded per call: a test path)
        WideChainDev &w = ch->whost;
        std::memset(&w, 0, sizeof w);
        w.n_pos = ch->n;
        w.has_tip = ch->tip;
        for (int j = 0; j < ch->n + (ch->tip ? 1 : 0); ++j)
            for (int k = 0; k < 7; ++k) w.origin[j][k] = ch->host.origin[j][k];
        for (int j = 0; j < ch->n; ++j) {
            for (int k = 0; k < 3; ++k) w.axis[j][k] = ch->host.axis[j][k];
            w.lb[j] = ch->host.lb[j]; w.ub[j] = ch->host.ub[j]; w.scale[j] = ch->scale[j];
        }
        if (!ch->wdev) HIP_TRY(hipMalloc(&ch->wdev, sizeof(WideChainDev)));
        HIP_TRY(hipMemcpy(ch->wdev, &w, sizeof(WideChainDev), hipMemcpyHostToDevice));
    }
    const bool quadk = !widek;
    // the throughput form for n <= 7: one restart per lane, bounded sub-problems in class order (ik_lane64.hpp)
    // (the default from one full load of the chip on -- 64 restarts for each of its four waves per CU: below that a
    // launch is as long as its longest restart, and the quad solver's trip is the shorter one; tools/lane_vs_quad_probe.py)
    bool lanek = quadk && ch->n <= 7 && sk != SK_QUAD;
    const bool lane_forced = lanek && sk == SK_LANE64;
    // Persistent waves, each pulling work items until the queue is dry: as many as a CU holds
    // (lane kernel: 2 workgroups, LDS-bound; cooperative kernel: 4, one per SIMD), times the CU count.
    const int cus = ch->num_cus > 0 ? ch->num_cus : 256;
    // (quad solver: a launch with no more work items than the chip has SIMDs runs one restart per wave on the
    // one-wave-per-SIMD build -- no scratch, the lowest latency per iteration; anything bigger on the
    // two-waves-per-SIMD build)
    const long long wide_waves_per_cu = 8;  // resident waves per CU of the general solver (two per SIMD)
    const bool quad_latency = quadk && (long long)cols <= (long long)cus * 4 && !lane_forced;
    // (not for a Speed batch's latency-sized rounds: restart-major hand-out with early exit keeps a few restarts per
    // target in flight and abandons most of the rest -- the quad solver's shorter trip wins there)
    lanek = lanek && !quad_latency
            && (lane_forced || ((long long)cols >= (long long)cus * lane_solve_waves_per_cu() * 64
                                && !(early && (flags & OPTIK_HIP_IK_RESTART_MAJOR))));
    long long cap = (long long)cus * (lanek ? lane_solve_waves_per_cu() : quadk ? (quad_latency ? 4 : quad_solve_waves_per_cu(ch->n)) : wide_waves_per_cu);
    const long long per_wave_max = (quadk && !lanek) ? QUADS_PER_WAVE_HOST : WAVE;
    // fewer work items than the chip holds: one restart per wave (or as few as fit).  A
    // restart-major Speed batch keeps about eight restarts per target in flight: the waves pull
    // the higher indices of the targets still unsolved as they go
    long long resident = (long long)cols;
    // (but never fewer than one restart per resident wave: a small batch has the chip to itself, and
    // the more of a target's restarts run at once the sooner its first success comes)
    const long long inflight = 8;  // restarts per target in flight
    // (a few targets have the chip to themselves: two restarts per resident wave at least, 32 per
    // target up to 256 targets -- measured: 64 targets 1.01 -> 0.79 ms, 256: 1.66 -> 1.47 ms, and the
    // few hundred targets a big batch's short engine round leaves over 8 ms sooner)
    if (early && (flags & OPTIK_HIP_IK_RESTART_MAJOR) && resident > (long long)T * inflight) {
        const long long floor_res = std::max(2 * cap, (long long)T * 32);
        resident = std::max((long long)T * inflight, std::min(resident, floor_res));
    }
    long long lanes = (resident + cap - 1) / cap;
    if (lanes < 1) lanes = 1;
    if (lanes > per_wave_max) lanes = per_wave_max;
    // The general solver's two forms (ik_wide.hpp): one restart per wave with its arrays in LDS and the wave's 64
    // lanes working on it together, or a restart per lane with the HBM workspace.  The first has the short
    // dependent chain and no HBM traffic, the second 64 times the restarts in flight -- and the first wins at
    // every size and joint count measured (tools/wide_chain_bench.py, 262 144 restarts: 1.31 / 0.88 / 0.83 / 1.26 M
    // restarts/s at 9 / 10 / 12 / 16 joints against 1.05 / 0.66 / 0.42 / 0.40 M; a launch on the HBM form takes
    // 50 - 100 ms however small it is).  Option wide_form = hbm selects the HBM form (tests, comparisons).
    bool wide_lds = false;
    if (widek) {
        wide_lds = opt().wide_form != 1;
        if (wide_lds) lanes = 1;
    }
    a.wq.lanes = (int)lanes;
    // a single call under the first-success rule on the quad solver: the first success goes to the host at once
    if (claim_request && quadk && !lanek && a.wq.find_any && T == 1 && ch->hw_claim) {
        a.wq.claim = ch->hw_claim;
        a.wq.claim_seq = ++ch->claim_seq;
        if (claim_armed) *claim_armed = true;
        if (stream == nullptr) ch->claim_pending = true;  // (the caller may return before this launch has ended)
    }
    long long grid_ll = (resident + lanes - 1) / lanes;
    if (grid_ll > cap) grid_ll = cap;
    const int grid = (int)grid_ll;

    const int ev_slot = ch->ev_count % optik_hip_chain::EV_POOL;
    if (ch->timing) {
        if (!ch->ev0[ev_slot]) { HIP_TRY(hipEventCreate(&ch->ev0[ev_slot])); HIP_TRY(hipEventCreate(&ch->ev1[ev_slot])); }
        HIP_TRY(hipEventRecord(ch->ev0[ev_slot], stream));
    }
    int lds = 0;
    if (widek) {
        // 9 .. 16 joint positions: one restart per lane on the general kernel, eight waves per CU, every
        // resident wave with its own block of the restart workspace (ik_wide.hpp)
        // (one restart per wave -- a single ik() call's rounds --: the restart's arrays in the wave's LDS)
        const bool lds_form = wide_lds;
        if (!lds_form && (size_t)grid > ch->wide_ws_waves) {
            if (ch->wide_ws) HIP_TRY(hipFree(ch->wide_ws));
            ch->wide_ws = nullptr; ch->wide_ws_waves = 0;
            HIP_TRY(hipMalloc(&ch->wide_ws, sizeof(double) * wide_ws_doubles_per_wave() * (size_t)grid));
            ch->wide_ws_waves = (size_t)grid;
        }
        WideSolveLaunch w;
        std::memset(&w, 0, sizeof w);
        w.chain = ch->wdev;
        w.ep = a.ep; w.sp = a.sp; w.wq = a.wq;
        std::memcpy(w.key, ch->key, sizeof w.key);
        w.deadline_ticks = a.deadline_ticks;
        w.ws = ch->wide_ws;
        lds = lds_form ? wide_lds_bytes() : (int)sizeof(WideChainDev);
        HIP_TRY(wide_solve_launch(grid, stream, w, lds_form, opt().wide_form != 2));
    } else if (lanek) {
        HIP_TRY(lane_solve_launch(ch->n, ch->tip, grid, stream, a, &lds));
    } else if (quadk) {
        HIP_TRY(quad_solve_launch(ch->n, ch->tip, grid, stream, a, &lds, quad_latency));
    }
    else return fail(OPTIK_HIP_EUNSUPPORTED, "no solver for this chain in this build");
    HIP_TRY(hipGetLastError());
    if (ch->timing) { HIP_TRY(hipEventRecord(ch->ev1[ev_slot], stream)); ch->ev_count += 1; }
    ch->last.grid = grid; ch->last.block = WAVE; ch->last.lds_bytes = lds; ch->last.tiles = n_tiles;

    if (want_win) {
        SelectLaunch s;
        std::memset(&s, 0, sizeof s);
        s.out_key = pk; s.out_x = px; s.out_f = pf;
        s.tile_recs = ch->tile_recs;
        s.tiles_per_target = (int)tiles_per_target;
        s.tile = SEL_TILE;
        s.n = ch->n;
        s.restart_begin = restart_begin;
        s.n_restarts = R;
        s.ld = cols;
        s.win_x = out->d_win_x; s.win_f = out->d_win_f;
        s.win_idx = (unsigned long long *)out->d_win_idx; s.win_key = out->d_win_key;
        s.reset_queue = ch->queue;
        s.reset_fs = early ? ch->first_success : nullptr;
        if (tiles_per_target == 1) {
            hipLaunchKernelGGL(ik_select_small_kernel, dim3(T), dim3(256), 0, stream, s);
            HIP_TRY(hipGetLastError());
        } else {
            hipLaunchKernelGGL(ik_tile_argmin_kernel, dim3((unsigned)n_tiles), dim3(256), 0, stream, s);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(ik_select_kernel, dim3(T), dim3(WAVE), 0, stream, s);
            HIP_TRY(hipGetLastError());
        }
        ch->queue_clean = true;
        ch->fs_clean = fs_clean_after;
        ch->clean_stream = stream;
    }
    return 0;
}

extern "C" {

// ---- streaming engine: submit jobs, then run them through the shared slot pool ----


int optik_hip_engine_submit(optik_hip_chain *ch, const optik_solver_config *cfg, const double *d_targets,
                            const double *d_x0, int32_t T, const double *ee_offset7, uint64_t restart_begin,
                            uint64_t restart_end, uint32_t flags, const optik_hip_ik_outputs *out) {
    if (!ch || !cfg || !d_targets || !d_x0 || !out || T < 1) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (restart_end <= restart_begin) return fail(OPTIK_HIP_EINVAL, "empty restart range");
    if (cfg->solution_mode != 1 && cfg->solution_mode != 2)
        return fail(OPTIK_HIP_EINVAL, "solution_mode must be 1 (Quality) or 2 (Speed)");
    if (restart_end > 1 || restart_begin > 0)
        for (int k = 0; k < ch->n; ++k)
            if (std::isnan(ch->scale[k]))
                return fail(OPTIK_HIP_EINVAL, "random restarts need finite joint limits (reference: random_range panics)");
    std::lock_guard<std::mutex> lock(ch->mu);
    if (ch->prismatic)
        return fail(OPTIK_HIP_EUNSUPPORTED,
                    "prismatic joints: only forward kinematics is available (the reference's Jacobian panics, kinematics.rs:185)");
    BIND_DEVICE(ch);
    const double ident[7] = {0, 0, 0, 0, 0, 0, 1};
    const double *ee = ee_offset7 ? ee_offset7 : ident;
    if (ch->eng_jobs.empty()) {
        ch->eng_cfg = *cfg;
        std::memcpy(ch->eng_ee, ee, sizeof ident);
        ch->eng_has_ee = ee_offset7 != nullptr;
    } else {
        const optik_solver_config &c0 = ch->eng_cfg;
        const bool same = c0.tol_f == cfg->tol_f && c0.tol_df == cfg->tol_df && c0.tol_dx == cfg->tol_dx
                          && std::memcmp(c0.linear_weight, cfg->linear_weight, sizeof c0.linear_weight) == 0
                          && std::memcmp(c0.angular_weight, cfg->angular_weight, sizeof c0.angular_weight) == 0
                          && std::memcmp(ch->eng_ee, ee, sizeof ident) == 0;
        if (!same) return fail(OPTIK_HIP_EINVAL, "jobs pooled in one engine run must share tolerances, weights and ee_offset");
    }
    const uint64_t R = restart_end - restart_begin;
    if (((R + 4095) / 4096) * (uint64_t)T * 256ull >= (1ull << 32))
        return fail(OPTIK_HIP_EINVAL, "too many restarts / targets in one job (2^24 or more selection tiles)");
    optik_hip_chain::EngineJobHost j;
    std::memset(&j.dev, 0, sizeof j.dev);
    j.T = T;
    j.out = *out;
    const size_t cols = (size_t)T * (size_t)R;
    const bool want_win = out->d_win_x || out->d_win_f || out->d_win_idx || out->d_win_key;
    double *px = out->d_x, *pf = out->d_f, *pk = nullptr;
    // (a failed allocation releases what this job already holds)
#define JOB_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            if (j.own_x) (void)hipFree(j.own_x);                                             \
            if (j.own_f) (void)hipFree(j.own_f);                                             \
            if (j.own_key) (void)hipFree(j.own_key);                                         \
            if (j.own_fs) (void)hipFree(j.own_fs);                                           \
            return fail(OPTIK_HIP_ENOMEM, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                    \
    } while (0)
    // (an 8-DoF chain's jobs run through ik_batch_locked, which brings its own scratch)
    if (want_win && ch->n <= 7) {
        JOB_TRY(hipMalloc(&j.own_key, sizeof(double) * cols));
        pk = j.own_key;
        if (!px && out->d_win_x) { JOB_TRY(hipMalloc(&j.own_x, sizeof(double) * cols * (size_t)ch->n)); px = j.own_x; }
        if (!pf && out->d_win_f) { JOB_TRY(hipMalloc(&j.own_f, sizeof(double) * cols)); pf = j.own_f; }
    }
    const bool early = (flags & OPTIK_HIP_IK_EARLY_EXIT) && cfg->solution_mode == 2;
    if (early) JOB_TRY(hipMalloc(&j.own_fs, sizeof(unsigned long long) * (size_t)T));
#undef JOB_TRY
    j.dev.targets = d_targets;
    j.dev.x0 = d_x0;
    j.dev.item_base = ch->eng_jobs.empty() ? 0ull : ch->eng_jobs.back().dev.item_base + ch->eng_jobs.back().dev.n_items;
    j.dev.n_items = cols;
    j.dev.n_restarts = R;
    j.dev.restart_begin = restart_begin;
    j.dev.out_x = px; j.dev.out_f = pf; j.dev.out_key = pk;
    j.dev.out_status = out->d_status; j.dev.out_evals = out->d_evals;
    j.dev.first_success = j.own_fs;
    j.full_pool = (flags & OPTIK_HIP_IK_FULL_POOL) != 0;
    j.dev.quality = (cfg->solution_mode == 1);
    j.dev.restart_major = (early && T > 1) ? 1 : 0;
    j.dev.find_any = (early && (flags & OPTIK_HIP_IK_FIND_ANY)) ? 1 : 0;
    j.dev.n_targets = (unsigned long long)T;
    ch->eng_jobs.push_back(j);
    return 0;
}

// Device memory and streams of the streaming engine for a pool of AC slots: grows the device's
// shared pool when it holds less and points the chain's eng_* fields at it (the caller holds the
// pool's run_mu).  ni: int planes (the same for every n).
static int engine_reserve(optik_hip_chain *ch, size_t AC, int /*nd*/, int ni, int /*rec_len*/, hipStream_t stream) {
    EnginePool &P = engine_pool_of(ch);
    constexpr int nd = EngLayout<7>::ND, rec_len = rec_stride<7>();
    constexpr size_t nn = 7;
    if (AC > P.C) {
        HIP_TRY(hipDeviceSynchronize());  // (another chain's last run may still be draining on its streams)
        engine_pool_free(P);
        HIP_TRY(hipMalloc(&P.d, sizeof(double) * (size_t)((nd + 1) / 2 * 2) * ((AC + 63) / 64 * 64)));  // whole 64-slot tiles of plane pairs
        HIP_TRY(hipMalloc(&P.i32, sizeof(int32_t) * (size_t)ni * AC));
        HIP_TRY(hipMalloc(&P.item, sizeof(unsigned long long) * AC));
        HIP_TRY(hipMalloc(&P.prob, sizeof(double) * AC * rec_len));
        HIP_TRY(hipMalloc(&P.y, sizeof(double) * AC * (2 * nn)));
        HIP_TRY(hipMalloc(&P.meta, sizeof(double) * AC * 2));
        HIP_TRY(hipMalloc(&P.list, sizeof(unsigned int) * 2 * AC));
        HIP_TRY(hipMalloc(&P.refill, sizeof(unsigned int) * AC));
        HIP_TRY(hipMalloc(&P.cont, sizeof(unsigned int) * (AC + ENG_MAX_POOLS * NN_CONT_SHARDS)));
        HIP_TRY(hipMemsetAsync(P.cont, 0, sizeof(unsigned int) * (AC + ENG_MAX_POOLS * NN_CONT_SHARDS), stream));
        HIP_TRY(hipMalloc(&P.order, sizeof(unsigned int) * 2 * NN_CLASSES * AC));
        HIP_TRY(hipMalloc(&P.carry, sizeof(double) * NN_CARRY * AC));
        HIP_TRY(hipMalloc(&P.compact, sizeof(unsigned int) * (2 * ENG_MAX_POOLS + 2 * AC)));
        P.C = AC;
        // first touch of the big buffers here, not in the first run that uses the slots (a
        // warm-up of one step followed by a 5-step run paid ~10 ms for it inside the timed run)
        HIP_TRY(hipMemsetAsync(P.d, 0, sizeof(double) * (size_t)((nd + 1) / 2 * 2) * ((AC + 63) / 64 * 64), stream));
        HIP_TRY(hipMemsetAsync(P.i32, 0, sizeof(int32_t) * (size_t)ni * AC, stream));
        HIP_TRY(hipMemsetAsync(P.item, 0, sizeof(unsigned long long) * AC, stream));
        HIP_TRY(hipMemsetAsync(P.prob, 0, sizeof(double) * AC * rec_len, stream));
        HIP_TRY(hipMemsetAsync(P.y, 0, sizeof(double) * AC * (2 * nn), stream));
        HIP_TRY(hipMemsetAsync(P.meta, 0, sizeof(double) * AC * 2, stream));
        HIP_TRY(hipMemsetAsync(P.order, 0, sizeof(unsigned int) * 2 * NN_CLASSES * AC, stream));
        HIP_TRY(hipMemsetAsync(P.carry, 0, sizeof(double) * NN_CARRY * AC, stream));
    }
    if (!ch->eng_pool_attached) {
        std::lock_guard<std::mutex> g(g_eng_pools_mu);
        P.users += 1;
        ch->eng_pool_attached = true;
    }
    // the chain's view of the pool (refreshed on every run: another chain may have grown it)
    ch->eng_C = P.C;
    ch->eng_d = P.d; ch->eng_i32 = P.i32; ch->eng_item = P.item; ch->eng_prob = P.prob; ch->eng_y = P.y;
    ch->eng_meta = P.meta; ch->eng_carry = P.carry; ch->eng_list = P.list; ch->eng_refill = P.refill;
    ch->eng_cont = P.cont; ch->eng_order = P.order; ch->eng_compact = P.compact;
    if (!ch->eng_djobs) HIP_TRY(hipMalloc(&ch->eng_djobs, sizeof(EngJob) * ENG_MAX_JOBS));
    constexpr int PCB = ENG_POOL_COUNTERS;
    if (!ch->eng_counters) HIP_TRY(hipMalloc(&ch->eng_counters, ENG_MAX_POOLS * PCB * sizeof(unsigned int)));
    if (!ch->eng_pinned) HIP_TRY(hipHostMalloc(&ch->eng_pinned, ENG_MAX_POOLS * 8 * sizeof(unsigned int)));
    if (!ch->queue) HIP_TRY(hipMalloc(&ch->queue, sizeof(unsigned long long)));
    if (!ch->eng_nn_total) HIP_TRY(hipMalloc(&ch->eng_nn_total, sizeof(unsigned long long) * (1 + ENG_EXEC_SHARDS)));
    HIP_TRY(hipMemsetAsync(ch->eng_nn_total, 0, sizeof(unsigned long long) * (1 + ENG_EXEC_SHARDS), stream));
    for (auto &pe : ch->eng_pool_ev) for (auto &e : pe) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!ch->eng_fork_ev) HIP_TRY(hipEventCreateWithFlags(&ch->eng_fork_ev, hipEventDisableTiming));
    for (int p2 = 1; p2 < ENG_MAX_POOLS; ++p2) {
        if (!ch->eng_streams[p2]) {
            HIP_TRY(hipStreamCreateWithFlags(&ch->eng_streams[p2], hipStreamNonBlocking));
        }
        if (!ch->eng_join_ev[p2]) HIP_TRY(hipEventCreateWithFlags(&ch->eng_join_ev[p2], hipEventDisableTiming));
    }
    return 0;
}

int optik_hip_engine_run(optik_hip_chain *ch, void *stream_v) { return optik_hip_engine_run_ex(ch, stream_v, 0.0); }

int optik_hip_engine_run_ex(optik_hip_chain *ch, void *stream_v, double deadline_s) {
    if (!ch) return fail(OPTIK_HIP_EINVAL, "bad argument");
    hipStream_t stream = (hipStream_t)stream_v;
    std::lock_guard<std::mutex> lock(ch->mu);
    BIND_DEVICE(ch);
    if (ch->eng_jobs.empty()) return 0;
    if (ch->claim_pending && stream != nullptr) HIP_TRY(hipStreamSynchronize(nullptr));  // (see ik_batch_locked)
    ch->claim_pending = false;
    const auto t_call = std::chrono::steady_clock::now();
    auto since_call = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call).count(); };
    int rc = 0;
    if (ch->n > 7) {
        // the engine's register-resident NNLS holds n + 1 <= 8 rows: an 8-DoF chain's jobs run
        // one after the other on the single-kernel path (same restarts, same results)
        for (auto &j : ch->eng_jobs) {
            if (rc) break;
            optik_solver_config cfg = ch->eng_cfg;
            cfg.solution_mode = j.dev.quality ? 1 : 2;
            double left = 0.0;
            if (deadline_s > 0.0) { left = deadline_s - since_call(); if (left <= 0.0) left = 1e-9; }
            rc = ik_batch_locked(ch, &cfg, j.dev.targets, j.dev.x0, j.T, ch->eng_has_ee ? ch->eng_ee : nullptr,
                                 j.dev.restart_begin, j.dev.restart_begin + j.dev.n_restarts,
                                 (j.own_fs ? OPTIK_HIP_IK_EARLY_EXIT : 0u) | (j.dev.find_any ? OPTIK_HIP_IK_FIND_ANY : 0u)
                                     | ((j.dev.restart_major && !j.full_pool) ? OPTIK_HIP_IK_RESTART_MAJOR : 0u), left,
                                 &j.out, stream);
            if (!rc && hipStreamSynchronize(stream) != hipSuccess) rc = fail(OPTIK_HIP_ENODEVICE, "engine job failed");
        }
        for (auto &j : ch->eng_jobs) {
            if (j.own_x) (void)hipFree(j.own_x);
            if (j.own_f) (void)hipFree(j.own_f);
            if (j.own_key) (void)hipFree(j.own_key);
            if (j.own_fs) (void)hipFree(j.own_fs);
        }
        ch->eng_jobs.clear();
        return rc;
    }
    // the device's slot pool is this run's from here to its last kernel (the runs of other chains of
    // the device wait: ik_engine.hpp's kernels fill the chip on their own)
    std::lock_guard<std::mutex> pool_lock(engine_pool_of(ch).run_mu);
    // at most ENG_MAX_JOBS jobs share one run of the pool; more are executed as consecutive runs
    for (size_t g0 = 0; g0 < ch->eng_jobs.size() && rc == 0; g0 += ENG_MAX_JOBS) {
    const size_t g1 = std::min(ch->eng_jobs.size(), g0 + (size_t)ENG_MAX_JOBS);
    const size_t n_jobs = g1 - g0;
    const unsigned long long base0 = ch->eng_jobs[g0].dev.item_base;
    const unsigned long long total = ch->eng_jobs[g1 - 1].dev.item_base + ch->eng_jobs[g1 - 1].dev.n_items - base0;

    // pool size: enough slots for every CU to hold several waves of each phase kernel
    size_t cap = 393216;  // 3 sub-pools x 131072 slots = 2048 waves: one full round of the chip (2 waves per SIMD) per kernel; 417792 is 6 % slower
    if (opt().engine_slots >= 256) cap = (size_t)opt().engine_slots;
    size_t C = (size_t)((total + 255ull) / 256ull * 256ull);
    if (C > cap) C = cap;
    {
        // Speed jobs with early exit abandon most restarts above the first success: a pool of a
        // few restarts per target keeps the phase kernels small (their cost is per slot scanned)
        bool all_early = true;
        unsigned long long targets = 0;
        for (size_t ji = g0; ji < g1; ++ji) {
            const auto &j = ch->eng_jobs[ji];
            all_early = all_early && j.own_fs != nullptr && !j.full_pool;
            targets += (unsigned long long)j.T;
        }
        if (all_early && opt().engine_slots < 256) {
            // (many targets: every slot beyond one per target runs a restart that is abandoned if an
            // earlier one of its target succeeds -- measured best, 16 indices per target: 196 608
            // slots at 65 536 targets (17.4 against 21.0 ms with 393 216) and at 131 072 (26.4 against
            // 27.7), 262 144 at 262 144 (39.4 against 40.3))
            size_t want = (size_t)((targets * 8ull + 255ull) / 256ull * 256ull);
            if (targets >= 32768) want = (size_t)(targets > 196608 ? (targets + 255ull) / 256ull * 256ull : 196608);
            if (want < 16384) want = 16384;
            if (want < C) C = want;
        }
    }
    const auto dbg_entry = std::chrono::steady_clock::now();
    auto run = [&]() -> int {
        // M(NN) is a statement macro instantiated for the chain's n
#define DISPATCH_N(M)                                                                            \
    do {                                                                                         \
        switch (ch->n) {                                                                         \
        case 1: M(1); break;                                                                     \
        case 2: M(2); break; case 3: M(3); break; case 4: M(4); break;                           \
        case 5: M(5); break; case 6: M(6); break; case 7: M(7); break;                           \
        default: return fail(OPTIK_HIP_EUNSUPPORTED, OPTIK_N_RANGE_MSG);                         \
        }                                                                                        \
    } while (0)
        int nd = 0, ni = 0, rec_len = 0;
#define M_LAYOUT(NN) nd = EngLayout<NN>::ND; ni = EngLayout<NN>::NI; rec_len = rec_stride<NN>()
        DISPATCH_N(M_LAYOUT);
#undef M_LAYOUT
        {
            // a run of bench size reserves the full capacity at once: the next, larger run (a
            // warm-up followed by the timed run) does not reallocate ~1 GB inside its timed region
            const int arc = engine_reserve(ch, (C >= 65536 && cap > C) ? cap : C, nd, ni, rec_len, stream);
            if (arc) return arc;
        }
        constexpr int PCB = ENG_POOL_COUNTERS;
        ch->eng_tcount = 0;

        std::vector<EngJob> hj(n_jobs);
        for (size_t i = 0; i < n_jobs; ++i) { hj[i] = ch->eng_jobs[g0 + i].dev; hj[i].item_base -= base0; }
        HIP_TRY(hipMemcpyAsync(ch->eng_djobs, hj.data(), sizeof(EngJob) * n_jobs, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));  // hj goes out of scope; tiny copy
        HIP_TRY(hipMemsetAsync(ch->queue, 0, sizeof(unsigned long long), stream));
        ch->queue_clean = false;  // (the engine's kernels leave the counter where the run ended)
        for (size_t ji = g0; ji < g1; ++ji) {
            auto &j = ch->eng_jobs[ji];
            if (j.own_fs) {
                HIP_TRY(hipMemsetAsync(j.own_fs, 0xff, sizeof(unsigned long long) * (size_t)j.T, stream));
                // restarts abandoned before they start never touch their outputs: key = no solution
                if (j.dev.out_key)
                    hipLaunchKernelGGL(fill_f64_kernel, dim3(1024), dim3(256), 0, stream, j.dev.out_key,
                                       (unsigned long long)j.dev.n_items, __builtin_huge_val());
            }
        }

        EngArgs a;
        std::memset(&a, 0, sizeof a);
        a.chain = ch->dev;
        make_eval_params(ch->eng_cfg.linear_weight, ch->eng_cfg.angular_weight, ch->eng_has_ee ? ch->eng_ee : nullptr, a.ep);
        fill_solve_params(&ch->eng_cfg, a.sp, opt().stop_x_legacy != 0);
        std::memcpy(a.key, ch->key, sizeof a.key);
        std::memcpy(a.scale, ch->scale, sizeof a.scale);
        a.d = ch->eng_d; a.i32 = ch->eng_i32; a.item = ch->eng_item; a.C = C;
        a.jobs = ch->eng_djobs; a.n_jobs = (int)n_jobs;
        a.total_items = total;
        a.next_item = ch->queue;
        a.nn_carry = ch->eng_carry;
        a.nn_prob = ch->eng_prob;
        a.nn_y = ch->eng_y;
        a.nn_meta = ch->eng_meta;
        a.nn_budget = opt().engine_nnls_budget > 0 ? opt().engine_nnls_budget : 1;
        a.nn_pred_viol = 1;
        // per-problem pass cap = predicted count + slack.  With the same-trip continuation launch
        // the cap sits at the prediction itself (slack 0): the problems that need more continue a
        // few microseconds later instead of holding their wave's other 15 (without it: no cap
        // 23.9, slack 1 -> 24.6, slack 0 -> 23.2 M restarts/s -- a suspended solve cost its slot a trip)
        const bool nn_cont = false;  // (measured r2: 23.5 M restarts/s with the continuation launch against 26.4 M without -- the extra launch on each trip's critical path costs more than the tighter cap saves)
        a.nn_slack = opt().engine_nnls_slack > 0 ? opt().engine_nnls_slack : 0;
        a.cont_pass = 0;
        a.cont_count = nullptr; a.cont_list = nullptr; a.cont_cap = 0;
        const unsigned cont_waves_per_cu = 4;  // grid of the continuation launch (it grid-strides over the lists)
        a.nn_total = ch->eng_nn_total;
        a.exec_evals = ch->eng_nn_total + 1;
        a.tail_deadline_ticks = 0;
        a.abort = 0;
        a.deadline_word = nullptr;
        if (deadline_s > 0.0) {
            // max_time on the device clock: the kernels stop what is in flight within a trip of its expiry
            if (!ch->eng_deadline) HIP_TRY(hipMalloc(&ch->eng_deadline, sizeof(unsigned long long)));
            const double left_s = deadline_s - since_call();
            const double khz = ch->wall_clock_khz > 0 ? (double)ch->wall_clock_khz : 100000.0;
            const unsigned long long ticks = left_s > 0.0 ? (unsigned long long)(left_s * khz * 1e3) + 1ull : 1ull;
            hipLaunchKernelGGL(eng_deadline_kernel, dim3(1), dim3(1), 0, stream, ch->eng_deadline, ticks);
            HIP_TRY(hipGetLastError());  // (a failed launch would leave the previous run's deadline in the word)
            a.deadline_word = ch->eng_deadline;
        }
        a.parity = 0;
        a.prof = nullptr;
        a.prof2 = nullptr;
        a.trace = nullptr;
        a.trip_log = nullptr;
        a.trip = 0;
#ifdef OPTIK_PROFILE
        const char *trip_log_path = std::getenv("OPTIK_ENG_TRIP_LOG");  // (diagnostic builds: per-trip in-use counts of sub-pool 0)
#else
        const char *trip_log_path = nullptr;
#endif
        constexpr int TRIP_LOG_MAX = 65536;
        if (trip_log_path) {
            if (!ch->eng_trip_log) HIP_TRY(hipMalloc(&ch->eng_trip_log, sizeof(unsigned int) * 2 * TRIP_LOG_MAX));
            HIP_TRY(hipMemsetAsync(ch->eng_trip_log, 0, sizeof(unsigned int) * 2 * TRIP_LOG_MAX, stream));
            a.trip_log = ch->eng_trip_log;
        }
#ifdef OPTIK_PROFILE
        if (!ch->prof) HIP_TRY(hipMalloc(&ch->prof, 8 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(ch->prof, 0, 8 * sizeof(unsigned long long), stream));
        a.prof = ch->prof;
        a.prof2 = ch->prof + 5;  // slots 5, 6 + (7 is the wave count: sub-phase 2 is derived)
#endif

        ch->eng_compactions = 0;
        const bool allow_compact = opt().engine_compact != 0;
        const int cus = ch->num_cus > 0 ? ch->num_cus : 256;
        const unsigned nn_waves_per_cu = 32;
        const unsigned nn_blocks = (unsigned)cus * nn_waves_per_cu * 64u / OPTIK_ENG_NNLS_BLOCK;
        HIP_TRY(hipMemsetAsync(ch->eng_counters, 0, ENG_MAX_POOLS * PCB * sizeof(unsigned int), stream));
        HIP_TRY(hipMemsetAsync(ch->eng_cont + ch->eng_C, 0, ENG_MAX_POOLS * NN_CONT_SHARDS * sizeof(unsigned int), stream));
        hipLaunchKernelGGL(eng_init_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, ch->eng_i32, ch->eng_list, (unsigned long long)ch->eng_C);
        HIP_TRY(hipGetLastError());

        // sub-pools: equal slot ranges (multiples of 256), each with its own stream and lists
        int n_pools = 3;  // (measured on MI355X: 1 -> 12.6, 2 -> 14.0, 3 -> 14.6, 4 -> 13.9 M restarts/s)
        size_t min_pool = 16384;  // small pools: one trip loop
        if (opt().engine_pools > 0) { n_pools = opt().engine_pools; min_pool = 1024; }
        if (n_pools < 1) n_pools = 1;
        if (n_pools > ENG_MAX_POOLS) n_pools = ENG_MAX_POOLS;
        while (n_pools > 1 && C / (size_t)n_pools < min_pool) --n_pools;
        ch->eng_pools = n_pools;
        struct Pool {
            EngArgs a;
            hipStream_t stream;
            unsigned blocks;
            int trip, pending, ring;
            bool done;
            unsigned int *pinned;
            hipEvent_t *ev;
            unsigned int *compact_counts;
            unsigned long long last_in_use;  // latest in-use count the host has seen
            unsigned long long initial_size;
            bool drained;                    // a count below the sub-pool's size was seen: the queue is empty
        };
        Pool pools[ENG_MAX_POOLS];
        {
            const size_t per = C / 256 / (size_t)n_pools * 256;
            for (int p2 = 0; p2 < n_pools; ++p2) {
                Pool &P = pools[p2];
                const size_t lo = per * (size_t)p2;
                const size_t size = (p2 == n_pools - 1) ? C - lo : per;
                P.a = a;
                P.a.slot_base = lo;
                P.a.n_slots = size;
                unsigned int *cnt = ch->eng_counters + (size_t)p2 * PCB;
                for (int par = 0; par < 2; ++par) {
                    P.a.nn_class_count[par] = cnt + par * NN_CLASSES;
                    P.a.nn_order[par] = ch->eng_order + (size_t)par * NN_CLASSES * ch->eng_C + lo;  // [class][C] + lo
                    P.a.nn_cls[par] = ch->eng_list + (size_t)par * ch->eng_C + lo;
                }
                P.a.refill_count = cnt + 2 * NN_CLASSES;
                P.a.n_active = cnt + 2 * NN_CLASSES + 1;
                P.a.refill_list = ch->eng_refill + lo;
                if (nn_cont) {
                    P.a.cont_count = ch->eng_cont + ch->eng_C + (size_t)p2 * NN_CONT_SHARDS;
                    P.a.cont_list = ch->eng_cont + lo;
                    P.a.cont_cap = (unsigned)(size / NN_CONT_SHARDS);
                }
                P.stream = p2 == 0 ? stream : ch->eng_streams[p2];
                P.blocks = (unsigned)((size + OPTIK_ENG_SLOT_BLOCK - 1) / OPTIK_ENG_SLOT_BLOCK);
                P.trip = 0; P.pending = 0; P.ring = 0; P.done = false;
                P.last_in_use = size; P.drained = false; P.initial_size = size;
                P.pinned = ch->eng_pinned + 8 * p2;
                P.ev = ch->eng_pool_ev[p2];
                P.compact_counts = ch->eng_compact + 2 * p2;
                if (p2 > 0) { P.a.trip_log = nullptr; P.a.prof = nullptr; }
            }
            // the other streams start after the set-up queued on the caller's stream
            HIP_TRY(hipEventRecord(ch->eng_fork_ev, stream));
            for (int p2 = 1; p2 < n_pools; ++p2) HIP_TRY(hipStreamWaitEvent(ch->eng_streams[p2], ch->eng_fork_ev, 0));
        }

#ifdef OPTIK_ENG_DEBUG
        constexpr bool ENG_DEBUG_PRINTS = true;   // (diagnostic builds: where an engine run's host time goes)
#else
        constexpr bool ENG_DEBUG_PRINTS = false;
#endif
        const int CHECK = 4;  // trips between termination checks
        const int check_drain = CHECK;  // ... of a sub-pool whose queue has run dry
        const int queue_depth = 2;  // chunks queued ahead of the one whose in-use count the host waits for
        const bool tip = ch->tip;
        double dbg_wait_bulk = 0.0, dbg_wait_drain = 0.0;  // host time blocked on the GPU (OPTIK_ENG_DEBUG)
        double dbg_drain_t0 = -1.0;  // when the first sub-pool fell under half of its live prefix
        const auto dbg_t0 = std::chrono::steady_clock::now();
        if (ENG_DEBUG_PRINTS)
            fprintf(stderr, "[optik engine] set-up %.2f ms (C = %zu)\n", std::chrono::duration<double>(dbg_t0 - dbg_entry).count() * 1e3, (size_t)C);
        // queues CHECK trips of one sub-pool, then looks at the in-use count of its previous chunk
        const size_t eval_lds = sizeof(EngJob) * n_jobs;  // the eval kernel's copy of the job table
        bool aborted = false;
        auto advance = [&](Pool &P, bool first_pool) -> int {
            EngArgs &a = P.a;
            hipStream_t stream = P.stream;
            unsigned &blocks = P.blocks;
            int &trip = P.trip;
            // max_time (lib.rs:260-264, 308): once it has expired every trip queued from here on
            // abandons what is in flight and drains the queue without starting anything
            if (deadline_s > 0.0 && !aborted && since_call() > deadline_s) aborted = true;
            a.abort = aborted ? 1 : 0;
            // (a draining sub-pool reports more often: the hand-over to the tail is decided on those counts)
            const int check = P.drained ? check_drain : CHECK;
            for (int k = 0; k < check; ++k, ++trip) {
                // this trip consumes list[trip & 1]; the other list (consumed last trip) is
                // reset for the finish kernel's re-deferrals and the next trip's update kernel
                a.parity = trip & 1;
                a.trip = trip < TRIP_LOG_MAX ? trip : TRIP_LOG_MAX - 1;
                a.host_in_use = (k == check - 1) ? &P.pinned[P.ring] : nullptr;
                // start / stop events on the kernels of sub-pool 0 (on its launch stream)
                const bool timed = first_pool && ch->timing && trip > 0 && ch->eng_tcount < optik_hip_chain::ENG_EV;
                const int ts = ch->eng_tcount;
                // (the NNLS kernel -- the longest -- is timed on every trip, the others on every 8th: a
                // trip in the middle of a chunk, not the one the host queues right after its wait)
                const bool timed_all = timed && (trip % 8 == 2);
                // The events are attached to the kernel dispatch itself (hipExtLaunchKernelGGL): they
                // carry the dispatch's own start and end timestamps -- what rocprofv3's kernel trace
                // reports -- not the time the launch waits for CUs behind the other sub-pools' kernels.
                hipEvent_t tev0 = nullptr, tev1 = nullptr;
#define TEV(kind) do { tev0 = tev1 = nullptr; if (timed && (kind == 2 || timed_all)) {                                   \
                        for (int w_ = 0; w_ < 2; ++w_) { hipEvent_t &tev_slot_ = ch->eng_tev[kind][ts][w_]; if (!tev_slot_) HIP_TRY(hipEventCreate(&tev_slot_)); } \
                        tev0 = ch->eng_tev[kind][ts][0]; tev1 = ch->eng_tev[kind][ts][1]; } } while (0)
#define ENG_LAUNCH_LDS(kernel, grid, block, lds) do { if (tev0) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, tev0, tev1, 0, a); \
                                                       else hipLaunchKernelGGL(kernel, grid, block, lds, stream, a); } while (0)
#define ENG_LAUNCH(kernel, grid, block) ENG_LAUNCH_LDS(kernel, grid, block, 0)
                TEV(0);
                if (trip > 0) {
#define M_EVAL_T(NN) ENG_LAUNCH_LDS((eng_eval_kernel<NN, true>), dim3(blocks), dim3(OPTIK_ENG_SLOT_BLOCK), eval_lds)
#define M_EVAL_F(NN) ENG_LAUNCH_LDS((eng_eval_kernel<NN, false>), dim3(blocks), dim3(OPTIK_ENG_SLOT_BLOCK), eval_lds)
                    if (tip) DISPATCH_N(M_EVAL_T);
                    else DISPATCH_N(M_EVAL_F);
#undef M_EVAL_T
#undef M_EVAL_F
                }
                TEV(1);
#define M_UPD(NN) ENG_LAUNCH((eng_update_kernel<NN>), dim3(blocks * (OPTIK_ENG_SLOT_BLOCK / OPTIK_ENG_UPD_BLOCK)), dim3(OPTIK_ENG_UPD_BLOCK))
                DISPATCH_N(M_UPD);
#undef M_UPD
#ifdef OPTIK_NNLS_TRACE
                if (first_pool) {   // per-wave timeline of one steady-state trip (debug builds only)
                    static unsigned long long *tr = nullptr;
                    const int t0 = getenv("OPTIK_NNLS_TRACE_TRIP") ? atoi(getenv("OPTIK_NNLS_TRACE_TRIP")) : 150;
                    if (!tr) { HIP_TRY(hipMalloc(&tr, sizeof(unsigned long long) * 4 * 65536)); HIP_TRY(hipMemset(tr, 0, sizeof(unsigned long long) * 4 * 65536)); }
                    a.trace = (trip == t0) ? tr : nullptr;
                    ch->nnls_trace = tr;
                }
#endif
                hipLaunchKernelGGL(eng_bucket_kernel, dim3((unsigned)cus), dim3(256), 0, stream, a);
                TEV(2);
#define M_NNLS(NN) ENG_LAUNCH((eng_nnls_coop_kernel<NN>), dim3(nn_blocks), dim3(OPTIK_ENG_NNLS_BLOCK))
                DISPATCH_N(M_NNLS);
#undef M_NNLS
                if (nn_cont) {
                    // the problems the main launch suspended at their predicted pass count
                    tev0 = tev1 = nullptr;
                    a.cont_pass = 1;
                    const unsigned cont_blocks = (unsigned)cus * cont_waves_per_cu * 64u / OPTIK_ENG_NNLS_BLOCK;
#define M_NNLS_C(NN) ENG_LAUNCH((eng_nnls_coop_kernel<NN>), dim3(cont_blocks), dim3(OPTIK_ENG_NNLS_BLOCK))
                    DISPATCH_N(M_NNLS_C);
#undef M_NNLS_C
                    a.cont_pass = 0;
                }
                TEV(3);
#define M_FIN(NN) ENG_LAUNCH((eng_finish_kernel<NN>), dim3(blocks * (OPTIK_ENG_SLOT_BLOCK / OPTIK_ENG_UPD_BLOCK)), dim3(OPTIK_ENG_UPD_BLOCK))
                DISPATCH_N(M_FIN);
#undef M_FIN
#undef TEV
#undef ENG_LAUNCH
#undef ENG_LAUNCH_LDS
                if (timed) ch->eng_tcount += 1;
                ch->eng_launches += 1;
            }
            HIP_TRY(hipGetLastError());
            // the chunk's last finish kernel wrote its in-use count to pinned memory; the host looks
            // at the previous chunk's value
            HIP_TRY(hipEventRecord(P.ev[P.ring], stream));
            // The host keeps `depth` chunks queued behind the one it waits for: with a single
            // chunk in flight (wait for chunk c - 1 right after queuing chunk c) every stream ran
            // dry for 0.3 - 0.6 ms each fourth trip while the host sat in the other sub-pools'
            // waits (rocprofv3 kernel trace, profiles/r2a_*: a quarter of each stream's time).
            // Once the sub-pool drains the lag is cut back to one chunk (trips are short then).
            const int depth = P.drained ? 1 : queue_depth;
            if (P.pending >= depth) {
                const int prev = (P.ring + 8 - depth) % 8;
                const auto w0 = std::chrono::steady_clock::now();
                HIP_TRY(hipEventSynchronize(P.ev[prev]));
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
                const unsigned long long in_use = P.pinned[prev];
                (in_use * 2 < a.n_slots || a.n_slots < 16384 ? dbg_wait_drain : dbg_wait_bulk) += waited;
                if (dbg_drain_t0 < 0.0 && in_use * 2 < P.a.n_slots) dbg_drain_t0 = std::chrono::duration<double>(std::chrono::steady_clock::now() - dbg_t0).count();
                P.last_in_use = in_use;
                if (in_use < P.initial_size) P.drained = true;
                if (in_use == 0) P.done = true;
                // drain: part of the live prefix no longer holds a restart (the count only
                // falls once the queue is empty, so the lagging value is an upper bound)
                unsigned long long n_new = (in_use + 255) / 256 * 256;
                if (n_new < 1024) n_new = 1024;
                if (!P.done && allow_compact && n_new * 5 <= a.n_slots * 4) {  // worth >= 20% of every per-slot launch
                    CompactArgs c;
                    c.d = ch->eng_d; c.i32 = ch->eng_i32; c.item = ch->eng_item;
                    c.C = C; c.slot_base = a.slot_base; c.n_slots = a.n_slots; c.n_new = n_new; c.nd = nd; c.ni = ni;
                    c.counts = P.compact_counts;
                    c.free_list = ch->eng_compact + 2 * ENG_MAX_POOLS + a.slot_base;
                    c.move_list = ch->eng_compact + 2 * ENG_MAX_POOLS + C + a.slot_base;
                    c.nn_prob = ch->eng_prob; c.nn_meta = ch->eng_meta; c.nn_carry = ch->eng_carry;
                    c.nn_cls = a.nn_cls[trip & 1];  // the entries the next trip consumes
                    c.rec_len = rec_len;
                    c.nn_y = nullptr;
                    c.ny = 2 * ch->n;
                    HIP_TRY(hipMemsetAsync(P.compact_counts, 0, 2 * sizeof(unsigned int), stream));
                    hipLaunchKernelGGL(eng_compact_scan_kernel, dim3((unsigned)((a.n_slots + 255) / 256)), dim3(256), 0, stream, c);
                    hipLaunchKernelGGL(eng_compact_move_kernel, dim3((unsigned)((a.n_slots - n_new + 255) / 256)), dim3(256), 0, stream, c);
                    HIP_TRY(hipGetLastError());
                    a.n_slots = n_new;
                    blocks = (unsigned)((n_new + OPTIK_ENG_SLOT_BLOCK - 1) / OPTIK_ENG_SLOT_BLOCK);
                    ch->eng_compactions += 1;
                }
            }
            if (P.pending < depth) P.pending += 1;
            else if (P.pending > depth) P.pending = depth;
            P.ring = (P.ring + 1) % 8;
            return 0;
        };
        ch->eng_launches = 0;
        // hand-over to the tail kernel -- the quad solver fed from the slot pool, ik_quad_tail.hpp -- once this few
        // restarts are left in all sub-pools together (it runs a restart ~2.7 times as fast as a pool trip that
        // is mostly empty, so it takes over early)
        unsigned long long tail_max = total / 8, tail_cap = 131072ull;
        if (tail_max > tail_cap) tail_max = tail_cap;
        if (tail_max < 64) tail_max = 64;
        if (opt().engine_tail_max >= 0) tail_max = (unsigned long long)opt().engine_tail_max;
        ch->eng_tail_restarts = 0;
        ch->eng_tail_solver = 0;
        for (bool all_done = false; !all_done;) {
            all_done = true;
            for (int p2 = 0; p2 < n_pools; ++p2) {
                if (pools[p2].done) continue;
                const int prc = advance(pools[p2], p2 == 0);
                if (prc != 0) return prc;
                all_done = all_done && pools[p2].done;
            }
            if (all_done || tail_max == 0 || aborted) continue;
            // every sub-pool still running has reported a count since the queue ran dry?
            unsigned long long left = 0;
            bool known = true;
            for (int p2 = 0; p2 < n_pools; ++p2) {
                const Pool &P = pools[p2];
                if (P.done) continue;
                known = known && P.drained;
                left += P.last_in_use;
            }
            if (!known || left > tail_max) continue;
            // the trip loops stop here; everything they queued precedes the tail on the caller's stream
            for (int p2 = 1; p2 < n_pools; ++p2) {
                if (pools[p2].done) continue;
                HIP_TRY(hipEventRecord(ch->eng_join_ev[p2], pools[p2].stream));
                HIP_TRY(hipStreamWaitEvent(stream, ch->eng_join_ev[p2], 0));
            }
            unsigned int *t_count = ch->eng_compact;          // (the compaction scratch is free from here on)
            unsigned int *t_list = ch->eng_refill;            // [C]
            HIP_TRY(hipMemsetAsync(t_count, 0, sizeof(unsigned int), stream));
            hipLaunchKernelGGL(eng_tail_list_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream,
                               ch->eng_i32, (unsigned long long)C, t_count, t_list);
            {
                // persistent waves of the quad solver, each quad pulling the next live slot of the list
                TailLaunch tq;
                std::memset(&tq, 0, sizeof tq);
                const EngArgs &pa0 = pools[0].a;
                tq.base.chain = pa0.chain;
                tq.base.ep = pa0.ep;
                tq.base.sp = pa0.sp;
                long long capq = (long long)cus * quad_solve_waves_per_cu(ch->n);
                long long lq = ((long long)left + capq - 1) / capq;
                if (lq < 1) lq = 1;
                if (lq > QUADS_PER_WAVE_HOST) lq = QUADS_PER_WAVE_HOST;
                tq.base.wq.lanes = (int)lq;
                long long gq = ((long long)left + lq - 1) / lq;
                if (gq > capq) gq = capq;
                if (gq < 1) gq = 1;
                if (deadline_s > 0.0) {
                    const double left_s = deadline_s - since_call();
                    const double khz = ch->wall_clock_khz > 0 ? (double)ch->wall_clock_khz : 100000.0;
                    tq.base.deadline_ticks = left_s > 0.0 ? (unsigned long long)(left_s * khz * 1e3) + 1ull : 1ull;
                }
                tq.tail.d = pa0.d; tq.tail.i32 = pa0.i32; tq.tail.item = pa0.item; tq.tail.C = pa0.C;
                tq.tail.jobs = pa0.jobs;
                tq.tail.list = t_list;
                tq.tail.count = t_count;
                tq.tail.cursor = reinterpret_cast<unsigned long long *>(t_count + 2);  // (8-byte aligned word of the same scratch)
                tq.tail.exec_evals = pa0.exec_evals;
                HIP_TRY(hipMemsetAsync(t_count + 2, 0, sizeof(unsigned long long), stream));
                HIP_TRY(quad_tail_launch(ch->n, tip, (int)gq, stream, tq));
                ch->eng_tail_restarts = (int)left;
                ch->eng_tail_solver = 3;
                for (int p2 = 0; p2 < n_pools; ++p2) pools[p2].done = true;
                all_done = true;
            }
        }
        if (ENG_DEBUG_PRINTS)
            fprintf(stderr, "[optik engine] loop %.2f ms (drain from %.2f ms), host waited on the GPU %.2f ms (bulk) + %.2f ms (drain), %d launches\n",
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - dbg_t0).count() * 1e3, dbg_drain_t0 * 1e3,
                    dbg_wait_bulk * 1e3, dbg_wait_drain * 1e3, ch->eng_launches);
        // the selection below runs on the caller's stream after every sub-pool
        for (int p2 = 1; p2 < n_pools; ++p2) {
            HIP_TRY(hipEventRecord(ch->eng_join_ev[p2], ch->eng_streams[p2]));
            HIP_TRY(hipStreamWaitEvent(stream, ch->eng_join_ev[p2], 0));
        }
        const int trip = pools[0].trip;
        ch->eng_trips = trip;

        // selection of every job (lib.rs:397-413)
        constexpr int SEL_TILE = 4096;
        for (size_t ji = g0; ji < g1; ++ji) {
            auto &j = ch->eng_jobs[ji];
            const optik_hip_ik_outputs &o = j.out;
            if (!(o.d_win_x || o.d_win_f || o.d_win_idx || o.d_win_key)) continue;
            const uint64_t R = j.dev.n_restarts;
            const uint64_t tiles_per_target = (R + SEL_TILE - 1) / SEL_TILE;
            const size_t n_tiles = (size_t)tiles_per_target * (size_t)j.T;
            if (n_tiles > ch->tile_cap) {
                HIP_TRY(hipStreamSynchronize(stream));
                if (ch->tile_recs) HIP_TRY(hipFree(ch->tile_recs));
                ch->tile_recs = nullptr;
                HIP_TRY(hipMalloc(&ch->tile_recs, sizeof(TileRec) * n_tiles));
                ch->tile_cap = n_tiles;
            }
            SelectLaunch s;
            std::memset(&s, 0, sizeof s);
            s.out_key = j.dev.out_key; s.out_x = j.dev.out_x; s.out_f = j.dev.out_f;
            s.tile_recs = ch->tile_recs;
            s.tiles_per_target = (int)tiles_per_target;
            s.tile = SEL_TILE;
            s.n = ch->n;
            s.restart_begin = j.dev.restart_begin;
            s.n_restarts = R;
            s.ld = (size_t)j.dev.n_items;
            s.win_x = o.d_win_x; s.win_f = o.d_win_f;
            s.win_idx = (unsigned long long *)o.d_win_idx; s.win_key = o.d_win_key;
            hipLaunchKernelGGL(ik_tile_argmin_kernel, dim3((unsigned)n_tiles), dim3(256), 0, stream, s);
            hipLaunchKernelGGL(ik_select_kernel, dim3(j.T), dim3(WAVE), 0, stream, s);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipStreamSynchronize(stream));
        if (ENG_DEBUG_PRINTS)
            fprintf(stderr, "[optik engine] run complete %.2f ms after the loop started (tail kernel took over <= %d restarts)\n",
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - dbg_t0).count() * 1e3, ch->eng_tail_restarts);
        for (int k = 0; k < 4; ++k) {
            double sum = 0.0;
            int cnt = 0;
            for (int i = 0; i < ch->eng_tcount; ++i) {
                const int trip_i = i + 1;  // sample i was taken on trip i + 1
                if (k != 2 && trip_i % 8 != 2) continue;
                float ms = 0.0f;
                HIP_TRY(hipEventElapsedTime(&ms, ch->eng_tev[k][i][0], ch->eng_tev[k][i][1]));
                sum += ms;
                ++cnt;
            }
            ch->eng_kernel_ms[k] = cnt ? sum / cnt : 0.0;
        }
        if (trip_log_path) {
            // per trip: slots in use, sub-problems, and the sampled kernel times (ms; 0 = not sampled)
            std::vector<unsigned int> h(2 * (size_t)std::min(trip, TRIP_LOG_MAX));
            HIP_TRY(hipMemcpy(h.data(), ch->eng_trip_log, h.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
            if (FILE *fp = fopen(trip_log_path, "w")) {
                fprintf(fp, "trip,slots_in_use,sub_problems,eval_ms,update_ms,nnls_ms,finish_ms\n");
                for (int t = 0; t < (int)(h.size() / 2); ++t) {
                    float ms[4] = {0, 0, 0, 0};
                    const int i = t - 1;
                    if (i >= 0 && i < ch->eng_tcount)
                        for (int k = 0; k < 4; ++k)
                            if (k == 2 || t % 8 == 2) hipEventElapsedTime(&ms[k], ch->eng_tev[k][i][0], ch->eng_tev[k][i][1]);
                    fprintf(fp, "%d,%u,%u,%.4f,%.4f,%.4f,%.4f\n", t, h[2 * t], h[2 * t + 1], ms[0], ms[1], ms[2], ms[3]);
                }
                fclose(fp);
            }
        }
#ifdef OPTIK_NNLS_TRACE
        if (const char *tf = getenv("OPTIK_NNLS_TRACE_FILE")) {
            std::vector<unsigned long long> h(4 * 65536);
            HIP_TRY(hipMemcpy(h.data(), ch->nnls_trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (FILE *fp = fopen(tf, "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), fp); fclose(fp); }
        }
#endif
        {
            unsigned long long h[1 + ENG_EXEC_SHARDS];
            HIP_TRY(hipMemcpy(h, ch->eng_nn_total, sizeof h, hipMemcpyDeviceToHost));
            ch->eng_nn_problems = h[0];
            ch->eng_exec_evals = 0;
            for (int k = 0; k < ENG_EXEC_SHARDS; ++k) ch->eng_exec_evals += h[1 + k];
        }
        return 0;
#undef DISPATCH_N
    };
    rc = run();
    }  // job groups
    for (auto &j : ch->eng_jobs) {
        if (j.own_x) (void)hipFree(j.own_x);
        if (j.own_f) (void)hipFree(j.own_f);
        if (j.own_key) (void)hipFree(j.own_key);
        if (j.own_fs) (void)hipFree(j.own_fs);
    }
    ch->eng_jobs.clear();
    return rc;
}

/* Allocates the engine's slot pool and work buffers for up to `slots` slots (0 = the default
 * capacity) ahead of the first run: set-up, not part of a timed region. */
int optik_hip_engine_reserve(optik_hip_chain *ch, uint64_t slots, void *stream_v) {
    if (!ch) return fail(OPTIK_HIP_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lock(ch->mu);
    BIND_DEVICE(ch);
    size_t cap = 393216;
    if (opt().engine_slots >= 256) cap = (size_t)opt().engine_slots;
    const size_t AC = slots ? (size_t)((slots + 255) / 256 * 256) : cap;
    int nd = 0, ni = 0, rec_len = 0;
    switch (ch->n) {
#define M_LAYOUT(NN) case NN: nd = EngLayout<NN>::ND; ni = EngLayout<NN>::NI; rec_len = rec_stride<NN>(); break
    M_LAYOUT(1); M_LAYOUT(2); M_LAYOUT(3); M_LAYOUT(4); M_LAYOUT(5); M_LAYOUT(6); M_LAYOUT(7);
#undef M_LAYOUT
    default: return fail(OPTIK_HIP_EUNSUPPORTED, OPTIK_N_RANGE_MSG);
    }
    hipStream_t stream = (hipStream_t)stream_v;
    std::lock_guard<std::mutex> pool_lock(engine_pool_of(ch).run_mu);
    const int rc = engine_reserve(ch, AC, nd, ni, rec_len, stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

int optik_hip_engine_last_trips(const optik_hip_chain *ch) { return ch ? ch->eng_trips : 0; }

int optik_hip_engine_last_tail(const optik_hip_chain *ch, int32_t *restarts) {
    if (restarts) *restarts = ch ? ch->eng_tail_restarts : 0;
    return ch ? ch->eng_tail_solver : 0;
}

int optik_hip_engine_last_pools(const optik_hip_chain *ch, int32_t *launches) {
    if (!ch) return 0;
    if (launches) *launches = ch->eng_launches;
    return ch->eng_pools;
}

int optik_hip_engine_stats(const optik_hip_chain *ch, double *kernel_ms4, int32_t *sampled_trips,
                           uint64_t *nnls_problems) {
    if (!ch) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (kernel_ms4) for (int k = 0; k < 4; ++k) kernel_ms4[k] = ch->eng_kernel_ms[k];
    if (sampled_trips) *sampled_trips = ch->eng_tcount;
    if (nnls_problems) *nnls_problems = ch->eng_nn_problems;
    return 0;
}

uint64_t optik_hip_engine_executed_evals(const optik_hip_chain *ch) { return ch ? ch->eng_exec_evals : 0; }
#ifdef OPTIK_PROFILE_NNLS
/* diagnostic builds only: reads and clears the per-step cycle counters of nnls_coop */
int optik_hip_nnls_step_profile(unsigned long long *out8) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(optik::g_nnls_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(optik::g_nnls_prof), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif

/* Tuning options (tests, tools): see `struct Options`.  Names: solve_kernel (0 auto, 1 quad, 2 lane64, 3 general),
 * engine_slots, engine_pools, engine_nnls_budget, engine_nnls_slack, engine_tail_max, engine_compact, wide_form
 * (0 lds, 1 hbm), range_rule (of chains created afterwards), stop_x_legacy.  Not synchronised with calls in flight. */
static long long *option_slot(const char *name, int **islot) {
    Options &o = opt();
    *islot = nullptr;
    if (!name) return nullptr;
    if (!std::strcmp(name, "solve_kernel")) { *islot = &o.solve_kernel; return nullptr; }
    if (!std::strcmp(name, "engine_slots")) return &o.engine_slots;
    if (!std::strcmp(name, "engine_pools")) { *islot = &o.engine_pools; return nullptr; }
    if (!std::strcmp(name, "engine_nnls_budget")) { *islot = &o.engine_nnls_budget; return nullptr; }
    if (!std::strcmp(name, "engine_nnls_slack")) { *islot = &o.engine_nnls_slack; return nullptr; }
    if (!std::strcmp(name, "engine_tail_max")) return &o.engine_tail_max;
    if (!std::strcmp(name, "engine_compact")) { *islot = &o.engine_compact; return nullptr; }
    if (!std::strcmp(name, "wide_form")) { *islot = &o.wide_form; return nullptr; }
    if (!std::strcmp(name, "range_rule")) { *islot = &o.range_rule; return nullptr; }
    if (!std::strcmp(name, "stop_x_legacy")) { *islot = &o.stop_x_legacy; return nullptr; }
    return nullptr;
}
int optik_hip_set_option(const char *name, long long value) {
    int *is = nullptr;
    long long *ls = option_slot(name, &is);
    if (ls) { *ls = value; return 0; }
    if (is) { *is = (int)value; return 0; }
    return fail(OPTIK_HIP_EINVAL, "unknown option");
}
long long optik_hip_get_option(const char *name) {
    int *is = nullptr;
    long long *ls = option_slot(name, &is);
    return ls ? *ls : (is ? (long long)*is : -1);
}

int optik_hip_ik_host(optik_hip_chain *ch, const optik_solver_config *cfg, const double *targets,
                      const double *x0, int32_t T, const double *ee_offset7, uint64_t restart_begin,
                      uint64_t restart_end, uint32_t flags, double deadline_s, double *win_x, double *win_f,
                      uint64_t *win_idx, double *win_key) {
    if (!ch || !targets || !x0 || T < 1) return fail(OPTIK_HIP_EINVAL, "bad argument");
    // the launch workspace of the chain is in use until the copies below are done
    std::lock_guard<std::mutex> host_lock(ch->host_mu);
    BIND_DEVICE(ch);
    const int n = ch->n;
    // one device block and one pinned staging block, kept with the chain (a call used to pay
    // six hipMalloc / hipFree pairs and six copies): in = targets [T][7], x0 [T][n];
    // out = win_x [T][n], win_f [T], win_key [T], win_idx [T]
    const size_t n_in = (size_t)(7 + n) * (size_t)T, n_out = (size_t)(n + 3) * (size_t)T;
    // (`mu` from here to the launch: the claim state and the staging blocks belong to the launch workspace)
    std::unique_lock<std::mutex> launch_lock(ch->mu);
    if (n_in + n_out > ch->hw_cap) {
        if (ch->claim_pending) { (void)hipStreamSynchronize(nullptr); ch->claim_pending = false; }  // (its launch reads the block about to go)
        if (ch->hw_dev) (void)hipFree(ch->hw_dev);
        if (ch->hw_pin) (void)hipHostFree(ch->hw_pin);
        ch->hw_dev = nullptr; ch->hw_pin = nullptr; ch->hw_cap = 0;
        HIP_TRY(hipMalloc(&ch->hw_dev, sizeof(double) * (n_in + n_out)));
        HIP_TRY(hipHostMalloc(&ch->hw_pin, sizeof(double) * 2 * (n_in + n_out)));  // (two blocks, see below)
        ch->hw_cap = n_in + n_out;
    }
    // A few targets (Robot::ik: one): the kernels read the inputs from and write the winners to the
    // pinned block directly -- no copy commands around the launch.  (Two such blocks, used in turn: a call
    // that returned on the first success -- below -- leaves a launch behind whose last restarts still read theirs.)
    const bool zero_copy = T <= 16;
    double *pin = ch->hw_pin;
    if (zero_copy) {
        pin += (ch->hw_flip & 1u) * ch->hw_cap;
        ch->hw_flip ^= 1u;
    } else if (ch->claim_pending) {
        // The staged path always uses block 0.  A launch that a first-success call left running may have been given
        // that block: its selection kernel still writes its winner there -- inside the region the targets are about
        // to be staged in -- so it has to end first (the two-block flip only protects the zero-copy calls).
        HIP_TRY(hipStreamSynchronize(nullptr));
        ch->claim_pending = false;
    }
    double *io = zero_copy ? pin : ch->hw_dev;
    double *d_t = io, *d_x0 = d_t + (size_t)7 * T;
    double *d_wx = io + n_in, *d_wf = d_wx + (size_t)n * T, *d_wk = d_wf + T;
    uint64_t *d_wi = reinterpret_cast<uint64_t *>(d_wk + T);
    std::memcpy(pin, targets, sizeof(double) * 7 * (size_t)T);
    std::memcpy(pin + (size_t)7 * T, x0, sizeof(double) * (size_t)n * (size_t)T);
    if (!zero_copy)
        HIP_TRY(hipMemcpyAsync(ch->hw_dev, pin, sizeof(double) * n_in, hipMemcpyHostToDevice, nullptr));
    optik_hip_ik_outputs o;
    std::memset(&o, 0, sizeof o);
    o.d_win_x = d_wx; o.d_win_f = d_wf; o.d_win_idx = d_wi; o.d_win_key = d_wk;
    // One target under the first-success rule (lib.rs:409-412, the reference's default): the first restart to
    // succeed writes its answer to a host-coherent block and the call returns as soon as it is there; the launch's
    // other restarts notice the flag at their next evaluation and the launch ends behind the caller's back (the
    // next launch of the chain queues behind it).  Without a success the call ends with the launch, as before.
    const bool claim = T == 1 && (flags & OPTIK_HIP_IK_FIND_ANY) && (flags & OPTIK_HIP_IK_EARLY_EXIT)
                       && !(flags & OPTIK_HIP_IK_ENGINE) && cfg->solution_mode == 2;
    if (claim && !ch->hw_claim) {
        HIP_TRY(hipHostMalloc(&ch->hw_claim, sizeof(unsigned long long) * (3 + MAX_DOF), hipHostMallocCoherent));
        std::memset(ch->hw_claim, 0, sizeof(unsigned long long) * (3 + MAX_DOF));
        HIP_TRY(hipEventCreateWithFlags(&ch->claim_done, hipEventDisableTiming));
    }
    bool armed = false;
    unsigned long long seq = 0;
    int rc;
    if (flags & OPTIK_HIP_IK_ENGINE) {
        launch_lock.unlock();  // (an engine run takes the lock itself)
        rc = optik_hip_engine_solve(ch, cfg, d_t, d_x0, T, ee_offset7, restart_begin, restart_end,
                                    flags & ~OPTIK_HIP_IK_ENGINE, deadline_s, &o, nullptr);
    } else {
        rc = ik_batch_locked(ch, cfg, d_t, d_x0, T, ee_offset7, restart_begin, restart_end, flags, deadline_s, &o, nullptr,
                             claim, &armed);
        seq = ch->claim_seq;
        // (the end of THIS launch, not of the null stream: other chains' calls may keep that one busy)
        if (!rc && armed && hipEventRecord(ch->claim_done, nullptr) != hipSuccess) rc = fail(OPTIK_HIP_ENODEVICE, "hipEventRecord failed");
        launch_lock.unlock();
    }
    if (rc) return rc;
    double *h_out = pin + n_in;
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(h_out, d_wx, sizeof(double) * n_out, hipMemcpyDeviceToHost, nullptr));
    if (armed) {
        volatile unsigned long long *cw = ch->hw_claim;
        for (unsigned spin = 1;; ++spin) {
            if (__atomic_load_n(ch->hw_claim, __ATOMIC_ACQUIRE) == seq) {
                if (win_x) std::memcpy(win_x, (const void *)(cw + 3), sizeof(double) * (size_t)n);
                if (win_f) std::memcpy(win_f, (const void *)(cw + 2), sizeof(double));
                if (win_idx) *win_idx = cw[1];
                if (win_key) *win_key = (double)cw[1];
                return 0;  // (claim_pending stays set: the launch ends behind the caller's back)
            }
            if ((spin & 63u) == 0) {
                const hipError_t q = hipEventQuery(ch->claim_done);
                if (q == hipSuccess) break;  // the launch is over and nobody succeeded (or the word is about to land)
                if (q != hipErrorNotReady) HIP_TRY(q);
            }
        }
        if (__atomic_load_n(ch->hw_claim, __ATOMIC_ACQUIRE) == seq) {
            if (win_x) std::memcpy(win_x, (const void *)(cw + 3), sizeof(double) * (size_t)n);
            if (win_f) std::memcpy(win_f, (const void *)(cw + 2), sizeof(double));
            if (win_idx) *win_idx = cw[1];
            if (win_key) *win_key = (double)cw[1];
            return 0;
        }
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    if (armed) {
        // (nothing of this call is left on the null stream -- unless a later launch of the chain armed a claim meanwhile)
        std::lock_guard<std::mutex> relock(ch->mu);
        if (ch->claim_seq == seq) ch->claim_pending = false;
    }
    if (win_x) std::memcpy(win_x, h_out, sizeof(double) * (size_t)n * (size_t)T);
    if (win_f) std::memcpy(win_f, h_out + (size_t)n * T, sizeof(double) * (size_t)T);
    if (win_key) std::memcpy(win_key, h_out + (size_t)(n + 1) * T, sizeof(double) * (size_t)T);
    if (win_idx) std::memcpy(win_idx, h_out + (size_t)(n + 2) * T, sizeof(uint64_t) * (size_t)T);
    return 0;
}

int optik_hip_engine_solve(optik_hip_chain *ch, const optik_solver_config *cfg, const double *d_targets,
                           const double *d_x0, int32_t T, const double *ee_offset7, uint64_t restart_begin,
                           uint64_t restart_end, uint32_t flags, double deadline_s,
                           const optik_hip_ik_outputs *out, void *stream) {
    if (!ch) return fail(OPTIK_HIP_EINVAL, "bad argument");
    std::lock_guard<std::mutex> session(ch->eng_session_mu);
    int rc = optik_hip_engine_submit(ch, cfg, d_targets, d_x0, T, ee_offset7, restart_begin, restart_end, flags, out);
    if (!rc) rc = optik_hip_engine_run_ex(ch, stream, deadline_s);
    return rc;
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

int optik_hip_probe_math(int32_t op, const double *poses7, int64_t count, double *out) {
    if (!poses7 || !out || count < 0 || op < 0 || op > 3) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (int rc = ensure_device()) return rc;
    if (count == 0) return 0;
    const int stride = op == 0 ? 3 : (op == 1 ? 9 : (op == 2 ? 6 : 36));
    // (both buffers are released on every path out)
    struct DevBuf {
        double *p = nullptr;
        ~DevBuf() { if (p) (void)hipFree(p); }
    } d_p, d_o;
    HIP_TRY(hipMalloc(&d_p.p, sizeof(double) * 7 * (size_t)count));
    HIP_TRY(hipMalloc(&d_o.p, sizeof(double) * (size_t)stride * (size_t)count));
    HIP_TRY(hipMemcpy(d_p.p, poses7, sizeof(double) * 7 * (size_t)count, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_math_kernel, dim3(64), dim3(64), 0, nullptr, op, d_p.p, (long long)count, d_o.p, stride);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_o.p, sizeof(double) * (size_t)stride * (size_t)count, hipMemcpyDeviceToHost));
    return 0;
}

void optik_hip_set_timing(optik_hip_chain *ch, int32_t enabled) {
    if (!ch) return;
    std::lock_guard<std::mutex> lock(ch->mu);
    ch->timing = enabled;
    ch->ev_count = 0;
    if (enabled) {
        // the engine's per-trip events are created here, not lazily inside a timed run
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < optik_hip_chain::ENG_EV; ++i)
                for (int w = 0; w < 2; ++w)
                    if (!ch->eng_tev[k][i][w] && (k == 2 || (i + 1) % 8 == 2))
                        (void)hipEventCreate(&ch->eng_tev[k][i][w]);
    }
}

/* OPTIK_PROFILE builds: phase cycle totals of the last solve launch (8 words:
 * refill, eval, update, publish, bfgs, lsq, nnls, trips); zeros otherwise. */
int optik_hip_phase_profile(optik_hip_chain *ch, unsigned long long *out8) {
    if (!ch || !out8) return fail(OPTIK_HIP_EINVAL, "bad argument");
    std::memset(out8, 0, 8 * sizeof(unsigned long long));
    if (!ch->prof) return 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out8, ch->prof, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
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
