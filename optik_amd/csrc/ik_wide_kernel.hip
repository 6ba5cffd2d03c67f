// ik_wide_kernel.hip -- kernels for chains with 9 .. 16 joint positions (ik_wide.hpp): the restart
// solver, and the batched objective / forward kinematics / seed kernels of the same chains.  Its own
// translation unit: one run-time-n body each instead of a template instantiation per joint count.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ik_wide.hpp"

namespace optik {

namespace {

__device__ __forceinline__ void stage_wide_chain(WideChainDev &dst, const WideChainDev *src) {
    constexpr int ND = (int)(sizeof(WideChainDev) / sizeof(double));
    static_assert(sizeof(WideChainDev) % sizeof(double) == 0, "WideChainDev is a whole number of doubles");
    const double *s = reinterpret_cast<const double *>(src);
    double *d = reinterpret_cast<double *>(&dst);
    for (int i = threadIdx.x; i < ND; i += blockDim.x) d[i] = s[i];
    __syncthreads();
}

__global__ __launch_bounds__(64, 2) void wide_solve_kernel(const WideSolveLaunch a) {
    __shared__ WideChainDev sch;
    stage_wide_chain(sch, a.chain);
    WorkQueue wq = a.wq;
    wq.deadline = a.deadline_ticks ? (unsigned long long)wall_clock64() + a.deadline_ticks : 0ull;
    wide_solve_wave(sch, a.ep, a.sp, a.key, wq,
                    WPG{a.ws + (size_t)blockIdx.x * (size_t)wide_ws::SLOTS * 64 + (threadIdx.x & 63u)});
}

// One restart per wave (wq.lanes == 1: a single ik() call's rounds): the restart's arrays in the wave's LDS.
__global__ __launch_bounds__(64) void wide_solve_lds_kernel(const WideSolveLaunch a) {
    __shared__ WideChainDev sch;
    __shared__ double ws_lds[wide_ws::SLOTS];
    stage_wide_chain(sch, a.chain);
    WorkQueue wq = a.wq;
    wq.deadline = a.deadline_ticks ? (unsigned long long)wall_clock64() + a.deadline_ticks : 0ull;
    wq.lanes = 1;
    wide_solve_wave(sch, a.ep, a.sp, a.key, wq, WPL{(lds_double *)ws_lds});
}

// The same with the wave's 64 lanes working on the restart together (WPC): the default latency form.
__global__ __launch_bounds__(64, 2) void wide_solve_coop_kernel(const WideSolveLaunch a) {
    __shared__ WideChainDev sch;
    __shared__ double ws_lds[wide_ws::SLOTS];
    stage_wide_chain(sch, a.chain);
    WorkQueue wq = a.wq;
    wq.deadline = a.deadline_ticks ? (unsigned long long)wall_clock64() + a.deadline_ticks : 0ull;
    wq.lanes = 1;
    wide_solve_wave(sch, a.ep, a.sp, a.key, wq, WPC{(lds_double *)ws_lds});
}

__global__ __launch_bounds__(256) void wide_eval_batch_kernel(const WideBatchLaunch a) {
    __shared__ WideChainDev sch;
    stage_wide_chain(sch, a.chain);
    const int n = sch.n_pos;
    const Pose target = load_pose(a.target);
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.B;
         b += (long long)gridDim.x * blockDim.x) {
        double q[WIDE_MAX_DOF], g[WIDE_MAX_DOF], tf[7 * WIDE_MAX_DOF];
        for (int i = 0; i < n; ++i) q[i] = a.q[(size_t)i * a.B + b];
        a.f[b] = wide_eval_fg(sch, a.ep, target, n, q, tf, g);
        if (a.g)
            for (int i = 0; i < n; ++i) a.g[(size_t)i * a.B + b] = g[i];
    }
}

__global__ __launch_bounds__(256) void wide_fk_batch_kernel(const WideBatchLaunch a) {
    __shared__ WideChainDev sch;
    stage_wide_chain(sch, a.chain);
    const int n = sch.n_pos;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.B;
         b += (long long)gridDim.x * blockDim.x) {
        double q[WIDE_MAX_DOF], tf[7 * WIDE_MAX_DOF];
        for (int i = 0; i < n; ++i) q[i] = a.q[(size_t)i * a.B + b];
        const Pose ee = wide_forward(sch, a.ep, n, q, tf);
        const double p[7] = {ee.t.x, ee.t.y, ee.t.z, ee.q.i, ee.q.j, ee.q.k, ee.q.w};
#pragma unroll
        for (int i = 0; i < 7; ++i) a.pose[(size_t)i * a.B + b] = p[i];
        if (a.jac) {
            // joint_jacobian, kinematics.rs:166-196
            const Q4 eeqc = qconj(ee.q);
            for (int k = 0; k < n; ++k) {
                const V3 tk{tf[7 * k + 0], tf[7 * k + 1], tf[7 * k + 2]};
                const Q4 tq{tf[7 * k + 3], tf[7 * k + 4], tf[7 * k + 5], tf[7 * k + 6]};
                const V3 ax{sch.axis[k][0], sch.axis[k][1], sch.axis[k][2]};
                const V3 angular = qrot(tq, ax);
                const V3 d{ee.t.x - tk.x, ee.t.y - tk.y, ee.t.z - tk.z};
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

__global__ __launch_bounds__(256) void wide_seed_batch_kernel(const WideBatchLaunch a) {
    __shared__ WideChainDev sch;
    stage_wide_chain(sch, a.chain);
    const int n = sch.n_pos;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.B;
         b += (long long)gridDim.x * blockDim.x) {
        double q[WIDE_MAX_DOF];
        wide_restart_seed(a.key, sch.lb, sch.scale, a.first + (unsigned long long)b, n, q);
        for (int i = 0; i < n; ++i) a.q_out[(size_t)i * a.B + b] = q[i];
    }
}

}  // namespace

size_t wide_ws_doubles_per_wave() { return (size_t)wide_ws::SLOTS * 64; }

hipError_t wide_solve_launch(int grid, hipStream_t stream, const WideSolveLaunch &a, bool lds_form, bool lds_coop) {
    // (lds_coop false: the one-lane LDS form instead of the cooperative one -- option wide_form = 2, comparisons)
    if (lds_form && lds_coop) hipLaunchKernelGGL(wide_solve_coop_kernel, dim3(grid), dim3(64), 0, stream, a);
    else if (lds_form) hipLaunchKernelGGL(wide_solve_lds_kernel, dim3(grid), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(wide_solve_kernel, dim3(grid), dim3(64), 0, stream, a);
    return hipGetLastError();
}
int wide_lds_bytes() { return (int)(sizeof(WideChainDev) + sizeof(double) * wide_ws::SLOTS); }

hipError_t wide_batch_launch(int op, int grid, hipStream_t stream, const WideBatchLaunch &a) {
    switch (op) {
    case 0: hipLaunchKernelGGL(wide_eval_batch_kernel, dim3(grid), dim3(256), 0, stream, a); break;
    case 1: hipLaunchKernelGGL(wide_fk_batch_kernel, dim3(grid), dim3(256), 0, stream, a); break;
    case 2: hipLaunchKernelGGL(wide_seed_batch_kernel, dim3(grid), dim3(256), 0, stream, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace optik
