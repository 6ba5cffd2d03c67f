// ik_launch.hpp -- what a launch of a restart-solving kernel receives (kernel arguments -> SGPRs),
// shared by the translation units that hold those kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "ik_solve.hpp"

namespace optik {

struct SolveLaunch {
    const ChainDev *chain;
    EvalParams ep;
    SolveParams sp;
    uint32_t key[8];        // ChaCha key = seed_from_u64(42)
    double scale[MAX_DOF];  // rand UniformFloat scale per joint
    WorkQueue wq;
    unsigned long long deadline_ticks;  // relative to kernel start, 0 = none
};

__device__ __forceinline__ void stage_chain(ChainDev &dst, const ChainDev *src) {
    constexpr int ND = (int)(sizeof(ChainDev) / sizeof(double));
    static_assert(sizeof(ChainDev) % sizeof(double) == 0, "ChainDev is a whole number of doubles");
    const double *s = reinterpret_cast<const double *>(src);
    double *d = reinterpret_cast<double *>(&dst);
    for (int i = threadIdx.x; i < ND; i += blockDim.x) d[i] = s[i];
    __syncthreads();
}

// ik_quad_kernel.hip: the quad-distributed solver (ik_quad.hpp), n <= 8.  Launches `grid` single-wave
// workgroups on `stream`; *lds_bytes = static LDS of the kernel.  latency_form: the one-wave-per-SIMD
// build without scratch (grids of at most 4 waves per CU).  Returns the hipGetLastError of the launch.
hipError_t quad_solve_launch(int n, bool tip, int grid, hipStream_t stream, const SolveLaunch &a, int *lds_bytes,
                             bool latency_form);
// resident single-wave workgroups per CU the kernel is built for (registers and LDS)
int quad_solve_waves_per_cu(int n);
// ik_lane_kernel.hip: one restart per lane, bounded sub-problems in class order (ik_lane64.hpp), n <= 7; one wave
// per SIMD
hipError_t lane_solve_launch(int n, bool tip, int grid, hipStream_t stream, const SolveLaunch &a, int *lds_bytes);
int lane_solve_waves_per_cu();

}  // namespace optik
