// ik_lane_kernel.hip -- the throughput form of the single-launch restart solver for chains of at most seven
// joints: one restart per lane, the wave's bounded sub-problems solved sixteen at a time in class order by
// quads of lanes (ik_lane64.hpp).  One wave per SIMD (512 registers, 38 KB of LDS per wave).
//
// Its own translation unit: tuned against the register allocator on its own.
#include <hip/hip_runtime.h>

#include "ik_launch.hpp"
#include "ik_lane64.hpp"

namespace optik {

template <int N, bool TIP>
__global__ __launch_bounds__(64, 1) void ik_lane_kernel(const SolveLaunch a) {
    __shared__ ChainDev sch;
    __shared__ __attribute__((aligned(16))) double nnls_lds[lane64_block_lds<N>()];
    __shared__ __attribute__((aligned(16))) double rec_lds[lane64_rec_lds<N>()];
    __shared__ int lor_lds[64], where_lds[64];
    // (the launch parameters in LDS, as in ik_quad_kernel: ~125 SGPRs otherwise)
    __shared__ __attribute__((aligned(8))) uint32_t launch_lds[(sizeof(SolveLaunch) + 3) / 4];
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&a);
        for (unsigned i = threadIdx.x; i < sizeof(SolveLaunch) / 4; i += 64) launch_lds[i] = src[i];
    }
    if (threadIdx.x < 16) nnls_lds[lane64_block_lds<N>() - 16 + threadIdx.x] = 0.0;  // the column of zeros
    lor_lds[threadIdx.x & 63u] = 0;
    where_lds[threadIdx.x & 63u] = 0;
    stage_chain(sch, a.chain);
    SolveLaunch &L = *reinterpret_cast<SolveLaunch *>(launch_lds);
    if (threadIdx.x == 0) L.wq.deadline = L.deadline_ticks ? wall_clock64() + L.deadline_ticks : 0ull;
    __syncthreads();
    lane64_wave<N, TIP>(sch, L.ep, L.sp, L.key, L.scale, L.wq, nnls_lds, rec_lds, lor_lds, where_lds);
}

int lane_solve_waves_per_cu() { return 4; }

hipError_t lane_solve_launch(int n, bool tip, int grid, hipStream_t stream, const SolveLaunch &a, int *lds_bytes) {
    if (lds_bytes)
        *lds_bytes = (int)(sizeof(ChainDev) + sizeof(SolveLaunch) + 128 * sizeof(int)
                           + sizeof(double) * (lane64_block_lds<7>() + lane64_rec_lds<7>()));
#define CALL_LANE(NN)                                                                                  \
    case NN:                                                                                           \
        if (tip) hipLaunchKernelGGL((ik_lane_kernel<NN, true>), dim3(grid), dim3(64), 0, stream, a);   \
        else hipLaunchKernelGGL((ik_lane_kernel<NN, false>), dim3(grid), dim3(64), 0, stream, a);      \
        break;
    switch (n) {
#ifdef OPTIK_LANE_ONLY_N
        CALL_LANE(OPTIK_LANE_ONLY_N)
#else
        CALL_LANE(1) CALL_LANE(2) CALL_LANE(3) CALL_LANE(4) CALL_LANE(5) CALL_LANE(6) CALL_LANE(7)
#endif
    default: return hipErrorInvalidValue;
    }
#undef CALL_LANE
    return hipGetLastError();
}

}  // namespace optik

#ifdef OPTIK_PROFILE
// diagnostic builds: the NNLS histograms of THIS object's calls (ik_nnls_quad.hpp: g_quad_nnls_hist -- every translation
// unit has its own copy of the counters), then reset
extern "C" int optik_hip_lane_nnls_hist(unsigned long long *out66) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out66, HIP_SYMBOL(optik::g_quad_nnls_hist), 66 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[66] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(optik::g_quad_nnls_hist), z, sizeof z) == hipSuccess ? 0 : -1;
}
// ... and the cycles of those loop trips by quads solving (18 words: ik_nnls_quad.hpp g_quad_nnls_tripcyc) + the 8 part counters
extern "C" int optik_hip_lane_nnls_cycles(unsigned long long *out26) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out26, HIP_SYMBOL(optik::g_quad_nnls_tripcyc), 18 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out26 + 18, HIP_SYMBOL(optik::g_quad_nnls_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[18] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(optik::g_quad_nnls_prof), z, 8 * sizeof(unsigned long long));
    return hipMemcpyToSymbol(HIP_SYMBOL(optik::g_quad_nnls_tripcyc), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
