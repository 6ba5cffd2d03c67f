// ik_quad_kernel.hip -- the single-launch restart solver with one restart per quad of lanes and the
// restart's state spread over the quad (ik_quad.hpp): the latency path of optik_hip_ik_batch and,
// with 16 restarts per wave, a throughput path whose per-restart state never touches HBM.
//
// Its own translation unit: the kernel is tuned against the register allocator (waves per SIMD), and
// rebuilding it must not wait for the streaming engine's two minutes of template instantiations.
#include <hip/hip_runtime.h>

#include "ik_launch.hpp"
#include "ik_quad.hpp"

namespace optik {

#ifndef OPTIK_QUAD_WAVES
#define OPTIK_QUAD_WAVES 2  // waves per SIMD the register allocator must leave room for
#endif

template <int N, bool TIP>
__global__ __launch_bounds__(64, OPTIK_QUAD_WAVES) void ik_quad_kernel(const SolveLaunch a) {
    __shared__ ChainDev sch;
    __shared__ __attribute__((aligned(16))) double nnls_lds[quad_wave_lds()];
    __shared__ double lane_lds[quad_lane_lds()];
    if (threadIdx.x < 8) nnls_lds[quad_wave_lds() - 8 + threadIdx.x] = 0.0;  // the column of zeros
    stage_chain(sch, a.chain);
    WorkQueue wq = a.wq;
    wq.deadline = a.deadline_ticks ? wall_clock64() + a.deadline_ticks : 0ull;
    quad_wave<N, TIP>(sch, a.ep, a.sp, a.key, a.scale, wq, nnls_lds, lane_lds);
}

int quad_solve_waves_per_cu() { return 4 * OPTIK_QUAD_WAVES; }

hipError_t quad_solve_launch(int n, bool tip, int grid, hipStream_t stream, const SolveLaunch &a, int *lds_bytes) {
    if (lds_bytes) *lds_bytes = (int)(sizeof(ChainDev) + sizeof(double) * (quad_wave_lds() + quad_lane_lds()));
#define CALL_QUAD(NN)                                                                                   \
    case NN:                                                                                            \
        if (tip) hipLaunchKernelGGL((ik_quad_kernel<NN, true>), dim3(grid), dim3(64), 0, stream, a);    \
        else hipLaunchKernelGGL((ik_quad_kernel<NN, false>), dim3(grid), dim3(64), 0, stream, a);       \
        break;
    switch (n) {
#ifdef OPTIK_QUAD_ONLY_N
        CALL_QUAD(OPTIK_QUAD_ONLY_N)
#else
        CALL_QUAD(1) CALL_QUAD(2) CALL_QUAD(3) CALL_QUAD(4) CALL_QUAD(5) CALL_QUAD(6) CALL_QUAD(7)
#endif
    default: return hipErrorInvalidValue;
    }
#undef CALL_QUAD
    return hipGetLastError();
}

}  // namespace optik
