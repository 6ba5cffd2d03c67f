// ik_quad_kernel.hip -- the single-launch restart solver with one restart per quad of lanes and the
// restart's state spread over the quad (ik_quad.hpp): the latency path of optik_hip_ik_batch and,
// with 16 restarts per wave, a throughput path whose per-restart state never touches HBM.
//
// Its own translation unit: the kernel is tuned against the register allocator (waves per SIMD), and
// rebuilding it must not wait for the other solvers' template instantiations.
#include <hip/hip_runtime.h>

#include "ik_launch.hpp"
#include "ik_quad.hpp"

namespace optik {

#ifndef OPTIK_QUAD_WAVES
#define OPTIK_QUAD_WAVES 2  // waves per SIMD of the throughput build (LDS allows two)
#endif
// This file is compiled TWICE (optik_amd/build.py): OPTIK_QUAD_PART = 2 holds the throughput form
// (n <= 7, W = OPTIK_QUAD_WAVES) and is built with `-mllvm -disable-machine-licm -mllvm
// -amdgpu-use-amdgpu-trackers=1` -- the machine-LICM pass hoists the ~60 double constants of sin / cos /
// atan2 and LDS address variants out of the solver loop, and at 256 registers the allocator then spills
// those loop invariants to scratch and reloads them every trip (412 -> 280 B of scratch, +4-5 %
// restarts/s); OPTIK_QUAD_PART = 1 holds the latency forms (W = 1: no scratch either way; built without the pass
// too since round 4 -- 2 % off a single call's latency) and the launch function.
#ifndef OPTIK_QUAD_PART
#define OPTIK_QUAD_PART 0  // 0: everything in one object (tools/, experiments)
#endif

// Two builds of the same body: W = waves per SIMD the register allocator leaves room for.  W = 2
// (256 registers, some cold spills; 8 waves per CU) is the throughput form; W = 1 (all 512
// registers, no scratch) the latency form for launches that cannot fill even one wave per SIMD
// with restarts -- a single ik() call's first rounds.
template <int N, bool TIP, int W>
__global__ __launch_bounds__(64, W) void ik_quad_kernel(const SolveLaunch a) {
    __shared__ ChainDev sch;
    __shared__ __attribute__((aligned(16))) double nnls_lds[quad_wave_lds<N>()];
    __shared__ double lane_lds[quad_lane_lds()];
    // The launch parameters (~540 bytes: weights, ChaCha key, scales, queue and output pointers) are
    // copied from the kernel-argument segment to LDS once and read from there: as kernel arguments
    // they would sit in ~125 SGPRs for the whole kernel, next to the ~60 double constants of
    // sin / cos / atan2 the compiler hoists out of the solver loop -- more than the 102 a wave has;
    // the overflow spills into VGPR lanes, and those VGPRs into scratch.
    __shared__ __attribute__((aligned(8))) uint32_t launch_lds[(sizeof(SolveLaunch) + 3) / 4];
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&a);
        for (unsigned i = threadIdx.x; i < sizeof(SolveLaunch) / 4; i += 64) launch_lds[i] = src[i];
    }
    if (threadIdx.x < 16) nnls_lds[quad_wave_lds<N>() - 16 + threadIdx.x] = 0.0;  // the column of zeros
    stage_chain(sch, a.chain);
    SolveLaunch &L = *reinterpret_cast<SolveLaunch *>(launch_lds);
    if (threadIdx.x == 0) L.wq.deadline = L.deadline_ticks ? wall_clock64() + L.deadline_ticks : 0ull;
    __syncthreads();
    quad_wave<N, TIP>(sch, L.ep, L.sp, L.key, L.scale, L.wq, nnls_lds, lane_lds);
}

// (n = 8: nine-row columns make the blocks 22 KB per wave: six waves per CU of the throughput form --
// 236 B of scratch, 12.1 against 9.3 M restarts/s with four waves of the latency form)
#if OPTIK_QUAD_PART != 1
// the throughput form, reached through one entry point per (n, tip) so that it can live in its own object
hipError_t quad_solve_launch_w2(int n, bool tip, int grid, hipStream_t stream, const SolveLaunch &a) {
#define CALL_QUAD2(NN)                                                                                  \
    case NN:                                                                                            \
        if (tip) hipLaunchKernelGGL((ik_quad_kernel<NN, true, OPTIK_QUAD_WAVES>), dim3(grid), dim3(64), 0, stream, a);  \
        else hipLaunchKernelGGL((ik_quad_kernel<NN, false, OPTIK_QUAD_WAVES>), dim3(grid), dim3(64), 0, stream, a);     \
        break;
    switch (n) {
#ifdef OPTIK_QUAD_ONLY_N
        CALL_QUAD2(OPTIK_QUAD_ONLY_N)
#else
        CALL_QUAD2(1) CALL_QUAD2(2) CALL_QUAD2(3) CALL_QUAD2(4) CALL_QUAD2(5) CALL_QUAD2(6) CALL_QUAD2(7)
        CALL_QUAD2(8)  // (nine-row columns: 26 KB of LDS per wave, six waves per CU)
#endif
    default: return hipErrorInvalidValue;
    }
#undef CALL_QUAD2
    return hipGetLastError();
}
#else
hipError_t quad_solve_launch_w2(int n, bool tip, int grid, hipStream_t stream, const SolveLaunch &a);
#endif

#if OPTIK_QUAD_PART != 2
int quad_solve_waves_per_cu(int n) { return n <= 7 ? 4 * OPTIK_QUAD_WAVES : (OPTIK_QUAD_WAVES > 1 ? 6 : 4); }

hipError_t quad_solve_launch(int n, bool tip, int grid, hipStream_t stream, const SolveLaunch &a, int *lds_bytes,
                             bool latency_form) {
    if (lds_bytes)
        *lds_bytes = (int)(sizeof(ChainDev) + sizeof(SolveLaunch)
                           + sizeof(double) * ((n <= 7 ? quad_wave_lds<7>() : quad_wave_lds<8>()) + quad_lane_lds()));
#define CALL_QUAD(NN)                                                                                   \
    case NN:                                                                                            \
        if (latency_form) {                                                                             \
            if (tip) hipLaunchKernelGGL((ik_quad_kernel<NN, true, 1>), dim3(grid), dim3(64), 0, stream, a);  \
            else hipLaunchKernelGGL((ik_quad_kernel<NN, false, 1>), dim3(grid), dim3(64), 0, stream, a);     \
        } else {                                                                                        \
            return quad_solve_launch_w2(n, tip, grid, stream, a);                                       \
        }                                                                                               \
        break;
    switch (n) {
#ifdef OPTIK_QUAD_ONLY_N
        CALL_QUAD(OPTIK_QUAD_ONLY_N)
#else
        CALL_QUAD(1) CALL_QUAD(2) CALL_QUAD(3) CALL_QUAD(4) CALL_QUAD(5) CALL_QUAD(6) CALL_QUAD(7)
        CALL_QUAD(8)
#endif
    default: return hipErrorInvalidValue;
    }
#undef CALL_QUAD
    return hipGetLastError();
}

#endif  // OPTIK_QUAD_PART != 2

}  // namespace optik

#if defined(OPTIK_PROFILE) && OPTIK_QUAD_PART != 1  // (next to the throughput form: its copy of the counters)
// diagnostic builds: cycles per part of the quad NNLS since the last call (ik_nnls_quad.hpp), then reset
extern "C" int optik_hip_quad_nnls_profile(unsigned long long *out8) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(optik::g_quad_nnls_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(optik::g_quad_nnls_prof), z, sizeof z) == hipSuccess ? 0 : -1;
}
// ... and the histograms of the same calls (ik_nnls_quad.hpp:g_quad_nnls_hist), then reset
extern "C" int optik_hip_quad_nnls_hist(unsigned long long *out66) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out66, HIP_SYMBOL(optik::g_quad_nnls_hist), 66 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[66] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(optik::g_quad_nnls_hist), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
