// ik_launch.hpp -- what a launch of a restart-solving kernel receives (kernel arguments -> SGPRs),
// shared by the translation units that hold those kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "ik_solve.hpp"

namespace optik {

// Where the lane-per-restart kernel leaves the restarts still running when its queue is dry, for the quad solver to
// finish (ik_spill.hpp).  Plain data: planes of doubles / ints over C slots (slot = a lane's global number), the
// list of slots that hold a restart and its length.  d == nullptr: the launch does not spill.
struct SpillPool {
    double *d;                   // [SpillLayout::ND][C]
    int32_t *i32;                // [SpillLayout::NI][C]
    unsigned long long *item;    // [C] output column of the slot's restart (target * n_restarts + restart)
    unsigned long long C;
    unsigned int *list;          // [C] slots holding a restart ...
    unsigned int *count;         // ... how many (appended to by the spilling waves; zero before the launch)
    unsigned long long *cursor;  // the tail kernel's hand-out counter (zero before the launch)
    unsigned long long *deadline;  // the lane kernel's absolute deadline (wall_clock64 ticks, 0 = none), for the tail
    int spill_at;                // a wave spills when at most this many of its lanes still hold a restart
    int pad;
};

struct SolveLaunch {
    const ChainDev *chain;
    EvalParams ep;
    SolveParams sp;
    uint32_t key[8];        // ChaCha key = seed_from_u64(42)
    double scale[MAX_DOF];  // rand UniformFloat scale per joint
    WorkQueue wq;
    unsigned long long deadline_ticks;  // relative to kernel start, 0 = none
    SpillPool spill;        // (lane-per-restart kernel and its tail)
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
// the spilled restarts of a lane-per-restart launch on the quad solver (n <= 7, the two-waves-per-SIMD build): queued
// behind that launch on the same stream; `a` is the launch's own record (a.wq.lanes is set to 16 quads per wave here)
hipError_t quad_tail_launch(int n, bool tip, int grid, hipStream_t stream, const SolveLaunch &a);

}  // namespace optik
