// ik_select.hip -- the selection of /root/reference/crates/optik/src/lib.rs:397-413 over the per-restart keys a solver
// launch leaves behind: Speed keeps the lowest successful restart index, Quality the solution closest to the seed
// (ties to the lower index).
//
//   ik_tile_argmin_kernel    per 4096-restart tile: wavefront-shuffle argmin of (key, index), one 16-byte record
//   ik_select_kernel         per target: the minimum of its tile records, the winner's x / f gathered
//   ik_select_small_kernel   both in one kernel when a target has a single tile (a single ik() call's first
//                            launches, a Speed batch's rounds)
// The last kernel of a launch also puts the launch's work-item counter and first-success words back to their
// initial values, so that the next launch needs no fill commands in front of it.
#include "ik_host.hpp"

namespace optik {
namespace host {
namespace {

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

}  // namespace

hipError_t select_launch(const SelectLaunch &s, int T, hipStream_t stream) {
    if (s.tiles_per_target == 1) {
        hipLaunchKernelGGL(ik_select_small_kernel, dim3(T), dim3(256), 0, stream, s);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(ik_tile_argmin_kernel, dim3((unsigned)(s.tiles_per_target * (long long)T)), dim3(256), 0, stream, s);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    hipLaunchKernelGGL(ik_select_kernel, dim3(T), dim3(WAVE), 0, stream, s);
    return hipGetLastError();
}

}  // namespace host
}  // namespace optik
