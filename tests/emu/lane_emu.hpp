// lane_emu.hpp -- TEST INFRASTRUCTURE: host emulation of a (partial) 64-lane wavefront.
//
// Lets the group-distributed solver (optik_amd/csrc/ik_quad.hpp: one restart per quad of four
// lanes, state spread over the lanes, cross-lane moves) run on the CPU so that its arithmetic can
// be compared bit for bit with the C oracle without a GPU.  One std::thread per lane; every
// cross-lane primitive (__shfl, __ballot, the quad moves, LDS hand-over points) is an exchange
// through a shared array between two barriers, so the threads advance in the lock step the
// hardware gives for free.  Only wave-uniform control flow around those primitives is supported
// -- which is also the rule the device code must obey.
//
// Never part of the product: liboptik_amd.so is built without OPTIK_LANE_EMU, and nothing under
// optik_amd/ or bench.py includes this file.  Compiled only by tests/emu/quad_emu.cpp (host
// clang, -ffp-contract=off like both the kernels and the oracle).
#pragma once

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <thread>

#define __device__
#define __host__
#define __forceinline__ inline

namespace optik_emu {

constexpr int MAX_LANES = 64;

struct Wave {
    int lanes = 4;  // emulated lanes (a multiple of 4, <= 64); the rest of the wave does not exist
    std::atomic<int> arrived{0};
    std::atomic<int> generation{0};
    unsigned long long xch[MAX_LANES];
};

struct Dim3 {
    unsigned x = 0, y = 0, z = 0;
};

inline thread_local Wave *t_wave = nullptr;
inline unsigned threadIdx_x();

inline void barrier() {
    Wave &w = *t_wave;
    const int gen = w.generation.load(std::memory_order_acquire);
    if (w.arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == w.lanes) {
        w.arrived.store(0, std::memory_order_relaxed);
        w.generation.store(gen + 1, std::memory_order_release);
    } else {
        // (a lane that never arrives = a cross-lane primitive under lane-divergent control flow: say so
        // instead of hanging the test run)
        int spins = 0;
        long yields = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (w.generation.load(std::memory_order_acquire) == gen) {
            if (++spins > 2000) {
                std::this_thread::yield();
                spins = 0;
                if ((++yields & 1023) == 0
                    && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
                    std::fprintf(stderr, "lane_emu: lane %u waited 120 s at a barrier: a collective was called under "
                                         "lane-divergent control flow\n", threadIdx_x());
                    std::abort();
                }
            }
        }
    }
}

template <class T>
inline unsigned long long to_bits(T v) {
    static_assert(sizeof(T) <= 8, "exchange slot is 8 bytes");
    unsigned long long b = 0;
    std::memcpy(&b, &v, sizeof(T));
    return b;
}
template <class T>
inline T from_bits(unsigned long long b) {
    T v;
    std::memcpy(&v, &b, sizeof(T));
    return v;
}

}  // namespace optik_emu

inline thread_local optik_emu::Dim3 threadIdx;
inline thread_local optik_emu::Dim3 blockIdx;
inline unsigned optik_emu::threadIdx_x() { return threadIdx.x; }

// ---- wave collectives (called by every emulated lane, in wave-uniform control flow) ----------

template <class T>
inline T __shfl(T v, int src, int /*width*/ = 64) {
    optik_emu::Wave &w = *optik_emu::t_wave;
    const int lane = (int)(threadIdx.x & 63u);
    w.xch[lane] = optik_emu::to_bits(v);
    optik_emu::barrier();
    // (a source lane outside the emulated part of the wave: the caller's own value -- as if that
    // lane held the same; only min / max style butterflies ever go there)
    const T r = (src >= 0 && src < w.lanes) ? optik_emu::from_bits<T>(w.xch[src]) : v;
    optik_emu::barrier();
    return r;
}

template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    return __shfl(v, (int)(threadIdx.x & 63u) ^ mask, width);
}

inline unsigned long long __ballot(bool p) {
    optik_emu::Wave &w = *optik_emu::t_wave;
    const int lane = (int)(threadIdx.x & 63u);
    w.xch[lane] = p ? 1ull : 0ull;
    optik_emu::barrier();
    unsigned long long m = 0;
    for (int i = 0; i < w.lanes; ++i) m |= (w.xch[i] & 1ull) << i;
    optik_emu::barrier();
    return m;
}

// ---- the DPP moves and lane counters of optik_amd/csrc/ik_lane.hpp, under their device names -------------------
// __builtin_amdgcn_update_dpp(old, v, dpp_ctrl, row_mask, bank_mask, bound_ctrl) with a quad_perm control (dpp_ctrl
// 0x00 .. 0xFF, the only kind the solvers use): lane i of every quad of four reads the lane (dpp_ctrl >> 2 (i & 3)) & 3
// of ITS quad -- the hardware's own decoding of the eight control bits, so that the control constants in ik_lane.hpp
// (0x00 / 0x55 / 0xAA / 0xFF broadcasts, 0xB1 / 0x4E butterflies, 0x39 / 0x4E / 0x93 rotations) are what the CPU suite
// checks, not a re-statement of what they are meant to do.  row_mask = bank_mask = 0xf and bound_ctrl = true are the only
// form used (every lane of a full quad has a valid source; `old` is never taken); anything else aborts.
namespace optik_emu {
template <class T>
inline T dpp_quad_perm(T v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    if (ctrl < 0 || ctrl > 0xff || row_mask != 0xf || bank_mask != 0xf || !bound_ctrl) {
        std::fprintf(stderr, "lane_emu: update_dpp form not emulated (ctrl 0x%x)\n", ctrl);
        std::abort();
    }
    const int lane = (int)(threadIdx_x() & 63u);
    return __shfl(v, (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3));
}
// v_mbcnt_lo / v_mbcnt_hi: add + the number of set mask bits below the lane in the low / high half of the wave
inline unsigned mbcnt_lo(unsigned mask, unsigned add) {
    const unsigned lane = threadIdx_x() & 63u;
    const unsigned below = lane >= 32u ? 0xffffffffu : ((1u << lane) - 1u);
    return add + (unsigned)__builtin_popcount(mask & below);
}
inline unsigned mbcnt_hi(unsigned mask, unsigned add) {
    const unsigned lane = threadIdx_x() & 63u;
    const unsigned below = lane <= 32u ? 0u : ((1u << (lane - 32u)) - 1u);
    return add + (unsigned)__builtin_popcount(mask & below);
}
}  // namespace optik_emu
#define __builtin_amdgcn_update_dpp(old, v, ctrl, row_mask, bank_mask, bound_ctrl) \
    optik_emu::dpp_quad_perm((v), (ctrl), (row_mask), (bank_mask), (bound_ctrl))
#define __builtin_amdgcn_mbcnt_lo(mask, add) optik_emu::mbcnt_lo((mask), (add))
#define __builtin_amdgcn_mbcnt_hi(mask, add) optik_emu::mbcnt_hi((mask), (add))

// LDS hand-over points: fence + wave barrier on the device, a real barrier between the threads here
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() optik_emu::barrier()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)

// ---- scalar intrinsics the device headers use ---------------------------------------------------

inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __double2hiint(double d) { return (int)(optik_emu::to_bits(d) >> 32); }
inline int __double2loint(double d) { return (int)(optik_emu::to_bits(d) & 0xffffffffull); }
inline double __hiloint2double(int hi, int lo) {
    return optik_emu::from_bits<double>(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
inline double __longlong_as_double(long long v) { return optik_emu::from_bits<double>((unsigned long long)v); }
inline unsigned long long wall_clock64() { return 0ull; }

inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) {
    unsigned long long cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return cur;
}
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_SYSTEM 1
#define __hip_atomic_load(ptr, order, scope) __atomic_load_n((ptr), (order))
#define __hip_atomic_store(ptr, val, order, scope) __atomic_store_n((ptr), (val), (order))
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
