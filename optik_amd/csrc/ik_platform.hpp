// ik_platform.hpp -- the ONE place where the device headers know about tests/emu.
//
// The library is never built with OPTIK_LANE_EMU.  tests/emu/quad_emu.cpp compiles the quad and lane-per-restart
// solvers for the host with the wave emulated by one thread per lane (tests/emu/lane_emu.hpp): there the HIP runtime
// header is replaced by the emulation -- which also provides the cross-lane builtins this code uses
// (__builtin_amdgcn_update_dpp with quad_perm controls, mbcnt, ballot, shuffles, fences) under their device
// names and with their device semantics, so ik_lane.hpp and everything above it is the same text on both sides --
// and the three things below differ: the register class an empty asm statement can launder a value through, how a
// pointer into LDS keeps its address space through that laundering, and the shape of the (partial) wave.
#pragma once

#ifdef OPTIK_LANE_EMU
#include "lane_emu.hpp"
#define OPTIK_REG_INOUT "+r"
#else
#include <hip/hip_runtime.h>
#define OPTIK_REG_INOUT "+v"
#endif
#include <stdint.h>

// wave cycles per phase are only counted on the device (-DOPTIK_PROFILE builds)
#if defined(OPTIK_PROFILE) && !defined(OPTIK_LANE_EMU)
#define OPTIK_DEVICE_PROFILE 1
#endif

namespace optik {

// quads of lanes the wave has, and the number of the wave's workgroup (the emulation runs one partial wave)
__device__ __forceinline__ int wave_quads() {
#ifdef OPTIK_LANE_EMU
    return optik_emu::t_wave->lanes / 4;
#else
    return 16;
#endif
}

// A pointer INTO LDS made opaque to the optimiser: only the offset is laundered, the address space is kept.  (A
// generic pointer laundered whole comes back as FLAT accesses: the LDS is reached through the aperture check of
// the vector-memory path, every access counts on both wait counters, and nothing the LDS returns can be waited for
// selectively.)
template <class T>
__device__ __forceinline__ const T *launder_lds(const T *p) {
#ifdef OPTIK_LANE_EMU
    asm volatile("" : "+r"(p));
    return p;
#else
    typedef const T __attribute__((address_space(3))) *lds_cptr;
    unsigned off = (unsigned)(__UINTPTR_TYPE__)(lds_cptr)p;
    asm volatile("" : "+v"(off));
    return (const T *)(lds_cptr)(__UINTPTR_TYPE__)off;
#endif
}

}  // namespace optik
