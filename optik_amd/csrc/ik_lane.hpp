// ik_lane.hpp -- cross-lane primitives of a QUAD: four adjacent lanes that share one restart.
//
// gfx950: a value moves between the lanes of a quad with DPP quad_perm row moves -- two
// v_mov_b32_dpp per double, no LDS round trip (ds_bpermute costs an LDS issue and ~60 cycles of
// latency per value).  The source lane must be a compile-time constant for DPP; every call site
// has one after unrolling (joint k lives in lane k & 3).
#pragma once

#include "ik_math.hpp"

namespace optik {

constexpr int QUAD = 4;

OPTIK_DEV int quad_lane() { return (int)(threadIdx.x & 3u); }
OPTIK_DEV int quad_base() { return (int)(threadIdx.x & 63u) & ~3; }

// The lane's number within its wave, produced where it is wanted (two mbcnt instructions) instead of
// being carried from the kernel's entry: what a region of the solver loop derives from it -- joint
// numbers, masks, LDS addresses -- is then computed in that region, not hoisted out of the loop as
// an invariant and spilled for its whole length.
OPTIK_DEV int wave_lane_now() {
    int v = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#ifndef OPTIK_NO_LAUNDER
    asm volatile("" : OPTIK_REG_INOUT(v));
#endif
    return v;
}
OPTIK_DEV int quad_lane_now() { return wave_lane_now() & 3; }

// (bound_ctrl = true: every lane of a full wave has a valid source, so no "old" value has to be
// materialised in the destination first -- with false the compiler emits a v_mov 0 before every DPP move)
template <int K>
OPTIK_DEV double quad_get_c(double v) {
    return __builtin_amdgcn_update_dpp(0.0, v, K * 0x55, 0xf, 0xf, true);  // quad_perm:[K,K,K,K]
}
template <int K>
OPTIK_DEV int quad_get_c(int v) {
    return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xf, 0xf, true);
}
// value held by lane k & 3 of the caller's quad (k: a constant after unrolling)
OPTIK_DEV double quad_get(double v, int k) {
    switch (k & 3) {
    case 0: return quad_get_c<0>(v);
    case 1: return quad_get_c<1>(v);
    case 2: return quad_get_c<2>(v);
    default: return quad_get_c<3>(v);
    }
}
OPTIK_DEV int quad_get(int v, int k) {
    switch (k & 3) {
    case 0: return quad_get_c<0>(v);
    case 1: return quad_get_c<1>(v);
    case 2: return quad_get_c<2>(v);
    default: return quad_get_c<3>(v);
    }
}

// value held by lane (own ^ mask) of the caller's quad, mask = 1 or 2: the butterflies of a quad-wide
// min / argmax.  DPP quad_perm [1,0,3,2] / [2,3,0,1] on the device (ds_bpermute costs an LDS round trip).
OPTIK_DEV double quad_xor(double v, int mask) {
    return mask == 1 ? __builtin_amdgcn_update_dpp(0.0, v, 0xB1, 0xf, 0xf, true)
                     : __builtin_amdgcn_update_dpp(0.0, v, 0x4E, 0xf, 0xf, true);
}
OPTIK_DEV int quad_xor(int v, int mask) {
    return mask == 1 ? __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true)
                     : __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
}

// value held by lane (own + r) & 3 of the caller's quad, r = 1, 2, 3 (a rotation of the quad): the
// diagonal rounds of a ChaCha block whose columns live in the four lanes.  quad_perm [1,2,3,0] /
// [2,3,0,1] / [3,0,1,2].
OPTIK_DEV uint32_t quad_rot(uint32_t v, int r) {
    const int i = (int)v;
    return (uint32_t)(r == 1 ? __builtin_amdgcn_update_dpp(0, i, 0x39, 0xf, 0xf, true)
                             : (r == 2 ? __builtin_amdgcn_update_dpp(0, i, 0x4E, 0xf, 0xf, true)
                                       : __builtin_amdgcn_update_dpp(0, i, 0x93, 0xf, 0xf, true)));
}

// does p hold in some / every lane of the caller's quad
OPTIK_DEV bool quad_any(bool p) {
    const unsigned long long m = __ballot(p);
    return ((m >> quad_base()) & 0xfull) != 0ull;
}
OPTIK_DEV bool quad_all(bool p) { return !quad_any(!p); }

// LDS written by some lanes of the wave is about to be read by others
OPTIK_DEV void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace optik
