// ik_host_params.hpp -- host-side derivation of what a launch passes to the kernels as wave-uniform
// parameters: the ChaCha key of seed_from_u64(42), rand's uniform scale per joint, the objective
// weights with nalgebra's is_identity decision, NLopt's stopping parameters.  Plain C++ (no HIP):
// shared by ik_capi.hip / ik_batch_ops.hip and by the host emulation of the quad solver under tests/emu/.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../include/optik_hip.h"
#include "ik_solve.hpp"

namespace optik {
namespace hostparams {

// rand_core 0.9 SeedableRng::seed_from_u64 (PCG32 expansion).
inline void seed_from_u64(uint64_t state, uint32_t key[8]) {
    const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
    for (int i = 0; i < 8; ++i) {
        state = state * MUL + INC;
        const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
}

// Scale of `rng.random_range(lb..=ub)` (lib.rs:89; rand 0.9.2, not vendored).  The call chain
// Rng::random_range -> SampleRange for RangeInclusive<f64> -> UniformFloat::sample_single_inclusive
// ends in `scale = high - low` (rule OPTIK_HIP_RANGE_SINGLE_INCLUSIVE, the default).  The other
// reading -- Uniform::new_inclusive(lo, hi).sample(rng): (high - low) / (1 - eps) with the 1-ulp
// decrease loop -- stays selectable (optik_hip_chain_set_range_rule / OPTIK_RANDOM_RANGE_RULE):
// the kernels only ever see the precomputed scale, so the choice is host-side data.
inline double uniform_scale(double low, double high, int rule) {
    if (rule == OPTIK_HIP_RANGE_SINGLE_INCLUSIVE) return high - low;
    const double max_rand = 1.0 - 2.220446049250313e-16;
    double scale = (high - low) / max_rand;
    while (scale * max_rand + low > high) {
        uint64_t u;
        std::memcpy(&u, &scale, 8);
        u -= 1;
        std::memcpy(&scale, &u, 8);
    }
    return scale;
}

// approx::relative_eq!(a, b, epsilon = eps), default max_relative = f64::EPSILON.
inline bool relative_eq(double a, double b, double eps) {
    if (a == b) return true;
    if (std::isinf(a) || std::isinf(b)) return false;
    const double d = std::fabs(a - b);
    if (d <= eps) return true;
    const double largest = std::fmax(std::fabs(a), std::fabs(b));
    return d <= largest * 2.220446049250313e-16;
}

// nalgebra is_identity on a 3-vector (objective.rs:13,25; quirk Q2).
inline bool vec3_is_identity(const double w[3]) {
    const double eps = 1e-20;
    return relative_eq(w[0], 1.0, eps) && relative_eq(w[1], 0.0, eps) && relative_eq(w[2], 0.0, eps);
}

inline void make_eval_params(const double wl[3], const double wa[3], const double *ee_offset7, EvalParams &ep) {
    std::memset(&ep, 0, sizeof ep);
    for (int i = 0; i < 3; ++i) {
        ep.w_lin[i] = wl[i];
        ep.w_ang[i] = wa[i];
        ep.w_lin2[i] = wl[i] * wl[i];  // objective.rs:102-103
        ep.w_ang2[i] = wa[i] * wa[i];
    }
    ep.skip_lin = vec3_is_identity(ep.w_lin);
    ep.skip_ang = vec3_is_identity(ep.w_ang);
    ep.skip_lin2 = vec3_is_identity(ep.w_lin2);
    ep.skip_ang2 = vec3_is_identity(ep.w_ang2);
    ep.grad_same_as_value = std::memcmp(ep.w_lin, ep.w_lin2, sizeof ep.w_lin) == 0
                            && std::memcmp(ep.w_ang, ep.w_ang2, sizeof ep.w_ang) == 0;
    const double ident[7] = {0, 0, 0, 0, 0, 0, 1};
    ep.has_ee_offset = ee_offset7 && std::memcmp(ee_offset7, ident, sizeof ident) != 0;
    std::memcpy(ep.ee_offset, ee_offset7 ? ee_offset7 : ident, sizeof ident);
}

inline void fill_solve_params(const optik_solver_config *cfg, SolveParams &sp, bool stop_x_legacy = false) {
    sp.stopval = cfg->tol_f;
    sp.ftol_abs = (cfg->tol_df > 0.0) ? cfg->tol_df : 1e-3 * cfg->tol_f;  // lib.rs:283-293
    sp.xtol_abs = cfg->tol_dx;
    sp.ok_stopval = cfg->tol_f >= 0.0;
    sp.ok_ftol = cfg->tol_df >= 0.0;
    sp.ok_xtol = cfg->tol_dx >= 0.0;
    // nlopt_stop_x of the bundled NLopt (2.7.1): a zero step counts as x-converged; the 2.5
    // behaviour (per-coordinate test only) for anyone pinning against an older build
    sp.stop_x_zero = stop_x_legacy ? 0 : 1;
}



}  // namespace hostparams
}  // namespace optik
