// ik_wide_launch.hpp -- chains with 9 .. 16 joint positions (ik_wide.hpp): the chain table and what
// the launches of ik_wide_kernel.hip receive.  Plain data shared by that translation unit and the
// host code of ik_capi.hip / ik_batch_ops.hip.
#pragma once

#include <hip/hip_runtime.h>

#include "ik_solve.hpp"

namespace optik {

constexpr int WIDE_MAX_DOF = 16;

// Chain table of a wide chain (HBM once per robot, staged into LDS by every kernel).
struct WideChainDev {
    int32_t n_pos;
    int32_t has_tip;
    int32_t pad0, pad1;
    double origin[WIDE_MAX_DOF + 1][7];
    double axis[WIDE_MAX_DOF][3];
    double lb[WIDE_MAX_DOF];
    double ub[WIDE_MAX_DOF];
    double scale[WIDE_MAX_DOF];  // rand UniformFloat scale per joint
};

struct WideSolveLaunch {
    const WideChainDev *chain;
    EvalParams ep;
    SolveParams sp;
    uint32_t key[8];  // ChaCha key = seed_from_u64(42)
    WorkQueue wq;
    unsigned long long deadline_ticks;  // relative to kernel start, 0 = none
    double *ws;                         // [grid][wide_ws_doubles_per_wave()]
};

struct WideBatchLaunch {
    const WideChainDev *chain;
    EvalParams ep;
    double target[7];
    uint32_t key[8];
    unsigned long long first;  // seeds: first restart index
    const double *q;           // [n][B]
    long long B;
    double *f;                 // eval: [B]
    double *g;                 // eval: [n][B] or null
    double *pose;              // fk: [7][B]
    double *jac;               // fk: [6n][B] or null
    double *q_out;             // seeds: [n][B]
};

// doubles of workspace one resident wave needs (64 lanes x the slots of a restart)
size_t wide_ws_doubles_per_wave();
// `grid` single-wave workgroups pulling work items until the queue is dry.  lds_form: one restart per
// wave with its arrays in LDS (a.wq.lanes must be 1; a.ws is not used) -- the latency form
hipError_t wide_solve_launch(int grid, hipStream_t stream, const WideSolveLaunch &a, bool lds_form, bool lds_coop = true);
int wide_lds_bytes();  // static LDS of the latency form
// op 0: objective + gradient, 1: forward kinematics (+ body Jacobian), 2: restart seeds
hipError_t wide_batch_launch(int op, int grid, hipStream_t stream, const WideBatchLaunch &a);

}  // namespace optik
