// ik_host.hpp -- what the host-side translation units of the kernel layer share: the chain handle, the tuning
// options, error reporting and the (n, trailing fixed joint) dispatch.
//
//   ik_capi.hip        chains, options, optik_hip_ik_batch / optik_hip_ik_host, timing (the C ABI of optik_hip.h)
//   ik_select.hip      the selection of lib.rs:397-413 over the per-restart keys
//   ik_batch_ops.hip   objective / gradient, FK / Jacobian and seed batches, the test probes
//   ik_lane_kernel.hip, ik_quad_kernel.hip, ik_wide_kernel.hip    the restart solvers (one restart loop each)
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/optik_hip.h"
#include "device_scope.hpp"
#include "ik_host_params.hpp"
#include "ik_launch.hpp"
#include "ik_wide_launch.hpp"

namespace optik {
namespace host {

constexpr int WAVE = 64;
constexpr int QUADS_PER_WAVE_HOST = 16;  // restarts a wave of the quad solver holds

// ---- selection (ik_select.hip) -----------------------------------------------------------------------------
struct TileRec {
    unsigned long long idx;  // winning restart index in the tile, ~0 if none
    double key;
};

struct SelectLaunch {
    const double *out_key;   // [T*R] selection key, +inf unless the restart succeeded
    const double *out_x;     // [n][T*R]
    const double *out_f;     // [T*R]
    TileRec *tile_recs;      // [T][tiles_per_target]
    int tiles_per_target;
    int tile;                // restarts per tile
    int n;
    int pad;
    unsigned long long restart_begin;
    unsigned long long n_restarts;
    size_t ld;               // T * R
    double *win_x;           // [T][n]
    double *win_f;
    unsigned long long *win_idx;
    double *win_key;
    // the launch's work-item counter and first-success words, put back to their initial values by the last
    // kernel of the launch so that the next launch needs no fill commands in front of it (null: leave them)
    unsigned long long *reset_queue;
    unsigned long long *reset_fs;  // [T]
};
constexpr int SEL_TILE = 4096;  // restarts per 256-thread selection block
// per-tile argmin + per-target reduction (one kernel when a target has a single tile); T blocks publish the winners
hipError_t select_launch(const SelectLaunch &s, int T, hipStream_t stream);

// ---- options ---------------------------------------------------------------------------------------------
// Every tuning option of the kernel layer, in one place.  The defaults come from the environment ONCE, at the
// first use (the OPTIK_* names below); tests and tools change them through optik_hip_set_option (optik_hip.h).
// Nothing else in this library reads the environment (robot_host.cpp: OPTIK_HOST_THREADS, OPTIK_DEVICES).
enum : int { SK_AUTO = 0, SK_QUAD = 1, SK_LANE64 = 2, SK_GENERAL = 3 };
struct Options {
    int solve_kernel = SK_AUTO;      // OPTIK_SOLVE_KERNEL = quad | lane64 | general: which single-launch solver (auto: by size)
    int wide_form = 0;               // OPTIK_WIDE_FORM = lds | hbm: the general solver's form (9 .. 16 joints); 2: one-lane LDS form
    int range_rule = OPTIK_HIP_RANGE_SINGLE_INCLUSIVE;  // OPTIK_RANDOM_RANGE_RULE = new_inclusive: rand 0.9.2 reading of new chains
    int stop_x_legacy = 0;           // (no environment name) nlopt_stop_x of NLopt 2.5: no zero-step rule
};
Options &opt();  // (ik_capi.hip)

// ---- errors ----------------------------------------------------------------------------------------------
extern thread_local std::string g_err;  // optik_hip_last_error() of the calling thread (ik_capi.hip)
inline int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return ::optik::host::fail(OPTIK_HIP_ENODEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

inline int ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(OPTIK_HIP_ENODEVICE, std::string("no HIP device available: ")
                                             + (e == hipSuccess ? "device count is 0" : hipGetErrorString(e)));
    return 0;
}

// A chain lives on the device that was current when it was created; its entry points make that
// device current for the calling thread for the duration of the call and restore the caller's
// device on every exit path (device_scope.hpp) -- a host that also drives torch / RCCL on the
// thread finds its own device current again.
#define BIND_DEVICE(CH)                                                                 \
    optik::DeviceScope dev_scope_((CH)->device_id);                                     \
    if (!dev_scope_.ok()) return ::optik::host::fail(OPTIK_HIP_ENODEVICE, "hipSetDevice(" + std::to_string((CH)->device_id) + ") failed")

// Dispatch on (n, trailing fixed joint): kernels are instantiated for 1 <= n <= 8 revolute joints, each with and
// without a trailing fixed joint.
#define OPTIK_N_RANGE_MSG "this kernel is built for 1 <= n <= 8 revolute joints"
#define OPTIK_DISPATCH_ONE(NN, CALL)                                                   \
    if (!done_ && n_ == NN) {                                                          \
        if (tip_) { CALL(NN, true); } else { CALL(NN, false); }                        \
        done_ = true;                                                                  \
    }
#define OPTIK_DISPATCH(CH, CALL)                                                       \
    do {                                                                               \
        const int n_ = (CH)->n;                                                        \
        const bool tip_ = (CH)->tip;                                                   \
        bool done_ = false;                                                            \
        OPTIK_DISPATCH_ONE(1, CALL)                                                     \
        OPTIK_DISPATCH_ONE(2, CALL) OPTIK_DISPATCH_ONE(3, CALL) OPTIK_DISPATCH_ONE(4, CALL) \
        OPTIK_DISPATCH_ONE(5, CALL) OPTIK_DISPATCH_ONE(6, CALL) OPTIK_DISPATCH_ONE(7, CALL) \
        OPTIK_DISPATCH_ONE(8, CALL)                                                     \
        if (!done_) return ::optik::host::fail(OPTIK_HIP_EUNSUPPORTED, OPTIK_N_RANGE_MSG); \
    } while (0)

}  // namespace host
}  // namespace optik

// ---- the chain handle ---------------------------------------------------------------------------------------
struct optik_hip_chain {
    optik::ChainDev host;
    optik::ChainDev *dev = nullptr;
    int n = 0;
    bool tip = false;
    uint32_t key[8];
    double scale[optik::WIDE_MAX_DOF];
    int range_rule = 0;  // OPTIK_HIP_RANGE_*: how `scale` was formed
    // a chain with 9 .. 16 joint positions (ik_wide.hpp): its own table, the general kernels
    bool wide = false;
    optik::WideChainDev whost;
    optik::WideChainDev *wdev = nullptr;
    double *wide_ws = nullptr;  // restart workspace of the resident waves
    size_t wide_ws_waves = 0;
    int device_id = 0;   // the HIP device the chain lives on (the current device at creation)
    // a chain with prismatic joints: FK only (as in the reference); the joint table for fk_general_kernel
    bool prismatic = false;
    int n_joints = 0;
    int32_t types[optik::MAX_JOINTS] = {};
    double axis_all[optik::MAX_JOINTS][3] = {};
    // launch workspace (grown on demand; one in-flight ik call per chain handle)
    std::mutex mu;
    std::mutex host_mu;  // serialises optik_hip_ik_host calls (they share the staging blocks below)
    optik::host::TileRec *tile_recs = nullptr;
    size_t tile_cap = 0;
    unsigned long long *first_success = nullptr;
    size_t fs_cap = 0;
    // (what the last launch's selection kernel left behind: the work-item counter at 0, this many leading
    // first-success words at ~0 -- a launch that finds them so skips its fill commands)
    // (host-side knowledge that holds for launches ORDERED behind that selection kernel: the stream it ran on is kept
    // with it, a launch on any other stream fills the words itself)
    bool queue_clean = false;
    size_t fs_clean = 0;
    hipStream_t clean_stream = nullptr;
    // scratch per-restart buffers when the caller does not provide them
    double *tmp_x = nullptr, *tmp_f = nullptr, *tmp_key = nullptr;
    size_t tmp_cols = 0;
    unsigned long long *queue = nullptr;  // work-item counter of the in-flight launch
    unsigned long long *prof = nullptr;   // phase timers (OPTIK_PROFILE builds)
    double *hw_dev = nullptr, *hw_pin = nullptr;  // optik_hip_ik_host: device block and pinned staging
    size_t hw_cap = 0;                            // doubles
    // optik_hip_ik_host, one target under the first-success rule: the block the first successful restart writes its
    // answer to (WorkQueue::claim; pinned, host-coherent) and the sequence number of the last launch that used it
    unsigned long long *hw_claim = nullptr;
    hipEvent_t claim_done = nullptr;  // recorded behind such a launch: what the polling host also looks at
    unsigned long long claim_seq = 0;
    // such a launch may still be running on the null stream (set, under `mu`, in the critical section that queues it;
    // cleared by whoever has waited for the null stream)
    bool claim_pending = false;
    unsigned hw_flip = 0;  // which half of the pinned block the next zero-copy call uses
    // timing
    int timing = 0;
    static constexpr int EV_POOL = 256;  // event pairs recorded round-robin around the solve kernel
    hipEvent_t ev0[EV_POOL] = {}, ev1[EV_POOL] = {};
    int ev_count = 0;                    // launches recorded since the last reset
    optik_hip_launch_info last{};
    int num_cus = 0;
    int wall_clock_khz = 0;
};

namespace optik {
namespace host {

inline int grid_for(const optik_hip_chain *ch, long long work, int block, int per_cu) {
    long long blocks = (work + block - 1) / block;
    const long long cap = (long long)(ch->num_cus > 0 ? ch->num_cus : 256) * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace host
}  // namespace optik
