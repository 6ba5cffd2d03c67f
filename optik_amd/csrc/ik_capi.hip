// ik_capi.hip -- the C ABI of include/optik_hip.h: chains, tuning options, the restart launch (optik_hip_ik_batch),
// its host-buffer form (optik_hip_ik_host), timing.
//
// The solvers are launched from here and defined in their own translation units: the lane-per-restart form
// (ik_lane_kernel.hip), the quad solver (ik_quad_kernel.hip), the general run-time-n solver (ik_wide_kernel.hip);
// optik_hip_ik_batch picks one by launch size and joint count.  The selection kernels: ik_select.hip; the batch
// operators: ik_batch_ops.hip.  No CPU fallback exists: every entry point fails loudly without a device.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ik_host.hpp"

using namespace optik;
using namespace optik::host;
using namespace optik::hostparams;

namespace optik {
namespace host {

thread_local std::string g_err;

static int solve_kernel_from(const char *e) {
    if (!e) return SK_AUTO;
    if (!std::strcmp(e, "quad")) return SK_QUAD;
    if (!std::strcmp(e, "lane64")) return SK_LANE64;
    if (!std::strcmp(e, "general")) return SK_GENERAL;
    return SK_AUTO;
}
Options &opt() {
    static Options o = [] {
        Options v;
        v.solve_kernel = solve_kernel_from(std::getenv("OPTIK_SOLVE_KERNEL"));
        if (const char *e = std::getenv("OPTIK_WIDE_FORM")) v.wide_form = std::strcmp(e, "hbm") == 0 ? 1 : 0;
        if (const char *e = std::getenv("OPTIK_RANDOM_RANGE_RULE"))
            if (std::strcmp(e, "new_inclusive") == 0 || std::strcmp(e, "1") == 0) v.range_rule = OPTIK_HIP_RANGE_NEW_INCLUSIVE;
        return v;
    }();
    return o;
}

}  // namespace host
}  // namespace optik

namespace {

int default_range_rule() { return opt().range_rule; }

void set_chain_scales(optik_hip_chain *ch, int rule) {
    ch->range_rule = rule;
    for (int k = 0; k < ch->n; ++k) {
        const double lb = ch->wide ? ch->whost.lb[k] : ch->host.lb[k], ub = ch->wide ? ch->whost.ub[k] : ch->host.ub[k];
        // infinite limits (continuous joints) make random_range panic in the
        // reference (quirk Q5); restarts > 0 are refused at launch time instead.
        ch->scale[k] = (std::isfinite(lb) && std::isfinite(ub)) ? uniform_scale(lb, ub, rule) : NAN;
        if (ch->wide) ch->whost.scale[k] = ch->scale[k];
    }
}

}  // namespace

extern "C" {

int optik_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *optik_hip_last_error(void) { return g_err.c_str(); }

int optik_hip_chain_create(const double *origins, const double *axes, const int32_t *types,
                           int32_t n_joints, const double *lb, const double *ub, int32_t n,
                           optik_hip_chain **out) {
    if (!origins || !axes || !types || !lb || !ub || !out) return fail(OPTIK_HIP_EINVAL, "null argument");
    if (n < 1 || n > WIDE_MAX_DOF) return fail(OPTIK_HIP_EUNSUPPORTED, "num_positions must be in 1..16");
    if (n_joints != n && n_joints != n + 1)
        return fail(OPTIK_HIP_EUNSUPPORTED, "chain must be n revolute joints plus an optional trailing fixed joint");
    bool prismatic = false;
    for (int j = 0; j < n; ++j) {
        if (types[j] == OPTIK_JOINT_PRISMATIC) prismatic = true;
        else if (types[j] != OPTIK_JOINT_REVOLUTE)
            return fail(OPTIK_HIP_EUNSUPPORTED, "the first n joints of the chain must be revolute or prismatic");
    }
    if (n_joints == n + 1 && types[n] != OPTIK_JOINT_FIXED)
        return fail(OPTIK_HIP_EUNSUPPORTED, "joint after the last revolute joint must be fixed");
    if (prismatic && n > MAX_DOF)
        return fail(OPTIK_HIP_EUNSUPPORTED, "prismatic joints are supported for chains of at most 8 joint positions");
    if (int rc = ensure_device()) return rc;

    auto *ch = new optik_hip_chain();
    std::memset(&ch->host, 0, sizeof ch->host);
    std::memset(&ch->whost, 0, sizeof ch->whost);
    ch->n = n;
    if (n > MAX_DOF) {
        // 9 .. 16 joint positions: the general kernels of ik_wide.hpp (one table, joint count at run time)
        ch->wide = true;
        ch->n_joints = n_joints;
        ch->tip = (n_joints == n + 1);
        WideChainDev &w = ch->whost;
        w.n_pos = n;
        w.has_tip = ch->tip;
        for (int j = 0; j < n_joints; ++j)
            for (int k = 0; k < 7; ++k) w.origin[j][k] = origins[j * 7 + k];
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < 3; ++k) w.axis[j][k] = axes[j * 3 + k];
        for (int k = 0; k < n; ++k) { w.lb[k] = lb[k]; w.ub[k] = ub[k]; }
        set_chain_scales(ch, default_range_rule());
        seed_from_u64(42, ch->key);  // RNG_SEED, lib.rs:360
        hipError_t e = hipMalloc(&ch->wdev, sizeof(WideChainDev));
        if (e == hipSuccess) e = hipMemcpy(ch->wdev, &ch->whost, sizeof(WideChainDev), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (ch->wdev) (void)hipFree(ch->wdev);
            delete ch;
            return fail(OPTIK_HIP_ENODEVICE, std::string("chain upload: ") + hipGetErrorString(e));
        }
        int dev = 0;
        hipGetDevice(&dev);
        ch->device_id = dev;
        hipDeviceGetAttribute(&ch->num_cus, hipDeviceAttributeMultiprocessorCount, dev);
        hipDeviceGetAttribute(&ch->wall_clock_khz, hipDeviceAttributeWallClockRate, dev);
        *out = ch;
        return 0;
    }
    ch->prismatic = prismatic;
    ch->n_joints = n_joints;
    for (int j = 0; j < n_joints; ++j) {
        ch->types[j] = types[j];
        for (int k = 0; k < 3; ++k) ch->axis_all[j][k] = axes[j * 3 + k];
    }
    ch->tip = (n_joints == n + 1);
    ch->host.n_pos = n;
    ch->host.has_tip = ch->tip;
    for (int j = 0; j < n_joints; ++j)
        for (int k = 0; k < 7; ++k) ch->host.origin[j][k] = origins[j * 7 + k];
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < 3; ++k) ch->host.axis[j][k] = axes[j * 3 + k];
    for (int k = 0; k < n; ++k) {
        ch->host.lb[k] = lb[k];
        ch->host.ub[k] = ub[k];
    }
    set_chain_scales(ch, default_range_rule());
    seed_from_u64(42, ch->key);  // RNG_SEED, lib.rs:360
    hipError_t e = hipMalloc(&ch->dev, sizeof(ChainDev));
    if (e == hipSuccess) e = hipMemcpy(ch->dev, &ch->host, sizeof(ChainDev), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        delete ch;
        return fail(OPTIK_HIP_ENODEVICE, std::string("chain upload: ") + hipGetErrorString(e));
    }
    int dev = 0;
    hipGetDevice(&dev);
    ch->device_id = dev;
    hipDeviceGetAttribute(&ch->num_cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&ch->wall_clock_khz, hipDeviceAttributeWallClockRate, dev);
    *out = ch;
    return 0;
}

void optik_hip_chain_destroy(optik_hip_chain *ch) {
    if (!ch) return;
    optik::DeviceScope dev_scope(ch->device_id);  // (the frees run on the chain's device)
    if (ch->claim_pending) (void)hipStreamSynchronize(nullptr);
    if (ch->dev) hipFree(ch->dev);
    if (ch->wdev) hipFree(ch->wdev);
    if (ch->wide_ws) hipFree(ch->wide_ws);
    if (ch->tile_recs) hipFree(ch->tile_recs);
    if (ch->first_success) hipFree(ch->first_success);
    if (ch->tmp_x) hipFree(ch->tmp_x);
    if (ch->tmp_f) hipFree(ch->tmp_f);
    if (ch->tmp_key) hipFree(ch->tmp_key);
    if (ch->queue) hipFree(ch->queue);
    if (ch->hw_dev) hipFree(ch->hw_dev);
    if (ch->hw_pin) hipHostFree(ch->hw_pin);
    if (ch->hw_claim) hipHostFree(ch->hw_claim);
    if (ch->claim_done) hipEventDestroy(ch->claim_done);
    for (int i = 0; i < optik_hip_chain::EV_POOL; ++i) {
        if (ch->ev0[i]) hipEventDestroy(ch->ev0[i]);
        if (ch->ev1[i]) hipEventDestroy(ch->ev1[i]);
    }
    delete ch;
}

int32_t optik_hip_chain_num_positions(const optik_hip_chain *ch) { return ch ? ch->n : 0; }

int optik_hip_chain_set_range_rule(optik_hip_chain *ch, int32_t rule) {
    if (!ch || (rule != OPTIK_HIP_RANGE_SINGLE_INCLUSIVE && rule != OPTIK_HIP_RANGE_NEW_INCLUSIVE))
        return fail(OPTIK_HIP_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lock(ch->mu);
    set_chain_scales(ch, rule);
    if (ch->wide) {  // (the scales of a wide chain are part of its device table)
        BIND_DEVICE(ch);
        HIP_TRY(hipMemcpy(ch->wdev, &ch->whost, sizeof(WideChainDev), hipMemcpyHostToDevice));
    }
    return 0;
}

int32_t optik_hip_chain_range_rule(const optik_hip_chain *ch) { return ch ? ch->range_rule : -1; }

}  // extern "C"

// optik_hip_ik_batch with the chain's launch mutex already held.
static int ik_batch_locked(optik_hip_chain *ch, const optik_solver_config *cfg, const double *d_targets,
                           const double *d_x0, int32_t T, const double *ee_offset7, uint64_t restart_begin,
                           uint64_t restart_end, uint32_t flags, double deadline_s, const optik_hip_ik_outputs *out,
                           void *stream_v, bool claim_request = false, bool *claim_armed = nullptr);

extern "C" {

int optik_hip_ik_batch(optik_hip_chain *ch, const optik_solver_config *cfg, const double *d_targets,
                       const double *d_x0, int32_t T, const double *ee_offset7, uint64_t restart_begin,
                       uint64_t restart_end, uint32_t flags, double deadline_s, const optik_hip_ik_outputs *out,
                       void *stream_v) {
    if (!ch) return fail(OPTIK_HIP_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lock(ch->mu);
    return ik_batch_locked(ch, cfg, d_targets, d_x0, T, ee_offset7, restart_begin, restart_end, flags, deadline_s, out,
                           stream_v);
}

}  // extern "C"

static int ik_batch_locked(optik_hip_chain *ch, const optik_solver_config *cfg, const double *d_targets,
                           const double *d_x0, int32_t T, const double *ee_offset7, uint64_t restart_begin,
                           uint64_t restart_end, uint32_t flags, double deadline_s, const optik_hip_ik_outputs *out,
                           void *stream_v, bool claim_request, bool *claim_armed) {
    if (claim_armed) *claim_armed = false;
    if (!ch || !cfg || !d_targets || !d_x0 || !out || T < 1) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (restart_end <= restart_begin) return fail(OPTIK_HIP_EINVAL, "empty restart range");
    if (cfg->solution_mode != 1 && cfg->solution_mode != 2)
        return fail(OPTIK_HIP_EINVAL, "solution_mode must be 1 (Quality) or 2 (Speed)");
    const uint64_t R = restart_end - restart_begin;
    if (restart_end > 1 || restart_begin > 0)
        for (int k = 0; k < ch->n; ++k)
            if (std::isnan(ch->scale[k]))
                return fail(OPTIK_HIP_EINVAL, "random restarts need finite joint limits (reference: random_range panics)");
    hipStream_t stream = (hipStream_t)stream_v;
    if (ch->prismatic)
        return fail(OPTIK_HIP_EUNSUPPORTED,
                    "prismatic joints: only forward kinematics is available (the reference's Jacobian panics, kinematics.rs:185)");
    BIND_DEVICE(ch);
    // (a single call that returned on its first success may have left its launch running on the null stream: a
    // launch on another stream shares the chain's workspace with it and waits; on the null stream it queues behind)
    // (on the null stream the flag stays: optik_hip_ik_host's staged path still has to know)
    if (ch->claim_pending && stream != nullptr) { HIP_TRY(hipStreamSynchronize(nullptr)); ch->claim_pending = false; }

    // selection tiles: 4096 restarts per 256-thread block
    const uint64_t tiles_per_target = (R + SEL_TILE - 1) / SEL_TILE;
    const uint64_t n_tiles64 = tiles_per_target * (uint64_t)T;
    // (HIP rejects a launch whose grid.x * block.x reaches 2^32: 256-thread tile blocks cap the tiles at 2^24 - 1)
    if (n_tiles64 * 256ull >= (1ull << 32))
        return fail(OPTIK_HIP_EINVAL, "too many restarts / targets in one launch (2^24 or more selection tiles)");
    const int n_tiles = (int)n_tiles64;
    const size_t cols = (size_t)T * (size_t)R;

    if ((size_t)n_tiles > ch->tile_cap) {
        if (ch->tile_recs) HIP_TRY(hipFree(ch->tile_recs));
        ch->tile_recs = nullptr;
        HIP_TRY(hipMalloc(&ch->tile_recs, sizeof(TileRec) * (size_t)n_tiles));
        ch->tile_cap = (size_t)n_tiles;
    }
    if (!ch->queue) { HIP_TRY(hipMalloc(&ch->queue, sizeof(unsigned long long))); ch->queue_clean = false; }
    // (the words are only known to be clean to a launch queued behind the selection kernel that cleaned them)
    if (stream != ch->clean_stream) { ch->queue_clean = false; ch->fs_clean = 0; }
    if (!ch->queue_clean) HIP_TRY(hipMemsetAsync(ch->queue, 0, sizeof(unsigned long long), stream));
    ch->queue_clean = false;  // (until this launch's selection kernel has put it back)
    const bool early = (flags & OPTIK_HIP_IK_EARLY_EXIT) && cfg->solution_mode == 2;
    size_t fs_clean_after = ch->fs_clean;  // (a launch without early exit leaves the words alone)
    if (early) {
        if ((size_t)T > ch->fs_cap) {
            if (ch->first_success) HIP_TRY(hipFree(ch->first_success));
            ch->first_success = nullptr;
            ch->fs_clean = 0;
            HIP_TRY(hipMalloc(&ch->first_success, sizeof(unsigned long long) * (size_t)T));
            ch->fs_cap = (size_t)T;
        }
        if (ch->fs_clean < (size_t)T)
            HIP_TRY(hipMemsetAsync(ch->first_success, 0xff, sizeof(unsigned long long) * (size_t)T, stream));
        fs_clean_after = std::max(ch->fs_clean, (size_t)T);  // once the selection kernel has put words [0, T) back
        ch->fs_clean = 0;
    }
    // the selection needs the per-restart x / f / key: scratch if the caller skips them
    const bool want_win = out->d_win_x || out->d_win_f || out->d_win_idx || out->d_win_key;
    double *px = out->d_x, *pf = out->d_f, *pk = nullptr;
    if (want_win) {
        const bool need_xf = (!px || !pf) && (out->d_win_x || out->d_win_f);
        if (cols > ch->tmp_cols) {
            if (ch->tmp_x) HIP_TRY(hipFree(ch->tmp_x));
            if (ch->tmp_f) HIP_TRY(hipFree(ch->tmp_f));
            if (ch->tmp_key) HIP_TRY(hipFree(ch->tmp_key));
            ch->tmp_x = ch->tmp_f = ch->tmp_key = nullptr;
            HIP_TRY(hipMalloc(&ch->tmp_key, sizeof(double) * cols));
            ch->tmp_cols = cols;
        }
        if (need_xf && !ch->tmp_x) {
            HIP_TRY(hipMalloc(&ch->tmp_x, sizeof(double) * ch->tmp_cols * (size_t)ch->n));
            HIP_TRY(hipMalloc(&ch->tmp_f, sizeof(double) * ch->tmp_cols));
        }
        pk = ch->tmp_key;
        if (!px && out->d_win_x) px = ch->tmp_x;
        if (!pf && out->d_win_f) pf = ch->tmp_f;
    }

    SolveLaunch a;
    std::memset(&a, 0, sizeof a);
    a.chain = ch->dev;  // (null for a wide chain: its launch takes ch->wdev)
    make_eval_params(cfg->linear_weight, cfg->angular_weight, ee_offset7, a.ep);
    fill_solve_params(cfg, a.sp, opt().stop_x_legacy != 0);
    std::memcpy(a.key, ch->key, sizeof a.key);
    std::memcpy(a.scale, ch->scale, sizeof a.scale);  // (n <= 8; a wide chain's scales are in its table)
    a.wq.next_item = ch->queue;
    a.wq.total_items = (unsigned long long)cols;
    a.wq.n_restarts = R;
    a.wq.restart_begin = restart_begin;
    a.wq.targets = d_targets;
    a.wq.x0 = d_x0;
    a.wq.first_success = early ? ch->first_success : nullptr;
    a.wq.find_any = (early && (flags & OPTIK_HIP_IK_FIND_ANY)) ? 1 : 0;
    a.wq.claim = nullptr;
    a.wq.claim_seq = 0;
    a.wq.restart_major = (flags & OPTIK_HIP_IK_RESTART_MAJOR) ? 1 : 0;
    a.wq.n_targets = (unsigned long long)T;
    a.wq.deadline = 0;
    a.wq.quality = (cfg->solution_mode == 1);
    a.wq.out_x = px;
    a.wq.out_f = pf;
    a.wq.out_key = pk;
    a.wq.out_status = out->d_status;
    a.wq.out_evals = out->d_evals;
    a.wq.prof = nullptr;
#ifdef OPTIK_PROFILE
    if (!ch->prof) HIP_TRY(hipMalloc(&ch->prof, 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(ch->prof, 0, 8 * sizeof(unsigned long long), stream));
    a.wq.prof = ch->prof;
#endif
    a.deadline_ticks = 0;
    if (deadline_s > 0.0) {
        const double khz = ch->wall_clock_khz > 0 ? (double)ch->wall_clock_khz : 100000.0;
        a.deadline_ticks = (unsigned long long)(deadline_s * khz * 1e3);
        if (a.deadline_ticks == 0) a.deadline_ticks = 1;
    }

    // Which solver (option solve_kernel; same results, bit for bit): the quad solver of ik_quad.hpp (a restart per
    // quad of lanes, its state spread over the quad, NNLS matrix in LDS; n <= 8), from one full load of the chip
    // on the lane-per-restart form of ik_lane64.hpp (n <= 7), or -- `general` -- the run-time-n solver of
    // ik_wide.hpp on a chain of at most 8 joints too: a third, independently written device solver for the parity
    // tests; chains of 9 .. 16 joints always run on it.
    const int sk = opt().solve_kernel;
    bool widek = ch->wide || sk == SK_GENERAL;
    if (widek && !ch->wide) {
        // the chain's table in the general kernels' layout (uploaded per call: a test path)
        WideChainDev &w = ch->whost;
        std::memset(&w, 0, sizeof w);
        w.n_pos = ch->n;
        w.has_tip = ch->tip;
        for (int j = 0; j < ch->n + (ch->tip ? 1 : 0); ++j)
            for (int k = 0; k < 7; ++k) w.origin[j][k] = ch->host.origin[j][k];
        for (int j = 0; j < ch->n; ++j) {
            for (int k = 0; k < 3; ++k) w.axis[j][k] = ch->host.axis[j][k];
            w.lb[j] = ch->host.lb[j]; w.ub[j] = ch->host.ub[j]; w.scale[j] = ch->scale[j];
        }
        if (!ch->wdev) HIP_TRY(hipMalloc(&ch->wdev, sizeof(WideChainDev)));
        HIP_TRY(hipMemcpy(ch->wdev, &w, sizeof(WideChainDev), hipMemcpyHostToDevice));
    }
    const bool quadk = !widek;
    // the throughput form for n <= 7: one restart per lane, bounded sub-problems in class order (ik_lane64.hpp)
    // (the default from one full load of the chip on -- 64 restarts for each of its four waves per CU: below that a
    // launch is as long as its longest restart, and the quad solver's trip is the shorter one; lane_vs_quad_probe.py (a rounds 3-5 tool: git history))
    bool lanek = quadk && ch->n <= 7 && sk != SK_QUAD;
    const bool lane_forced = lanek && sk == SK_LANE64;
    // Persistent waves, each pulling work items until the queue is dry: as many as a CU holds
    // (lane kernel: 2 workgroups, LDS-bound; cooperative kernel: 4, one per SIMD), times the CU count.
    const int cus = ch->num_cus > 0 ? ch->num_cus : 256;
    // (quad solver: a launch with no more work items than the chip has SIMDs runs one restart per wave on the
    // one-wave-per-SIMD build -- no scratch, the lowest latency per iteration; anything bigger on the
    // two-waves-per-SIMD build)
    const long long wide_waves_per_cu = 8;  // resident waves per CU of the general solver (two per SIMD)
    const bool quad_latency = quadk && (long long)cols <= (long long)cus * 4 && !lane_forced;
    // (not for a Speed batch's latency-sized rounds: restart-major hand-out with early exit keeps a few restarts per
    // target in flight and abandons most of the rest -- the quad solver's shorter trip wins there)
    lanek = lanek && !quad_latency
            && (lane_forced || ((long long)cols >= (long long)cus * lane_solve_waves_per_cu() * 64
                                && !(early && (flags & OPTIK_HIP_IK_RESTART_MAJOR))));
    long long cap = (long long)cus * (lanek ? lane_solve_waves_per_cu() : quadk ? (quad_latency ? 4 : quad_solve_waves_per_cu(ch->n)) : wide_waves_per_cu);
    const long long per_wave_max = (quadk && !lanek) ? QUADS_PER_WAVE_HOST : WAVE;
    // fewer work items than the chip holds: one restart per wave (or as few as fit).  A
    // restart-major Speed batch keeps about eight restarts per target in flight: the waves pull
    // the higher indices of the targets still unsolved as they go
    long long resident = (long long)cols;
    // (but never fewer than one restart per resident wave: a small batch has the chip to itself, and
    // the more of a target's restarts run at once the sooner its first success comes)
    const long long inflight = 8;  // restarts per target in flight
    // (a few targets have the chip to themselves: two restarts per resident wave at least, 32 per
    // target up to 256 targets -- measured: 64 targets 1.01 -> 0.79 ms, 256: 1.66 -> 1.47 ms, and the
    // few hundred targets a big batch's first round leaves over 8 ms sooner)
    if (early && (flags & OPTIK_HIP_IK_RESTART_MAJOR) && resident > (long long)T * inflight) {
        const long long floor_res = std::max(2 * cap, (long long)T * 32);
        resident = std::max((long long)T * inflight, std::min(resident, floor_res));
    }
    long long lanes = (resident + cap - 1) / cap;
    if (lanes < 1) lanes = 1;
    if (lanes > per_wave_max) lanes = per_wave_max;
    // The general solver's two forms (ik_wide.hpp): one restart per wave with its arrays in LDS and the wave's 64
    // lanes working on it together, or a restart per lane with the HBM workspace.  The first has the short
    // dependent chain and no HBM traffic, the second 64 times the restarts in flight -- and the first wins at
    // every size and joint count measured (wide_chain_bench.py (a rounds 3-5 tool: git history), 262 144 restarts: 1.31 / 0.88 / 0.83 / 1.26 M
    // restarts/s at 9 / 10 / 12 / 16 joints against 1.05 / 0.66 / 0.42 / 0.40 M; a launch on the HBM form takes
    // 50 - 100 ms however small it is).  Option wide_form = hbm selects the HBM form (tests, comparisons).
    bool wide_lds = false;
    if (widek) {
        wide_lds = opt().wide_form != 1;
        if (wide_lds) lanes = 1;
    }
    a.wq.lanes = (int)lanes;
    // a single call under the first-success rule on the quad solver: the first success goes to the host at once
    if (claim_request && quadk && !lanek && a.wq.find_any && T == 1 && ch->hw_claim) {
        a.wq.claim = ch->hw_claim;
        a.wq.claim_seq = ++ch->claim_seq;
        if (claim_armed) *claim_armed = true;
        if (stream == nullptr) ch->claim_pending = true;  // (the caller may return before this launch has ended)
    }
    long long grid_ll = (resident + lanes - 1) / lanes;
    if (grid_ll > cap) grid_ll = cap;
    const int grid = (int)grid_ll;

    const int ev_slot = ch->ev_count % optik_hip_chain::EV_POOL;
    if (ch->timing) {
        if (!ch->ev0[ev_slot]) { HIP_TRY(hipEventCreate(&ch->ev0[ev_slot])); HIP_TRY(hipEventCreate(&ch->ev1[ev_slot])); }
        HIP_TRY(hipEventRecord(ch->ev0[ev_slot], stream));
    }
    int lds = 0;
    if (widek) {
        // 9 .. 16 joint positions: one restart per lane on the general kernel, eight waves per CU, every
        // resident wave with its own block of the restart workspace (ik_wide.hpp)
        // (one restart per wave -- a single ik() call's rounds --: the restart's arrays in the wave's LDS)
        const bool lds_form = wide_lds;
        if (!lds_form && (size_t)grid > ch->wide_ws_waves) {
            if (ch->wide_ws) HIP_TRY(hipFree(ch->wide_ws));
            ch->wide_ws = nullptr; ch->wide_ws_waves = 0;
            HIP_TRY(hipMalloc(&ch->wide_ws, sizeof(double) * wide_ws_doubles_per_wave() * (size_t)grid));
            ch->wide_ws_waves = (size_t)grid;
        }
        WideSolveLaunch w;
        std::memset(&w, 0, sizeof w);
        w.chain = ch->wdev;
        w.ep = a.ep; w.sp = a.sp; w.wq = a.wq;
        std::memcpy(w.key, ch->key, sizeof w.key);
        w.deadline_ticks = a.deadline_ticks;
        w.ws = ch->wide_ws;
        lds = lds_form ? wide_lds_bytes() : (int)sizeof(WideChainDev);
        HIP_TRY(wide_solve_launch(grid, stream, w, lds_form, opt().wide_form != 2));
    } else if (lanek) {
        HIP_TRY(lane_solve_launch(ch->n, ch->tip, grid, stream, a, &lds));
    } else if (quadk) {
        HIP_TRY(quad_solve_launch(ch->n, ch->tip, grid, stream, a, &lds, quad_latency));
    }
    else return fail(OPTIK_HIP_EUNSUPPORTED, "no solver for this chain in this build");
    HIP_TRY(hipGetLastError());
    if (ch->timing) { HIP_TRY(hipEventRecord(ch->ev1[ev_slot], stream)); ch->ev_count += 1; }
    ch->last.grid = grid; ch->last.block = WAVE; ch->last.lds_bytes = lds; ch->last.tiles = n_tiles;

    if (want_win) {
        SelectLaunch s;
        std::memset(&s, 0, sizeof s);
        s.out_key = pk; s.out_x = px; s.out_f = pf;
        s.tile_recs = ch->tile_recs;
        s.tiles_per_target = (int)tiles_per_target;
        s.tile = SEL_TILE;
        s.n = ch->n;
        s.restart_begin = restart_begin;
        s.n_restarts = R;
        s.ld = cols;
        s.win_x = out->d_win_x; s.win_f = out->d_win_f;
        s.win_idx = (unsigned long long *)out->d_win_idx; s.win_key = out->d_win_key;
        s.reset_queue = ch->queue;
        s.reset_fs = early ? ch->first_success : nullptr;
        HIP_TRY(select_launch(s, T, stream));
        ch->queue_clean = true;
        ch->fs_clean = fs_clean_after;
        ch->clean_stream = stream;
    }
    return 0;
}

extern "C" {

/* Tuning options (tests, tools): see `struct Options` (ik_host.hpp).  Names: solve_kernel (0 auto, 1 quad, 2 lane64,
 * 3 general), wide_form (0 lds, 1 hbm), range_rule (of chains created afterwards), stop_x_legacy.  Not synchronised
 * with calls in flight. */
static long long *option_slot(const char *name, int **islot) {
    Options &o = opt();
    *islot = nullptr;
    if (!name) return nullptr;
    if (!std::strcmp(name, "solve_kernel")) { *islot = &o.solve_kernel; return nullptr; }
    if (!std::strcmp(name, "wide_form")) { *islot = &o.wide_form; return nullptr; }
    if (!std::strcmp(name, "range_rule")) { *islot = &o.range_rule; return nullptr; }
    if (!std::strcmp(name, "stop_x_legacy")) { *islot = &o.stop_x_legacy; return nullptr; }
    return nullptr;
}
int optik_hip_set_option(const char *name, long long value) {
    int *is = nullptr;
    long long *ls = option_slot(name, &is);
    if (ls) { *ls = value; return 0; }
    if (is) { *is = (int)value; return 0; }
    return fail(OPTIK_HIP_EINVAL, "unknown option");
}
long long optik_hip_get_option(const char *name) {
    int *is = nullptr;
    long long *ls = option_slot(name, &is);
    return ls ? *ls : (is ? (long long)*is : -1);
}

int optik_hip_ik_host(optik_hip_chain *ch, const optik_solver_config *cfg, const double *targets,
                      const double *x0, int32_t T, const double *ee_offset7, uint64_t restart_begin,
                      uint64_t restart_end, uint32_t flags, double deadline_s, double *win_x, double *win_f,
                      uint64_t *win_idx, double *win_key) {
    if (!ch || !cfg || !targets || !x0 || T < 1) return fail(OPTIK_HIP_EINVAL, "bad argument");  // (cfg is read below, before ik_batch_locked's own check)
    // the launch workspace of the chain is in use until the copies below are done
    std::lock_guard<std::mutex> host_lock(ch->host_mu);
    BIND_DEVICE(ch);
    const int n = ch->n;
    // one device block and one pinned staging block, kept with the chain (a call used to pay
    // six hipMalloc / hipFree pairs and six copies): in = targets [T][7], x0 [T][n];
    // out = win_x [T][n], win_f [T], win_key [T], win_idx [T]
    const size_t n_in = (size_t)(7 + n) * (size_t)T, n_out = (size_t)(n + 3) * (size_t)T;
    // (`mu` from here to the launch: the claim state and the staging blocks belong to the launch workspace)
    std::unique_lock<std::mutex> launch_lock(ch->mu);
    if (n_in + n_out > ch->hw_cap) {
        if (ch->claim_pending) { (void)hipStreamSynchronize(nullptr); ch->claim_pending = false; }  // (its launch reads the block about to go)
        if (ch->hw_dev) (void)hipFree(ch->hw_dev);
        if (ch->hw_pin) (void)hipHostFree(ch->hw_pin);
        ch->hw_dev = nullptr; ch->hw_pin = nullptr; ch->hw_cap = 0;
        HIP_TRY(hipMalloc(&ch->hw_dev, sizeof(double) * (n_in + n_out)));
        HIP_TRY(hipHostMalloc(&ch->hw_pin, sizeof(double) * 2 * (n_in + n_out)));  // (two blocks, see below)
        ch->hw_cap = n_in + n_out;
    }
    // A few targets (Robot::ik: one): the kernels read the inputs from and write the winners to the
    // pinned block directly -- no copy commands around the launch.  (Two such blocks, used in turn: a call
    // that returned on the first success -- below -- leaves a launch behind whose last restarts still read theirs.)
    const bool zero_copy = T <= 16;
    double *pin = ch->hw_pin;
    if (zero_copy) {
        pin += (ch->hw_flip & 1u) * ch->hw_cap;
        ch->hw_flip ^= 1u;
    } else if (ch->claim_pending) {
        // The staged path always uses block 0.  A launch that a first-success call left running may have been given
        // that block: its selection kernel still writes its winner there -- inside the region the targets are about
        // to be staged in -- so it has to end first (the two-block flip only protects the zero-copy calls).
        HIP_TRY(hipStreamSynchronize(nullptr));
        ch->claim_pending = false;
    }
    double *io = zero_copy ? pin : ch->hw_dev;
    double *d_t = io, *d_x0 = d_t + (size_t)7 * T;
    double *d_wx = io + n_in, *d_wf = d_wx + (size_t)n * T, *d_wk = d_wf + T;
    uint64_t *d_wi = reinterpret_cast<uint64_t *>(d_wk + T);
    std::memcpy(pin, targets, sizeof(double) * 7 * (size_t)T);
    std::memcpy(pin + (size_t)7 * T, x0, sizeof(double) * (size_t)n * (size_t)T);
    if (!zero_copy)
        HIP_TRY(hipMemcpyAsync(ch->hw_dev, pin, sizeof(double) * n_in, hipMemcpyHostToDevice, nullptr));
    optik_hip_ik_outputs o;
    std::memset(&o, 0, sizeof o);
    o.d_win_x = d_wx; o.d_win_f = d_wf; o.d_win_idx = d_wi; o.d_win_key = d_wk;
    // One target under the first-success rule (lib.rs:409-412, the reference's default): the first restart to
    // succeed writes its answer to a host-coherent block and the call returns as soon as it is there; the launch's
    // other restarts notice the flag at their next evaluation and the launch ends behind the caller's back (the
    // next launch of the chain queues behind it).  Without a success the call ends with the launch, as before.
    const bool claim = T == 1 && (flags & OPTIK_HIP_IK_FIND_ANY) && (flags & OPTIK_HIP_IK_EARLY_EXIT)
                       && cfg->solution_mode == 2;
    if (claim && !ch->hw_claim) {
        HIP_TRY(hipHostMalloc(&ch->hw_claim, sizeof(unsigned long long) * (3 + MAX_DOF), hipHostMallocCoherent));
        std::memset(ch->hw_claim, 0, sizeof(unsigned long long) * (3 + MAX_DOF));
        HIP_TRY(hipEventCreateWithFlags(&ch->claim_done, hipEventDisableTiming));
    }
    bool armed = false;
    unsigned long long seq = 0;
    int rc = ik_batch_locked(ch, cfg, d_t, d_x0, T, ee_offset7, restart_begin, restart_end, flags, deadline_s, &o, nullptr,
                             claim, &armed);
    seq = ch->claim_seq;
    // (the end of THIS launch, not of the null stream: other chains' calls may keep that one busy)
    if (!rc && armed && hipEventRecord(ch->claim_done, nullptr) != hipSuccess) rc = fail(OPTIK_HIP_ENODEVICE, "hipEventRecord failed");
    launch_lock.unlock();
    if (rc) return rc;
    double *h_out = pin + n_in;
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(h_out, d_wx, sizeof(double) * n_out, hipMemcpyDeviceToHost, nullptr));
    if (armed) {
        volatile unsigned long long *cw = ch->hw_claim;
        for (unsigned spin = 1;; ++spin) {
            if (__atomic_load_n(ch->hw_claim, __ATOMIC_ACQUIRE) == seq) {
                if (win_x) std::memcpy(win_x, (const void *)(cw + 3), sizeof(double) * (size_t)n);
                if (win_f) std::memcpy(win_f, (const void *)(cw + 2), sizeof(double));
                if (win_idx) *win_idx = cw[1];
                if (win_key) *win_key = (double)cw[1];
                return 0;  // (claim_pending stays set: the launch ends behind the caller's back)
            }
            if ((spin & 63u) == 0) {
                const hipError_t q = hipEventQuery(ch->claim_done);
                if (q == hipSuccess) break;  // the launch is over and nobody succeeded (or the word is about to land)
                if (q != hipErrorNotReady) HIP_TRY(q);
            }
        }
        if (__atomic_load_n(ch->hw_claim, __ATOMIC_ACQUIRE) == seq) {
            if (win_x) std::memcpy(win_x, (const void *)(cw + 3), sizeof(double) * (size_t)n);
            if (win_f) std::memcpy(win_f, (const void *)(cw + 2), sizeof(double));
            if (win_idx) *win_idx = cw[1];
            if (win_key) *win_key = (double)cw[1];
            return 0;
        }
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    if (armed) {
        // (nothing of this call is left on the null stream -- unless a later launch of the chain armed a claim meanwhile)
        std::lock_guard<std::mutex> relock(ch->mu);
        if (ch->claim_seq == seq) ch->claim_pending = false;
    }
    if (win_x) std::memcpy(win_x, h_out, sizeof(double) * (size_t)n * (size_t)T);
    if (win_f) std::memcpy(win_f, h_out + (size_t)n * T, sizeof(double) * (size_t)T);
    if (win_key) std::memcpy(win_key, h_out + (size_t)(n + 1) * T, sizeof(double) * (size_t)T);
    if (win_idx) std::memcpy(win_idx, h_out + (size_t)(n + 2) * T, sizeof(uint64_t) * (size_t)T);
    return 0;
}

void optik_hip_set_timing(optik_hip_chain *ch, int32_t enabled) {
    if (!ch) return;
    std::lock_guard<std::mutex> lock(ch->mu);
    ch->timing = enabled;
    ch->ev_count = 0;
}

/* OPTIK_PROFILE builds: phase cycle totals of the last solve launch (8 words:
 * refill, eval, update, publish, bfgs, lsq, nnls, trips); zeros otherwise. */
int optik_hip_phase_profile(optik_hip_chain *ch, unsigned long long *out8) {
    if (!ch || !out8) return fail(OPTIK_HIP_EINVAL, "bad argument");
    std::memset(out8, 0, 8 * sizeof(unsigned long long));
    if (!ch->prof) return 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out8, ch->prof, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

int optik_hip_timing_mean(optik_hip_chain *ch, double *mean_ms, int32_t *count) {
    if (!ch || !mean_ms || !count) return fail(OPTIK_HIP_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lock(ch->mu);
    const int n = ch->ev_count < optik_hip_chain::EV_POOL ? ch->ev_count : optik_hip_chain::EV_POOL;
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        HIP_TRY(hipEventSynchronize(ch->ev1[i]));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ch->ev0[i], ch->ev1[i]));
        sum += ms;
    }
    *mean_ms = n ? sum / n : 0.0;
    *count = n;
    return 0;
}

int optik_hip_last_launch(const optik_hip_chain *ch, optik_hip_launch_info *info) {
    if (!ch || !info) return fail(OPTIK_HIP_EINVAL, "bad argument");
    *info = ch->last;
    info->kernel_ms = 0.0f;
    if (ch->timing && ch->ev_count > 0) {
        const int slot = (ch->ev_count - 1) % optik_hip_chain::EV_POOL;
        HIP_TRY(hipEventSynchronize(ch->ev1[slot]));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ch->ev0[slot], ch->ev1[slot]));
        info->kernel_ms = ms;
    }
    return 0;
}

}  // extern "C"
