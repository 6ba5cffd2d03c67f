// quad_emu.cpp -- TEST INFRASTRUCTURE: runs the quad-distributed restart solver of
// optik_amd/csrc/ik_quad.hpp on the HOST (one thread per emulated lane, tests/emu/lane_emu.hpp) so
// that tests/test_quad_emulation.py can compare every restart with the C oracle bit for bit without
// a GPU.  The device headers are compiled as they are (only the cross-lane primitives and the HIP
// intrinsics are swapped for their emulation), with -ffp-contract=off like the kernels.
//
// Not part of the product: nothing under optik_amd/ or bench.py builds, loads or calls this.
#define OPTIK_LANE_EMU 1
#include <thread>
#include <vector>

#include "ik_quad.hpp"
#include "ik_lane64.hpp"
#include "ik_host_params.hpp"

using namespace optik;

namespace {

template <int N, bool TIP>
void run_wave(const ChainDev &ch, const EvalParams &ep, const SolveParams &sp, const uint32_t (&key)[8],
              const double (&scale)[MAX_DOF], const WorkQueue &wq, int quads) {
    optik_emu::Wave wave;
    wave.lanes = 4 * quads;
    std::vector<double> lds((size_t)quad_wave_lds<N>(), 0.0), lane_lds((size_t)quad_lane_lds(), 0.0);
    std::vector<std::thread> th;
    for (int lane = 0; lane < wave.lanes; ++lane) {
        th.emplace_back([&, lane]() {
            optik_emu::t_wave = &wave;
            threadIdx.x = (unsigned)lane;
            quad_wave<N, TIP>(ch, ep, sp, key, scale, wq, lds.data(), lane_lds.data());
        });
    }
    for (auto &t : th) t.join();
}

// the lane-per-restart form (ik_lane64.hpp): `lanes` emulated lanes (a multiple of 4), each with its own restart
template <int N, bool TIP>
void run_wave_lane64(const ChainDev &ch, const EvalParams &ep, const SolveParams &sp, const uint32_t (&key)[8],
                     const double (&scale)[MAX_DOF], const WorkQueue &wq, int lanes) {
    optik_emu::Wave wave;
    wave.lanes = lanes;
    std::vector<double> lds((size_t)lane64_block_lds<N>(), 0.0), rec((size_t)lane64_rec_lds<N>(), 0.0);
    std::vector<int> lor(64, 0), where(64, 0);
    std::vector<std::thread> th;
    for (int lane = 0; lane < wave.lanes; ++lane) {
        th.emplace_back([&, lane]() {
            optik_emu::t_wave = &wave;
            threadIdx.x = (unsigned)lane;
            lane64_wave<N, TIP>(ch, ep, sp, key, scale, wq, lds.data(), rec.data(), lor.data(), where.data());
        });
    }
    for (auto &t : th) t.join();
}

}  // namespace

extern "C" {

// The cross-lane moves of ik_lane.hpp on an 8-lane partial wave, every lane holding its own number: out[m * 8 + lane]
// for m = quad_get(., 0..3), quad_xor(., 1), quad_xor(., 2), quad_rot(., 1..3) -- through the same DPP controls the
// device executes, decoded as the hardware decodes them (lane_emu.hpp).
int quad_emu_lane_moves(int *out) {
    optik_emu::Wave wave;
    wave.lanes = 8;
    std::vector<std::thread> th;
    for (int lane = 0; lane < wave.lanes; ++lane) {
        th.emplace_back([&, lane]() {
            optik_emu::t_wave = &wave;
            threadIdx.x = (unsigned)lane;
            const int me = wave_lane_now();
            for (int k = 0; k < 4; ++k) out[k * 8 + lane] = quad_get(me, k);
            out[4 * 8 + lane] = quad_xor(me, 1);
            out[5 * 8 + lane] = quad_xor(me, 2);
            for (int r = 1; r <= 3; ++r) out[(5 + r) * 8 + lane] = (int)quad_rot((uint32_t)me, r);
            out[9 * 8 + lane] = (int)quad_get(1.5 * me, 2) == (int)(1.5 * ((me & ~3) | 2)) ? 1 : 0;  // (a 64-bit value: both halves)
        });
    }
    for (auto &t : th) t.join();
    return 0;
}

// origins [J][7] (t, quat ijkw), axes [n][3], J = n or n + 1; restarts [begin, end) of ONE target.
// out_x [n][R], out_f / out_key [R], out_status / out_evals [R].  quads: restarts in flight (1 .. 16).
int quad_emu_solve(const double *origins, const double *axes, int n, int n_joints, const double *lb, const double *ub,
                   const optik_solver_config *cfg, const double *target7, const double *x0, const double *ee_offset7,
                   uint64_t restart_begin, uint64_t restart_end, int quads, int range_rule, double *out_x, double *out_f,
                   double *out_key, int32_t *out_status, int32_t *out_evals, int lane64 /* 1: ik_lane64.hpp, 4 * quads lanes */) {
    if (n < 1 || n > 8 || quads < 1 || quads > 16 || restart_end <= restart_begin) return -1;
    if (lane64 && n > 7) return -1;
    ChainDev ch;
    std::memset(&ch, 0, sizeof ch);
    ch.n_pos = n;
    ch.has_tip = n_joints == n + 1;
    for (int j = 0; j < n_joints; ++j)
        for (int c = 0; c < 7; ++c) ch.origin[j][c] = origins[j * 7 + c];
    for (int j = 0; j < n; ++j) {
        for (int c = 0; c < 3; ++c) ch.axis[j][c] = axes[j * 3 + c];
        ch.lb[j] = lb[j];
        ch.ub[j] = ub[j];
    }
    EvalParams ep;
    hostparams::make_eval_params(cfg->linear_weight, cfg->angular_weight, ee_offset7, ep);
    SolveParams sp;
    hostparams::fill_solve_params(cfg, sp);
    uint32_t key[8];
    hostparams::seed_from_u64(42, key);
    double scale[MAX_DOF] = {0};
    for (int j = 0; j < n; ++j) scale[j] = hostparams::uniform_scale(lb[j], ub[j], range_rule);

    unsigned long long counter = 0;
    WorkQueue wq;
    std::memset(&wq, 0, sizeof wq);
    const uint64_t R = restart_end - restart_begin;
    wq.next_item = &counter;
    wq.total_items = R;
    wq.n_restarts = R;
    wq.restart_begin = restart_begin;
    wq.targets = target7;
    wq.x0 = x0;
    wq.first_success = nullptr;
    wq.n_targets = 1;
    wq.quality = cfg->solution_mode == 1;
    wq.lanes = lane64 ? 4 * quads : quads;
    wq.out_x = out_x;
    wq.out_f = out_f;
    wq.out_key = out_key;
    wq.out_status = out_status;
    wq.out_evals = out_evals;

#define RUN(NN)                                                                              \
    case NN:                                                                                 \
        if (ch.has_tip) run_wave<NN, true>(ch, ep, sp, key, scale, wq, quads);               \
        else run_wave<NN, false>(ch, ep, sp, key, scale, wq, quads);                         \
        break;
#define RUN64(NN)                                                                            \
    case NN:                                                                                 \
        if (ch.has_tip) run_wave_lane64<NN, true>(ch, ep, sp, key, scale, wq, 4 * quads);    \
        else run_wave_lane64<NN, false>(ch, ep, sp, key, scale, wq, 4 * quads);              \
        break;
    if (lane64) {
        switch (n) {
            RUN64(1) RUN64(2) RUN64(3) RUN64(4) RUN64(5) RUN64(6) RUN64(7)
        default: return -1;
        }
        return 0;
    }
    switch (n) {
        RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    default: return -1;
    }
#undef RUN
#undef RUN64
    return 0;
}

}  // extern "C"
