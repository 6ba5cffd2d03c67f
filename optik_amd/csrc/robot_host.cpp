// robot_host.cpp -- the reference's host API on top of the HIP kernel layer.
//
// Mirrors `Robot` (/root/reference/crates/optik/src/lib.rs:36-99, 241-415) and
// exports the C ABI of crates/optik-cpp/src/lib.rs:26-183 (include/optik.h).
// What stays on the host is what the reference does once per call or per robot:
// URDF loading, the seed-limit check, the restart/time budget and the loop over
// launches.  FK, Jacobian, the restarts and the selection all run in the kernels
// of ik_capi.hip / ik_batch_ops.hip -- there is no CPU implementation of them in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/optik.h"
#include "device_scope.hpp"
#include "urdf_chain.hpp"

using optik_host::Chain;
using optik_host::HPose;

// What the robot keeps on one GPU: the uploaded chain and reusable staging for the host API.
struct DeviceCtx {
    int device = 0;
    optik_hip_chain *chain = nullptr;
    double *d_scratch = nullptr;  // q[n] | pose[7] | jac[6n]
    double *h_scratch = nullptr;  // the same, pinned host memory the FK kernel reads and writes directly
    int num_cus = 0;
    // optik_robot_ik_batch_ex workspace, grown on demand and kept across calls:
    // device block = targets [T][7] | x0 [T][n] | win_x [T][n] | win_f [T] | win_key [T] | win_idx [T]
    double *d_batch = nullptr;
    double *h_batch = nullptr;  // pinned mirror
    size_t batch_cap = 0;       // doubles
    std::mutex batch_mu;        // one batch at a time per device
};

struct optik_robot {
    Chain chain;
    int n = 0;
    std::vector<double> lb, ub;
    std::vector<double> origins, axes;  // n_joints x 7, n_joints x 3
    std::vector<int32_t> types;
    // set_parallelism (lib.rs:66-72).  The rayon pool size has no counterpart, but its one
    // observable consequence has: with one thread SolutionMode::Speed returns the lowest
    // successful restart (deterministic; tests/test_ik.rs:45-89 sets 1 for exactly that), with more
    // it returns whichever success comes first (find_any, lib.rs:409-412; README.md:17, 96).
    // 0 = never set = the reference's default pool (ThreadPoolBuilder::default(): every core,
    // lib.rs:42-47) and n > 1 let a Speed call stop at the first success anywhere; 1 gives the
    // deterministic 1-thread answer.
    unsigned parallelism = 0;
    mutable std::mutex mu;     // guards the lazily created device contexts and the FK scratch
    // GPUs this robot spreads restart ranges / targets over (optik_robot_set_devices,
    // OPTIK_DEVICES); empty = the HIP device current at first use.  The same id may be listed
    // more than once (two contexts on one GPU: how the sharding is tested on a 1-GPU box).
    std::vector<int> device_ids;
    mutable std::vector<std::unique_ptr<DeviceCtx>> devs;
    // over how many of them the widest round of the last ik / ik_batch call was actually cut (optik_robot_last_parts)
    mutable std::atomic<int32_t> last_parts{0};
};

namespace {

thread_local std::string g_robot_err;

[[noreturn]] void panic(const std::string &msg) {
    // A Rust panic crossing `extern "C"` aborts the process; keep the message.
    std::fprintf(stderr, "optik: %s\n", msg.c_str());
    std::fflush(stderr);
    std::abort();
}

int set_err(int code, const std::string &msg) {
    g_robot_err = msg;
    return code;
}

std::vector<int> devices_from_env();

// fn(begin, end) over [0, count) on a few host threads when the range is long (the per-target
// host work of a batch -- validation, pose conversion, staging, gathering -- is ~50 ns a target:
// 13 of 60 ms at 262 144 targets on one thread)
template <class Fn>
void parallel_ranges(size_t count, Fn fn) {
    static const unsigned max_threads = [] {
        const char *e = std::getenv("OPTIK_HOST_THREADS");
        unsigned h = e ? (unsigned)std::atoi(e) : std::thread::hardware_concurrency() / 2;
        return h < 1 ? 1u : (h > 8 ? 8u : h);
    }();
    const size_t parts = count < 32768 ? 1 : std::min<size_t>(max_threads, count / 16384);
    if (parts <= 1) { fn((size_t)0, count); return; }
    std::vector<std::thread> th;
    for (size_t p = 1; p < parts; ++p) th.emplace_back(fn, count * p / parts, count * (p + 1) / parts);
    fn((size_t)0, count / parts);
    for (auto &t : th) t.join();
}

optik_robot *make_robot(const std::string &urdf, const char *base, const char *ee) {
    auto *r = new optik_robot();
    try {
        r->chain = optik_host::chain_from_urdf(urdf, base, ee);
    } catch (...) {
        delete r;
        throw;
    }
    r->n = r->chain.num_positions();
    if (r->n > OPTIK_HIP_MAX_DOF) {
        const int n = r->n;
        delete r;
        throw std::runtime_error("chain has " + std::to_string(n) + " joint positions; the gfx950 kernels are built for "
                                 "at most " + std::to_string(OPTIK_HIP_MAX_DOF) + " (a limit of this implementation, not of the reference)");
    }
    for (const auto &j : r->chain.joints) {
        for (int k = 0; k < 3; ++k) r->origins.push_back(j.origin.t[k]);
        for (int k = 0; k < 4; ++k) r->origins.push_back(j.origin.q[k]);
        for (int k = 0; k < 3; ++k) r->axes.push_back(j.axis[k]);
        r->types.push_back(j.kind);
        if (j.kind != optik_host::FIXED) {  // joint_limits(), lib.rs:78-84
            r->lb.push_back(j.lower);
            r->ub.push_back(j.upper);
        }
    }
    r->device_ids = devices_from_env();
    return r;
}

// Context k of the robot (its k-th listed GPU), created on first use.  Returns nullptr and sets
// the error string on failure.
DeviceCtx *device_ctx(const optik_robot *r, size_t k = 0) {
    std::lock_guard<std::mutex> lock(r->mu);
    if (r->devs.empty()) {
        const size_t count = r->device_ids.empty() ? 1 : r->device_ids.size();
        for (size_t i = 0; i < count; ++i) r->devs.emplace_back(new DeviceCtx());
    }
    if (k >= r->devs.size()) { g_robot_err = "no such device context"; return nullptr; }
    DeviceCtx *c = r->devs[k].get();
    if (c->chain) return c;
    int devid = 0;
    if (r->device_ids.empty()) (void)hipGetDevice(&devid);
    else devid = r->device_ids[k];
    // (the caller's device is current again when this returns, also when the set-up fails)
    optik::DeviceScope dev_scope(devid);
    if (!dev_scope.ok()) {
        g_robot_err = "hipSetDevice(" + std::to_string(devid) + ") failed";
        return nullptr;
    }
    optik_hip_chain *h = nullptr;
    const int rc = optik_hip_chain_create(r->origins.data(), r->axes.data(), r->types.data(),
                                          (int32_t)r->types.size(), r->lb.data(), r->ub.data(), r->n, &h);
    if (rc) {
        g_robot_err = std::string("GPU chain creation failed: ") + optik_hip_last_error();
        return nullptr;
    }
    if (hipMalloc(&c->d_scratch, sizeof(double) * (size_t)(r->n + 7 + 6 * r->n)) != hipSuccess
        || hipHostMalloc(&c->h_scratch, sizeof(double) * (size_t)(r->n + 7 + 6 * r->n)) != hipSuccess) {
        if (c->d_scratch) { (void)hipFree(c->d_scratch); c->d_scratch = nullptr; }
        optik_hip_chain_destroy(h);
        g_robot_err = "GPU scratch allocation failed";
        return nullptr;
    }
    (void)hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, devid);
    c->device = devid;
    c->chain = h;
    return c;
}

size_t device_count(const optik_robot *r) { return r->device_ids.empty() ? 1 : r->device_ids.size(); }

// OPTIK_DEVICES = "all" | "N" (the first N devices) | "0,2,3": GPUs a robot spreads work over
std::vector<int> devices_from_env() {
    std::vector<int> ids;
    const char *e = std::getenv("OPTIK_DEVICES");
    if (!e || !*e) return ids;
    int have = 0;
    (void)hipGetDeviceCount(&have);
    const std::string v(e);
    if (v == "all") { for (int i = 0; i < have; ++i) ids.push_back(i); return ids; }
    if (v.find(',') == std::string::npos) {
        const int cnt = std::atoi(v.c_str());
        for (int i = 0; i < cnt && i < have; ++i) ids.push_back(i);
        return ids;
    }
    std::stringstream ss(v);
    for (std::string tok; std::getline(ss, tok, ',');)
        if (!tok.empty()) ids.push_back(std::atoi(tok.c_str()));
    return ids;
}

// UnitQuaternion::from_rotation_matrix (nalgebra 0.34, not vendored): the four closed-form branches.  It ends in
// Self::new_unchecked(res): NO normalisation -- both bindings' conversions go through it, so one function serves both
// (round 5 normalised on the Python path only: ADVICE r5).
void quat_from_rotation(const double R[3][3], double &w, double &i, double &j, double &k) {
    const double tr = R[0][0] + R[1][1] + R[2][2];
    if (tr > 0.0) {
        const double d = std::sqrt(tr + 1.0) * 2.0;
        w = 0.25 * d; i = (R[2][1] - R[1][2]) / d; j = (R[0][2] - R[2][0]) / d; k = (R[1][0] - R[0][1]) / d;
    } else if (R[0][0] > R[1][1] && R[0][0] > R[2][2]) {
        const double d = std::sqrt(1.0 + R[0][0] - R[1][1] - R[2][2]) * 2.0;
        w = (R[2][1] - R[1][2]) / d; i = 0.25 * d; j = (R[0][1] + R[1][0]) / d; k = (R[0][2] + R[2][0]) / d;
    } else if (R[1][1] > R[2][2]) {
        const double d = std::sqrt(1.0 + R[1][1] - R[0][0] - R[2][2]) * 2.0;
        w = (R[0][2] - R[2][0]) / d; i = (R[0][1] + R[1][0]) / d; j = 0.25 * d; k = (R[1][2] + R[2][1]) / d;
    } else {
        const double d = std::sqrt(1.0 + R[2][2] - R[0][0] - R[1][1]) * 2.0;
        w = (R[1][0] - R[0][1]) / d; i = (R[0][2] + R[2][0]) / d; j = (R[1][2] + R[2][1]) / d; k = 0.25 * d;
    }
}

// 4x4 column-major homogeneous matrix -> pose7 as the pyo3 path does it (optik-py/src/lib.rs:8-15:
// try_convert::<Matrix4, Isometry3> -> Isometry3::from_superset_unchecked -> UnitQuaternion::from_rotation_matrix on the
// upper-left block, as it stands).  The C path's iterative UnitQuaternion::from_matrix is pose7_from_mat16_iterative
// below: the same rotation, last bits apart.
void pose7_from_mat16(const double *m, double *p) {
    double R[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r][c] = m[c * 4 + r];
    p[0] = m[12]; p[1] = m[13]; p[2] = m[14];
    quat_from_rotation(R, p[6], p[3], p[4], p[5]);
}

// Rotation3::from_axis_angle (nalgebra: Rodrigues' formula entry by entry; angle == 0 -> identity).
void rot_from_axis_angle(const double u[3], double angle, double O[3][3]) {
    if (angle == 0.0) {
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) O[r][c] = r == c ? 1.0 : 0.0;
        return;
    }
    const double ux = u[0], uy = u[1], uz = u[2];
    const double sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
    double sn, cs;  // (f64::sin_cos: platform libm; ONE sincos call here and in the oracle -- sin + cos can differ from it in the last bit)
    ::sincos(angle, &sn, &cs);
    const double omc = 1.0 - cs;
    O[0][0] = sqx + (1.0 - sqx) * cs; O[0][1] = ux * uy * omc - uz * sn; O[0][2] = ux * uz * omc + uy * sn;
    O[1][0] = ux * uy * omc + uz * sn; O[1][1] = sqy + (1.0 - sqy) * cs; O[1][2] = uy * uz * omc - ux * sn;
    O[2][0] = ux * uz * omc - uy * sn; O[2][1] = uy * uz * omc + ux * sn; O[2][2] = sqz + (1.0 - sqz) * cs;
}

// 3x3 product as nalgebra forms it (gemv per column: the terms of an entry accumulated in order k = 0, 1, 2).
void mat3_mul(const double A[3][3], const double B[3][3], double O[3][3]) {
    double T[3][3];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            double acc = A[r][0] * B[0][c];
            acc = A[r][1] * B[1][c] + acc;
            acc = A[r][2] * B[2][c] + acc;
            T[r][c] = acc;
        }
    std::memcpy(O, T, sizeof T);
}

// ||M - R||_F^2, column by column (Matrix::norm_squared).
double diff_norm_squared(const double M[3][3], const double R[3][3]) {
    double res = 0.0;
    for (int c = 0; c < 3; ++c) {
        const double d0 = M[0][c] - R[0][c], d1 = M[1][c] - R[1][c], d2 = M[2][c] - R[2][c];
        res += d0 * d0 + d1 * d1 + d2 * d2;
    }
    return res;
}

// UnitQuaternion::from_matrix (optik-cpp/src/lib.rs:141-142) = Rotation3::from_matrix_eps(m, f64::EPSILON, 0 =
// unlimited, identity) -- nalgebra 0.34 (Cargo.lock:579-581, not vendored), "A Robust Method to Extract the
// Rotational Part of Deformations" (Mueller et al.) with nalgebra's perturbation step at a stalled iterate --
// followed by from_rotation_matrix.  Restated from the crate's published source; the iteration cap is this
// implementation's (nalgebra's is usize::MAX: a rotation matrix converges in a few dozen steps).
void quat_from_matrix_iterative(const double M[3][3], double &w, double &i, double &j, double &k) {
    const double eps = 2.220446049250313e-16;
    const double eps_disturbance = std::fmax(std::sqrt(eps), eps * eps);
    double axes[3] = {1.0, 0.0, 0.0};  // perturbation_axes = Vector3::x_axis()
    double rot[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int it = 0; it < 100000; ++it) {
        // axis = sum_c rot.col(c) x m.col(c), denom = sum_c rot.col(c) . m.col(c)
        double axis[3] = {0, 0, 0}, denom = 0.0;
        for (int c = 0; c < 3; ++c) {
            const double a0 = rot[0][c], a1 = rot[1][c], a2 = rot[2][c], b0 = M[0][c], b1 = M[1][c], b2 = M[2][c];
            const double cr[3] = {a1 * b2 - a2 * b1, a2 * b0 - a0 * b2, a0 * b1 - a1 * b0};
            const double dt = a0 * b0 + a1 * b1 + a2 * b2;
            if (c == 0) { axis[0] = cr[0]; axis[1] = cr[1]; axis[2] = cr[2]; denom = dt; }
            else { axis[0] += cr[0]; axis[1] += cr[1]; axis[2] += cr[2]; denom += dt; }
        }
        const double dv = std::fabs(denom) + eps;
        double aa[3] = {axis[0] / dv, axis[1] / dv, axis[2] / dv};
        // Unit::try_new_and_get(axisangle, eps)
        const double sq = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
        if (sq > eps * eps) {
            const double nrm = std::sqrt(sq);
            const double u[3] = {aa[0] / nrm, aa[1] / nrm, aa[2] / nrm};
            double Rd[3][3];
            rot_from_axis_angle(u, nrm, Rd);
            mat3_mul(Rd, rot, rot);
        } else {
            // stuck at a stationary point of ||m - rot||: a maximum, unless a small rotation makes it worse
            double pert[3][3];
            std::memcpy(pert, rot, sizeof pert);
            const double n0 = diff_norm_squared(M, rot);
            double n1;
            for (;;) {
                double Rp[3][3];
                rot_from_axis_angle(axes, eps_disturbance, Rp);
                mat3_mul(pert, Rp, pert);
                n1 = diff_norm_squared(M, pert);
                if (!(std::fabs(n0 - n1) <= eps)) break;  // abs_diff_ne!(.., epsilon = f64::EPSILON): true for NaN too
            }
            if (n0 < n1) break;  // a minimum: done
            const double t = axes[0];  // perturbation_axes.yzx()
            axes[0] = axes[1]; axes[1] = axes[2]; axes[2] = t;
            std::memcpy(rot, pert, sizeof pert);
        }
    }
    quat_from_rotation(rot, w, i, j, k);
}

// 4x4 column-major -> pose7 as the C path of the reference does it (optik-cpp/src/lib.rs:137-144).
void pose7_from_mat16_iterative(const double *m, double *p) {
    double M[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[r][c] = m[c * 4 + r];
    p[0] = m[12]; p[1] = m[13]; p[2] = m[14];
    quat_from_matrix_iterative(M, p[6], p[3], p[4], p[5]);
}

// pose7 -> 4x4 column-major (Isometry3::to_matrix, optik-cpp/src/lib.rs:115).
void mat16_from_pose7(const double *p, double *m) {
    const double i = p[3], j = p[4], k = p[5], w = p[6];
    const double ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    const double ij = i * j * 2.0, wk = w * k * 2.0, wj = w * j * 2.0, ik = i * k * 2.0, jk = j * k * 2.0,
                 wi = w * i * 2.0;
    const double R[3][3] = {{ww + ii - jj - kk, ij - wk, wj + ik},
                            {wk + ij, ww - ii + jj - kk, jk - wi},
                            {ik - wj, wi + jk, ww - ii - jj + kk}};
    for (int c = 0; c < 3; ++c) {
        for (int r = 0; r < 3; ++r) m[c * 4 + r] = R[r][c];
        m[c * 4 + 3] = 0.0;
    }
    m[12] = p[0]; m[13] = p[1]; m[14] = p[2]; m[15] = 1.0;
}

int fk_on_device(const optik_robot *r, const double *x, const double *ee16, double *pose7, double *jac) {
    DeviceCtx *c = device_ctx(r);
    if (!c) return -1;
    optik_hip_chain *h = c->chain;
    double ee7[7];
    if (ee16) pose7_from_mat16(ee16, ee7);
    std::lock_guard<std::mutex> lock(r->mu);
    optik::DeviceScope dev_scope(c->device);
    if (!dev_scope.ok()) return set_err(-1, "hipSetDevice failed");
    // one configuration: the kernel reads q from and writes the pose / Jacobian to pinned host
    // memory (one launch and one wait instead of three copies around them: 43 -> ~20 us per call)
    double *p_q = c->h_scratch, *p_pose = p_q + r->n, *p_jac = p_pose + 7;
    std::memcpy(p_q, x, sizeof(double) * (size_t)r->n);
    if (optik_hip_fk_batch(h, ee16 ? ee7 : nullptr, p_q, 1, p_pose, jac ? p_jac : nullptr, nullptr))
        return set_err(-1, optik_hip_last_error());
    if (hipStreamSynchronize(nullptr) != hipSuccess) return set_err(-1, "kernel failed");
    if (pose7) std::memcpy(pose7, p_pose, sizeof(double) * 7);
    if (jac) std::memcpy(jac, p_jac, sizeof(double) * 6 * (size_t)r->n);
    return 0;
}

double *malloc_copy(const double *src, size_t count) {
    double *p = (double *)std::malloc(sizeof(double) * (count ? count : 1));
    if (!p) panic("out of memory");
    std::memcpy(p, src, sizeof(double) * count);
    return p;
}

}  // namespace

extern "C" {

const char *optik_robot_last_error(void) { return g_robot_err.c_str(); }

int optik_robot_try_from_urdf_str(const char *urdf, const char *base_link, const char *ee_link,
                                  optik_robot **out) {
    if (!urdf || !base_link || !ee_link || !out) return set_err(-1, "null argument");
    try {
        *out = make_robot(urdf, base_link, ee_link);
        return 0;
    } catch (const std::exception &e) {
        return set_err(-1, e.what());
    }
}

optik_robot *optik_robot_from_urdf_str(const char *urdf, const char *base_link, const char *ee_link) {
    optik_robot *r = nullptr;
    if (optik_robot_try_from_urdf_str(urdf, base_link, ee_link, &r)) panic(g_robot_err);
    return r;
}

optik_robot *optik_robot_from_urdf_file(const char *path, const char *base_link, const char *ee_link) {
    std::ifstream in(path ? path : "");
    if (!in) panic("error parsing URDF file!");  // lib.rs:55
    std::stringstream ss;
    ss << in.rdbuf();
    return optik_robot_from_urdf_str(ss.str().c_str(), base_link, ee_link);
}

void optik_robot_free(optik_robot *r) {
    if (!r) return;
    for (auto &c : r->devs) {
        if (!c->chain) continue;
        optik::DeviceScope dev_scope(c->device);  // the caller's device is current again afterwards
        optik_hip_chain_destroy(c->chain);
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        if (c->h_scratch) (void)hipHostFree(c->h_scratch);
        if (c->d_batch) (void)hipFree(c->d_batch);
        if (c->h_batch) (void)hipHostFree(c->h_batch);
    }
    delete r;
}

int optik_robot_set_devices(optik_robot *r, const int32_t *device_ids, int32_t count) {
    if (!r || count < 0 || (count > 0 && !device_ids)) return set_err(-1, "bad argument");
    std::lock_guard<std::mutex> lock(r->mu);
    if (!r->devs.empty()) return set_err(-1, "set_devices must be called before the robot's first GPU call");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) have = 0;
    for (int i = 0; i < count; ++i)
        if (device_ids[i] < 0 || device_ids[i] >= have) return set_err(-1, "no such HIP device");
    r->device_ids.assign(device_ids, device_ids + count);
    return 0;
}

int32_t optik_robot_num_devices(const optik_robot *r) { return r ? (int32_t)device_count(r) : 0; }
int32_t optik_robot_last_parts(const optik_robot *r) { return r ? r->last_parts.load() : 0; }

void optik_robot_set_parallelism(optik_robot *r, unsigned int n) {
    // The GPU grid is sized from the device; n only selects Speed's early-exit rule (see optik_robot).
    if (r) r->parallelism = n;
}

unsigned int optik_robot_num_positions(const optik_robot *r) { return r ? (unsigned)r->n : 0u; }

const double *optik_robot_joint_limits(const optik_robot *r) {
    std::vector<double> v(r->lb);
    v.insert(v.end(), r->ub.begin(), r->ub.end());
    return malloc_copy(v.data(), v.size());
}

int optik_robot_fk_ex(const optik_robot *r, const double *x, const double *ee16, double *pose16) {
    double p7[7];
    if (fk_on_device(r, x, ee16, p7, nullptr)) return -1;
    mat16_from_pose7(p7, pose16);
    return 0;
}

int optik_robot_joint_jacobian_ex(const optik_robot *r, const double *x, const double *ee16, double *jac) {
    return fk_on_device(r, x, ee16, nullptr, jac);
}

const double *optik_robot_fk(const optik_robot *r, const double *x) {
    double m[16];
    if (optik_robot_fk_ex(r, x, nullptr, m)) panic(g_robot_err);
    return malloc_copy(m, 16);
}

const double *optik_robot_joint_jacobian(const optik_robot *r, const double *x) {
    std::vector<double> j(6 * (size_t)r->n);
    if (optik_robot_joint_jacobian_ex(r, x, nullptr, j.data())) panic(g_robot_err);
    return malloc_copy(j.data(), j.size());
}

const double *optik_robot_random_configuration(const optik_robot *r) {
    // rand::rng(): thread-local, non-deterministic (optik-cpp/src/lib.rs:122)
    thread_local std::mt19937_64 gen{std::random_device{}()};
    std::vector<double> q((size_t)r->n);
    for (int k = 0; k < r->n; ++k) {
        if (!std::isfinite(r->lb[k]) || !std::isfinite(r->ub[k]))
            panic("random_configuration: non-finite joint limits");  // rand: Uniform::new_inclusive fails
        q[k] = std::uniform_real_distribution<double>(r->lb[k], std::nextafter(r->ub[k], INFINITY))(gen);
        if (q[k] > r->ub[k]) q[k] = r->ub[k];
    }
    return malloc_copy(q.data(), q.size());
}

// Robot::ik, lib.rs:241-415.
int optik_robot_ik_ex(const optik_robot *r, const CSolverConfig *config, const double *target16,
                      const double *x0, const double *ee16, double *x_out, double *f_out,
                      uint64_t *winner_out) {
    return optik_robot_ik_pose(r, config, target16, 0u, x0, ee16, x_out, f_out, winner_out);
}

int optik_pose_from_matrix(const double *m16, uint32_t flags, double *pose7) {
    if (!m16 || !pose7) return set_err(-1, "null argument");
    if (flags & OPTIK_POSE_FROM_MATRIX) pose7_from_mat16_iterative(m16, pose7);
    else pose7_from_mat16(m16, pose7);
    return 0;
}

int optik_robot_ik_pose(const optik_robot *r, const CSolverConfig *config, const double *target16, uint32_t flags,
                        const double *x0, const double *ee16, double *x_out, double *f_out, uint64_t *winner_out) {
    if (!r || !config || !target16 || !x0) return set_err(-1, "null argument");
    const bool iterative = (flags & OPTIK_POSE_FROM_MATRIX) != 0;
    // lib.rs:251-254
    for (int i = 0; i < r->n; ++i)
        if (x0[i] < r->lb[i] || x0[i] > r->ub[i])
            return set_err(-2, "seed joint position outside of joint limits");
    DeviceCtx *c0 = device_ctx(r);
    if (!c0) return -1;
    double tgt7[7], ee7[7];
    if (iterative) pose7_from_mat16_iterative(target16, tgt7);
    else pose7_from_mat16(target16, tgt7);
    if (ee16) pose7_from_mat16(ee16, ee7);

    const auto start = std::chrono::steady_clock::now();
    auto elapsed = [&]() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    };
    const uint64_t max_restarts = config->max_restarts > 0 ? config->max_restarts : UINT64_MAX;  // lib.rs:273-277
    const bool quality = config->solution_mode == 1;
    // The first launch covers as many restart indices as the chip holds resident waves (one
    // restart per wave: no wave pays for the phases of 63 other restarts, and a third to a
    // half of the restarts succeed, so it almost always contains the answer); later launches
    // cover enough 64-lane tiles to fill every CU a few times over -- on every GPU of the robot
    // at once: device g of G takes the g-th contiguous part of the round's index range
    // (lib.rs:297-300 shards the same range over rayon workers) and the host keeps the
    // minimum of the G (key, index) records.
    // A call that is still running after those two launches (a hard target, or Quality with a large
    // budget) is throughput-bound: rounds of 1 M restarts per GPU, each one launch of the lane-per-restart form.
    const uint64_t cus = (uint64_t)(c0->num_cus > 0 ? c0->num_cus : 256);
    // restarts of the first, latency-sized launch: four per CU -- one per SIMD, the most the quad solver's one-restart-
    // per-wave form takes -- under the first-success rule (the more restarts race, the sooner the first one succeeds: the
    // fastest of 512 needs 18.6 evaluations on average, of 1024 17.9; 1024 waves run each a little slower, but since
    // the call returns on the first success the launch's tail no longer counts: 201 -> 194 us per call, 221 -> 216 us
    // back to back); 128 under the deterministic rule (parallelism 1: the answer is the lowest successful index,
    // nearly always below ten, and only the restarts below it decide when the launch ends: 685 -> 660 us per call,
    // profiles/r4g_single_call.txt)
#ifndef OPTIK_FIRST_PER_CU
#define OPTIK_FIRST_PER_CU 4  // (tools/build_lib_variant.py fpcN -DOPTIK_FIRST_PER_CU=N --only=robot_host.o: the comparison above)
#endif
    const uint64_t first_per_cu = OPTIK_FIRST_PER_CU;
    const uint64_t first_batch = (!quality && r->parallelism == 1) ? 128 : cus * first_per_cu;
    const uint64_t later_batch = cus * 2 * 64 * 2;
    // (a big round is one launch of the lane-per-restart form and worth ~1 M restarts, because every launch pays its
    // own few ms of drain; rounds 1-3 ran them on a streaming engine that the lane form beat at every size in round 4
    // and that was retired in round 5: profiles/r5a_engine_retire_probe.txt)
#ifndef OPTIK_QUALITY_BATCH_LOG2
#define OPTIK_QUALITY_BATCH_LOG2 22
#endif
    // restarts per GPU per big round: 1 M under Speed (a round that finds a solution ends the call); 4 M under Quality,
    // which runs every restart anyway -- one drain (~3.5 ms) per 4 M instead of per 1 M: a 4 M-restart call 138 -> 129 ms
    // (288 MB of per-restart keys, points and residuals per device)
    // Under a time budget the rounds stay at 1 M: a launch that meets its deadline still has to fetch, seed and publish
    // every item left and the selection runs over the whole round, so the overshoot of max_time grows with the round
    // (ADVICE r5); the 4 M rounds' gain was measured without a deadline.
    const uint64_t big_batch = (uint64_t)1 << ((quality && !(config->max_time > 0.0)) ? OPTIK_QUALITY_BATCH_LOG2 : 20);
    const uint64_t big_from = 262144;              // (from this many restarts left on: rounds of big_batch)
    const size_t G = device_count(r);
    const uint32_t speed_flags = OPTIK_HIP_IK_EARLY_EXIT | (r->parallelism != 1 ? OPTIK_HIP_IK_FIND_ANY : 0u);
    struct Part {
        DeviceCtx *ctx = nullptr;
        uint64_t begin = 0, end = 0, widx = UINT64_MAX;
        double wf = 0.0, wkey = 0.0;
        std::vector<double> wx;
        int rc = 0;
        std::string err;
    };
    bool have = false;
    double best_key = 0.0, best_f = 0.0;
    uint64_t best_idx = UINT64_MAX;
    std::vector<double> best_x((size_t)r->n);
    for (uint64_t begin = 0; begin < max_restarts;) {
        // lib.rs:393: stop issuing restarts once out of time
        double deadline = 0.0;
        if (config->max_time > 0.0) {
            deadline = config->max_time - elapsed();
            if (deadline <= 0.0) break;
        }
        // Quality with a restart budget and no time budget runs every restart whatever happens: no
        // latency-sized first launches (each of them waits for its slowest restart -- 1 000 restarts
        // took two launches of 2.7 ms), the whole range at once: one launch of the solve kernel
        // below ~260 000 restarts (big_from), rounds of 1 M from index 0 above.
        // (with a time budget too when the whole range is one solve-kernel launch: its waves watch the clock)
        const bool all_at_once = quality && config->max_restarts > 0
                                 && (config->max_time <= 0.0 || config->max_restarts < big_from);
        // (the latency-sized first launch stays on one GPU; a Quality range run at once is cut over the G devices as
        // soon as every part is a few waves per CU)
        const size_t g_round = (begin == 0 && !(all_at_once && max_restarts / G >= cus * 64)) ? 1 : G;
        const bool big_round = max_restarts - begin >= big_from && (all_at_once || begin >= first_batch + later_batch);
        const uint64_t batch = big_round ? big_batch
                               : all_at_once ? big_from
                               : begin == 0 ? first_batch : later_batch;
        // (several GPUs: what is left is cut evenly when it is less than a full round of each)
        uint64_t per_part = batch;
        if (g_round > 1 && max_restarts != UINT64_MAX) {
            const uint64_t even = (max_restarts - begin + g_round - 1) / g_round;
            if (even < per_part) per_part = even;
        }
        std::vector<Part> parts;
        const bool begin_was_zero = begin == 0;
        for (size_t g = 0; g < g_round && begin < max_restarts; ++g) {
            Part p;
            p.ctx = g == 0 ? c0 : device_ctx(r, g);
            if (!p.ctx) return -1;
            p.begin = begin;
            p.end = (max_restarts - begin > per_part) ? begin + per_part : max_restarts;
            p.wx.resize((size_t)r->n);
            begin = p.end;
            parts.push_back(std::move(p));
        }
        auto run_part = [&](Part &p) {
            p.rc = optik_hip_ik_host(p.ctx->chain, config, tgt7, x0, 1, ee16 ? ee7 : nullptr, p.begin, p.end,
                                     quality ? 0u : speed_flags, deadline,
                                     p.wx.data(), &p.wf, &p.widx, &p.wkey);
            if (p.rc) p.err = optik_hip_last_error();
        };
        if (begin_was_zero) r->last_parts.store(0);
        if ((int32_t)parts.size() > r->last_parts.load()) r->last_parts.store((int32_t)parts.size());
        if (parts.size() == 1) {
            run_part(parts[0]);
        } else {
            std::vector<std::thread> th;
            for (size_t g = 1; g < parts.size(); ++g) th.emplace_back(run_part, std::ref(parts[g]));
            run_part(parts[0]);
            for (auto &t : th) t.join();
        }
        bool found_round = false;
        for (const Part &p : parts) {
            if (p.rc) return set_err(-1, p.err);
            if (p.widx == UINT64_MAX) continue;
            found_round = true;
            // lib.rs:397-413: Quality keeps the solution closest to the seed, Speed the first one
            if (!have || p.wkey < best_key || (p.wkey == best_key && p.widx < best_idx)) {
                have = true; best_key = p.wkey; best_idx = p.widx; best_f = p.wf; best_x = p.wx;
            }
        }
        if (found_round && !quality) break;
    }
    if (!have) return 1;
    if (x_out) std::memcpy(x_out, best_x.data(), sizeof(double) * (size_t)r->n);
    if (f_out) *f_out = best_f;
    if (winner_out) *winner_out = best_idx;
    return 0;
}

namespace {

// optik_robot_ik_batch_ex for the targets of one GPU: rounds of `round` restart indices per
// target (sizes: see the loop), each round ONE launch of optik_hip_ik_batch (which picks the solver by
// launch size); targets already solved (Speed) drop out of later rounds; max_time is enforced inside a
// launch and between rounds.
int ik_batch_on_device(const optik_robot *r, DeviceCtx *c, const CSolverConfig *config, int32_t T,
                       const double *targets16, bool row_major, bool iterative, const double *x0, const double *ee7,
                       std::chrono::steady_clock::time_point start, double *x_out, double *f_out,
                       int32_t *found_out, std::string &err) {
    const int n = r->n;
    auto elapsed = [&]() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    };
    const uint64_t max_restarts = config->max_restarts > 0 ? config->max_restarts : UINT64_MAX;
    const bool quality = config->solution_mode == 1;
    // work items (target x restart index) per round: ~4 M -- 288 MB of per-restart keys, points and residuals
    const uint64_t round_items = (uint64_t)4 << 20;
    std::lock_guard<std::mutex> lock(c->batch_mu);
    optik::DeviceScope dev_scope(c->device);
    if (!dev_scope.ok()) { err = "hipSetDevice failed"; return -1; }

    std::vector<double> tgt7((size_t)T * 7), best_key((size_t)T, 0.0);
    std::vector<uint64_t> best_idx((size_t)T, UINT64_MAX);
    parallel_ranges((size_t)T, [&](size_t t0, size_t t1) {
        for (size_t t = t0; t < t1; ++t) {
            const double *m = targets16 + t * 16;
            double cm[16];
            if (row_major) {
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) cm[b * 4 + a] = m[a * 4 + b];
                m = cm;
            }
            if (iterative) pose7_from_mat16_iterative(m, &tgt7[t * 7]);
            else pose7_from_mat16(m, &tgt7[t * 7]);
        }
    });
    std::vector<int> live((size_t)T);
    for (int t = 0; t < T; ++t) live[t] = t;
    for (int t = 0; t < T; ++t) if (found_out) found_out[t] = 0;

    // one device block and its pinned mirror, kept with the context across calls
    const size_t n_in = (size_t)(7 + n) * (size_t)T, n_out = (size_t)(n + 3) * (size_t)T;
    if (n_in + n_out > c->batch_cap) {
        if (c->d_batch) (void)hipFree(c->d_batch);
        if (c->h_batch) (void)hipHostFree(c->h_batch);
        c->d_batch = nullptr; c->h_batch = nullptr; c->batch_cap = 0;
        if (hipMalloc(&c->d_batch, sizeof(double) * (n_in + n_out)) != hipSuccess
            || hipHostMalloc(&c->h_batch, sizeof(double) * (n_in + n_out)) != hipSuccess) {
            err = "batch workspace allocation failed";
            return -1;
        }
        c->batch_cap = n_in + n_out;
    }
    // Speed: the first round is latency-sized -- 128 indices: with a third of the restarts succeeding
    // it leaves no reachable target unsolved, and the abandoned indices of a round still cost the
    // solve kernel's groups a queue fetch each (measured per first-round size 16 / 32 / 64 / 128 / 256:
    // 1 024 targets 3.7 / 2.0 / 2.1 / 2.2 / 2.2 ms, 16 384 targets 9.8 / 9.5 / 8.7 / 8.3 / 9.4 ms; 32 768
    // targets 15.9 ms at 64, 13.5 at 128, 16.8 at 256); what is
    // still unsolved after it is hard or unreachable and throughput-bound, so every later round covers
    // four times as many indices (ten unreachable targets x 100 000 restarts are 8 rounds instead of 390)
    const uint64_t first_round = 128;
    uint64_t speed_round = first_round;
    for (uint64_t begin = 0; begin < max_restarts && !live.empty();) {
        double deadline = 0.0;
        if (config->max_time > 0.0) {
            deadline = config->max_time - elapsed();
            if (deadline <= 0.0) break;  // lib.rs:393
        }
        const size_t L = live.size();
        uint64_t round = quality ? 256 : speed_round;
        // (a round's items are capped -- down to 64 indices per target, which still leave no reachable target
        // unsolved; batches of more than 65 536 targets down to 8: nearly all of a round's higher indices are
        // abandoned unissued -- a third to a half of the restarts succeed -- and skipping an item still costs the
        // refilling wave a queue fetch; the handful of targets a short round leaves unsolved go through the next)
        const uint64_t min_round = L <= 65536 ? (first_round < 64 ? first_round : 64u) : 8u;
        const uint64_t cap_items = L <= 65536 ? round_items : 2 * round_items;
        while (round > min_round && round * (uint64_t)L > cap_items) round >>= 1;
        if (!quality && speed_round < ((uint64_t)1 << 40)) speed_round *= 4;
        // Quality runs every restart of every target to the end: nothing to gain from short rounds -- as many
        // indices per round as ~4 M items allow
        if (quality)
            while (round * 2 * (uint64_t)L <= round_items && round < max_restarts - begin) round <<= 1;
        const uint64_t end = (max_restarts - begin > round) ? begin + round : max_restarts;
        double *h_t = c->h_batch, *h_x0 = h_t + 7 * L, *h_out = h_x0 + (size_t)n * L;
        double *d_t = c->d_batch, *d_x0 = d_t + 7 * L, *d_wx = d_x0 + (size_t)n * L, *d_wf = d_wx + (size_t)n * L,
               *d_wk = d_wf + L;
        uint64_t *d_wi = reinterpret_cast<uint64_t *>(d_wk + L);
        parallel_ranges(L, [&](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; ++k) {
                std::memcpy(&h_t[k * 7], &tgt7[(size_t)live[k] * 7], sizeof(double) * 7);
                std::memcpy(&h_x0[k * n], &x0[(size_t)live[k] * n], sizeof(double) * (size_t)n);
            }
        });
        if (hipMemcpyAsync(d_t, h_t, sizeof(double) * (size_t)(7 + n) * L, hipMemcpyHostToDevice, nullptr) != hipSuccess) {
            err = "upload failed";
            return -1;
        }
        optik_hip_ik_outputs o;
        std::memset(&o, 0, sizeof o);
        o.d_win_x = d_wx; o.d_win_f = d_wf; o.d_win_idx = d_wi; o.d_win_key = d_wk;
        // One launch per round.  The first rounds of a Speed batch are latency-bound -- early exit abandons most of
        // their restarts, what is left is each target's few first restarts run to the end: restart-major hand-out
        // keeps only a few restarts per target in flight (optik_hip_ik_batch then stays on the quad solver's shorter
        // trip).  Everything else -- a Quality batch, the later rounds of a Speed batch (whatever 256 restarts did
        // not solve runs nearly all of its restarts) -- is throughput-bound: target-major, the lane-per-restart form
        // from one full load of the chip on.  (Rounds 1-4 ran those rounds, and the first round of batches of 40 960
        // targets or more, on a streaming engine: retired in round 5, the single launch is 15 - 28 % faster at every
        // size where the engine was used -- profiles/r5a_engine_retire_probe.txt.)
        const uint32_t mode_flags =
            quality ? 0u : (OPTIK_HIP_IK_EARLY_EXIT | (r->parallelism != 1 ? OPTIK_HIP_IK_FIND_ANY : 0u));
        const int rck = optik_hip_ik_batch(c->chain, config, d_t, d_x0, (int32_t)L, ee7, begin, end,
                                           mode_flags | ((!quality && begin < 256) ? OPTIK_HIP_IK_RESTART_MAJOR : 0u),
                                           deadline, &o, nullptr);
        if (rck) { err = optik_hip_last_error(); return -1; }
        if (hipMemcpyAsync(h_out, d_wx, sizeof(double) * (size_t)(n + 3) * L, hipMemcpyDeviceToHost, nullptr) != hipSuccess
            || hipStreamSynchronize(nullptr) != hipSuccess) {
            err = "download failed";
            return -1;
        }
        const double *wx = h_out, *wf = wx + (size_t)n * L, *wk = wf + L;
        const uint64_t *wi = reinterpret_cast<const uint64_t *>(wk + L);
        // every target appears once in `live`, so ranges of k touch disjoint targets
        std::vector<uint8_t> keep(L);
        parallel_ranges(L, [&](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; ++k) {
                const int t = live[k];
                keep[k] = 1;
                if (wi[k] == UINT64_MAX) continue;
                const bool better = best_idx[t] == UINT64_MAX || wk[k] < best_key[t]
                                    || (wk[k] == best_key[t] && wi[k] < best_idx[t]);
                if (better) {
                    best_idx[t] = wi[k]; best_key[t] = wk[k];
                    if (x_out) std::memcpy(&x_out[(size_t)t * n], &wx[k * n], sizeof(double) * (size_t)n);
                    if (f_out) f_out[t] = wf[k];
                    if (found_out) found_out[t] = 1;
                }
                if (!quality) keep[k] = 0;  // Speed: the first solution ends this target
            }
        });
        std::vector<int> still;
        for (size_t k = 0; k < L; ++k)
            if (keep[k]) still.push_back(live[k]);
        live.swap(still);
        begin = end;
    }
    return 0;
}

}  // namespace

// Many independent ik() calls at once (the motion-planning workload of examples/example.rs:
// a stream of targets, each with its own seed): every target gets the semantics of Robot::ik
// with the same SolverConfig.  The targets are split into contiguous parts over the robot's
// GPUs (BASELINE.json config 5: no collective, the host gathers), each part runs on its GPU
// from its own host thread.
int optik_robot_ik_batch_ex(const optik_robot *r, const CSolverConfig *config, int32_t T,
                            const double *targets16, const double *x0, const double *ee16, double *x_out,
                            double *f_out, int32_t *found_out) {
    return optik_robot_ik_batch_poses(r, config, T, targets16, OPTIK_POSE_FROM_MATRIX, x0, ee16, x_out, f_out, found_out);
}

int optik_robot_ik_batch_poses(const optik_robot *r, const CSolverConfig *config, int32_t T,
                               const double *targets16, uint32_t flags, const double *x0, const double *ee16,
                               double *x_out, double *f_out, int32_t *found_out) {
    if (!r || !config || !targets16 || !x0 || T < 1) return set_err(-1, "bad argument");
    const int n = r->n;
    const bool row_major = (flags & OPTIK_BATCH_ROW_MAJOR) != 0;
    if (flags & OPTIK_BATCH_VALIDATE_POSES) {
        // nalgebra try_convert::<Matrix4, Isometry3> (optik-py/src/lib.rs:8-15): bottom row exactly
        // (0, 0, 0, 1), R'R = I within 100 eps per entry, det R > 0.  Neither test depends on
        // whether the 3x3 block is read by rows or by columns up to the bottom-row position.
        const double eps = 100.0 * 2.220446049250313e-16;
        std::atomic<bool> all_ok{true};
        parallel_ranges((size_t)T, [&](size_t t0, size_t t1) {
        for (size_t t = t0; t < t1; ++t) {
            const double *m = targets16 + t * 16;
            auto M = [&](int a, int b) { return row_major ? m[a * 4 + b] : m[b * 4 + a]; };
            bool ok = M(3, 0) == 0.0 && M(3, 1) == 0.0 && M(3, 2) == 0.0 && M(3, 3) == 1.0;
            for (int a = 0; a < 3 && ok; ++a)
                for (int b = 0; b < 3 && ok; ++b) {
                    const double d = M(0, a) * M(0, b) + M(1, a) * M(1, b) + M(2, a) * M(2, b) - (a == b ? 1.0 : 0.0);
                    ok = std::fabs(d) <= eps;  // false for NaN
                }
            if (ok) {
                const double det = M(0, 0) * (M(1, 1) * M(2, 2) - M(1, 2) * M(2, 1))
                                   - M(0, 1) * (M(1, 0) * M(2, 2) - M(1, 2) * M(2, 0))
                                   + M(0, 2) * (M(1, 0) * M(2, 1) - M(1, 1) * M(2, 0));
                ok = det > 0.0;
            }
            if (!ok) { all_ok = false; return; }
        }
        });
        if (!all_ok) return set_err(-3, "invalid target transform specified");
    }
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < n; ++i)
            if (x0[(size_t)t * n + i] < r->lb[i] || x0[(size_t)t * n + i] > r->ub[i])
                return set_err(-2, "seed joint position outside of joint limits");
    double ee7[7];
    if (ee16) pose7_from_mat16(ee16, ee7);
    const auto start = std::chrono::steady_clock::now();
    size_t G = device_count(r);
    if (G > (size_t)T) G = (size_t)T;
    struct Part { DeviceCtx *ctx; int32_t t0, t1; int rc; std::string err; };
    std::vector<Part> parts;
    for (size_t g = 0; g < G; ++g) {
        Part p{device_ctx(r, g), (int32_t)((int64_t)T * (int64_t)g / (int64_t)G),
               (int32_t)((int64_t)T * (int64_t)(g + 1) / (int64_t)G), 0, {}};
        if (!p.ctx) return -1;
        parts.push_back(p);
    }
    r->last_parts.store((int32_t)parts.size());
    auto run_part = [&](Part &p) {
        p.rc = ik_batch_on_device(r, p.ctx, config, p.t1 - p.t0, targets16 + (size_t)p.t0 * 16, row_major,
                                  (flags & OPTIK_POSE_FROM_MATRIX) != 0,
                                  x0 + (size_t)p.t0 * n, ee16 ? ee7 : nullptr, start,
                                  x_out ? x_out + (size_t)p.t0 * n : nullptr, f_out ? f_out + p.t0 : nullptr,
                                  found_out ? found_out + p.t0 : nullptr, p.err);
    };
    if (parts.size() == 1) {
        run_part(parts[0]);
    } else {
        std::vector<std::thread> th;
        for (size_t g = 1; g < parts.size(); ++g) th.emplace_back(run_part, std::ref(parts[g]));
        run_part(parts[0]);
        for (auto &t : th) t.join();
    }
    for (const Part &p : parts)
        if (p.rc) return set_err(-1, p.err);
    return 0;
}

const double *optik_robot_ik(const optik_robot *r, const CSolverConfig *config, const double *target,
                             const double *x0) {
    std::vector<double> x((size_t)r->n);
    // (the C path converts the target with the iterative UnitQuaternion::from_matrix: optik-cpp/src/lib.rs:141-142)
    const int rc = optik_robot_ik_pose(r, config, target, OPTIK_POSE_FROM_MATRIX, x0, nullptr, x.data(), nullptr, nullptr);
    if (rc < 0) panic(g_robot_err);
    if (rc == 1) return nullptr;  // optik-cpp/src/lib.rs:158-160
    return malloc_copy(x.data(), x.size());
}

// Robot::diff_ik, lib.rs:123-239: the largest 0 <= alpha <= 1 for which joint velocities v with
// |v_i| <= v_max_i realise the end-effector twist alpha * V_WE (world frame):
//     max alpha   s.t.   J_W(q) v = alpha V,   -v_max <= v <= v_max,   0 <= alpha <= 1.
// The reference hands this LP to Clarabel (an interior-point solver; un-vendored).  It has at
// most 8 unknowns, so it is solved exactly here: the equality constraints are eliminated
// (null space of [J_W | -V] by Gauss-Jordan with full pivoting) and the vertices of the
// remaining polytope (any dimension: d = n + 1 - rank) are enumerated.  The optimal alpha is
// unique.  Where the reference's own code can run the optimal v is unique too: lib.rs:196-197
// sizes the equality block as `b.extend(vec![0.0; n]); K.push(ZeroConeT(n))` for a 6-row
// matrix, so DefaultSolver::new only accepts n = 6 (for any other n the dimensions disagree
// and the `expect("solver initialization failed")` panics), and a non-singular 6 x 6 Jacobian
// leaves a single ray v = alpha J^-1 V (tests: closed form on UR3e).  For n != 6 -- an
// extension -- and at singularities the optimal face may have positive dimension: for d = 2
// the minimum-norm point of the optimal edge is returned, for d > 2 an optimal vertex of
// minimum norm among the vertices (an interior-point solver would return a point inside the
// face).  FK and the Jacobian come from the HIP kernels.
int optik_robot_diff_ik_ex(const optik_robot *r, const double *x0, const double *V_WE, const double *v_max,
                           const double *ee16, double *alpha_out, double *v_out) {
    if (!r || !x0 || !V_WE || !v_max) return set_err(-1, "null argument");
    const int n = r->n, nz = n + 1;
    if (n > 8)
        return set_err(-1, "diff_ik: chains of more than 8 joint positions are not supported (the reference's own "
                           "diff_ik only runs for n = 6: lib.rs:196-197 builds a 6-row block for n columns)");
    double p7[7];
    std::vector<double> jac(6 * (size_t)n);
    if (fk_on_device(r, x0, ee16, p7, jac.data())) return -1;
    for (int i = 0; i < n; ++i)
        if (!(v_max[i] >= 0.0)) return 1;  // infeasible box
    // body-frame Jacobian -> world frame: both 3-row blocks rotated by R_WE (lib.rs:190-197)
    const double qi = p7[3], qj = p7[4], qk = p7[5], qw = p7[6];
    const double R[3][3] = {{qw * qw + qi * qi - qj * qj - qk * qk, 2 * (qi * qj - qw * qk), 2 * (qw * qj + qi * qk)},
                            {2 * (qw * qk + qi * qj), qw * qw - qi * qi + qj * qj - qk * qk, 2 * (qj * qk - qw * qi)},
                            {2 * (qi * qk - qw * qj), 2 * (qw * qi + qj * qk), qw * qw - qi * qi - qj * qj + qk * qk}};
    double M[6][9];  // [J_W | -V], 6 x (n + 1)
    for (int c = 0; c < n; ++c)
        for (int blk = 0; blk < 2; ++blk)
            for (int a = 0; a < 3; ++a) {
                double acc = 0.0;
                for (int b = 0; b < 3; ++b) acc += R[a][b] * jac[(size_t)c * 6 + blk * 3 + b];
                M[blk * 3 + a][c] = acc;
            }
    double scale = 0.0;
    for (int a = 0; a < 6; ++a) {
        M[a][n] = -V_WE[a];
        for (int c = 0; c < nz; ++c) scale = std::max(scale, std::fabs(M[a][c]));
    }
    // Gauss-Jordan with full pivoting: pivot columns pc[0..rank), the others are free
    int pc[6], rank = 0;
    bool is_pivot[9] = {false};
    for (int step = 0; step < 6; ++step) {
        int br = -1, bc = -1;
        double best = 1e-12 * (scale > 0.0 ? scale : 1.0);
        for (int a = step; a < 6; ++a)
            for (int c = 0; c < nz; ++c)
                if (!is_pivot[c] && std::fabs(M[a][c]) > best) { best = std::fabs(M[a][c]); br = a; bc = c; }
        if (br < 0) break;
        for (int c = 0; c < nz; ++c) std::swap(M[step][c], M[br][c]);
        const double piv = M[step][bc];
        for (int c = 0; c < nz; ++c) M[step][c] /= piv;
        for (int a = 0; a < 6; ++a)
            if (a != step) {
                const double f = M[a][bc];
                if (f != 0.0) for (int c = 0; c < nz; ++c) M[a][c] -= f * M[step][c];
            }
        is_pivot[bc] = true;
        pc[rank++] = bc;
    }
    const int d = nz - rank;  // dimension of {z = (v, alpha) : [J_W | -V] z = 0}
    std::vector<double> best_z((size_t)nz, 0.0);  // z = 0 (alpha = 0, v = 0) is always feasible
    double best_alpha = 0.0, best_norm = 0.0;
    if (d >= 1) {
        // basis B (nz x d): free variable k = 1, pivot variables from the reduced rows
        int freec[9], nf = 0;
        for (int c = 0; c < nz; ++c) if (!is_pivot[c]) freec[nf++] = c;
        double B[9][9];
        for (int k = 0; k < d; ++k) {
            for (int c = 0; c < nz; ++c) B[c][k] = 0.0;
            B[freec[k]][k] = 1.0;
            for (int rr = 0; rr < rank; ++rr) B[pc[rr]][k] = -M[rr][freec[k]];
        }
        // half-spaces lo_c <= (B t)_c <= hi_c; vertices = d of them tight
        const int nh = 2 * nz;
        auto bound = [&](int h, double &sgn) -> double {  // constraint h: sgn * (B t)_c <= value
            const int c = h / 2;
            const bool upper = (h % 2) == 0;
            sgn = upper ? 1.0 : -1.0;
            if (c == n) return upper ? 1.0 : 0.0;
            return v_max[c];
        };
        const double tol = 1e-9;
        auto consider = [&](const double *t) {
            double z[9];
            for (int c = 0; c < nz; ++c) { z[c] = 0.0; for (int k = 0; k < d; ++k) z[c] += B[c][k] * t[k]; }
            for (int h = 0; h < nh; ++h) {
                double sgn; const double val = bound(h, sgn);
                if (sgn * z[h / 2] > val + tol * (1.0 + val)) return;
            }
            double nrm = 0.0;
            for (int c = 0; c < n; ++c) nrm += z[c] * z[c];
            if (z[n] > best_alpha + 1e-12 || (std::fabs(z[n] - best_alpha) <= 1e-12 && nrm < best_norm)) {
                best_alpha = z[n]; best_norm = nrm;
                for (int c = 0; c < nz; ++c) best_z[(size_t)c] = z[c];
            }
        };
        auto vertex = [&](const int *idx) {  // the point where the d constraints idx[] are tight
            double A[9][10];
            for (int q = 0; q < d; ++q) {
                double sgn; const double val = bound(idx[q], sgn);
                for (int k = 0; k < d; ++k) A[q][k] = sgn * B[idx[q] / 2][k];
                A[q][d] = val;
            }
            for (int q = 0; q < d; ++q) {  // Gauss-Jordan, partial pivoting
                int pr = q;
                for (int a = q + 1; a < d; ++a) if (std::fabs(A[a][q]) > std::fabs(A[pr][q])) pr = a;
                if (std::fabs(A[pr][q]) < 1e-13) return;  // the constraints are parallel: no vertex
                for (int k = 0; k <= d; ++k) std::swap(A[q][k], A[pr][k]);
                for (int a = 0; a < d; ++a)
                    if (a != q) {
                        const double f = A[a][q] / A[q][q];
                        for (int k = q; k <= d; ++k) A[a][k] -= f * A[q][k];
                    }
            }
            double t[9];
            for (int q = 0; q < d; ++q) t[q] = A[q][d] / A[q][q];
            consider(t);
        };
        // every choice of d of the nh half-spaces (d <= 9, nh <= 18: at most 48 620 small solves,
        // and d > 2 only at a kinematic singularity)
        int idx[9];
        for (int q = 0; q < d; ++q) idx[q] = q;
        for (bool more = d <= nh; more;) {
            vertex(idx);
            int q = d - 1;
            while (q >= 0 && idx[q] == nh - d + q) --q;
            if (q < 0) { more = false; break; }
            ++idx[q];
            for (int k = q + 1; k < d; ++k) idx[k] = idx[k - 1] + 1;
        }
        // a redundant arm at the optimum: slide along the optimal face to the minimum-norm v
        if (d == 2) {
            // direction inside the face: alpha fixed -> B[n] . dt = 0
            const double dt[2] = {-B[n][1], B[n][0]};
            double dz[9], dd = 0.0, zd = 0.0;
            for (int c = 0; c < nz; ++c) dz[c] = B[c][0] * dt[0] + B[c][1] * dt[1];
            for (int c = 0; c < n; ++c) { dd += dz[c] * dz[c]; zd += best_z[(size_t)c] * dz[c]; }
            if (dd > 0.0) {
                double lo = -1e300, hi = 1e300;  // feasible range of the step along dz
                for (int c = 0; c < n; ++c) {
                    if (std::fabs(dz[c]) < 1e-14) continue;
                    double a1 = (-v_max[c] - best_z[(size_t)c]) / dz[c], a2 = (v_max[c] - best_z[(size_t)c]) / dz[c];
                    if (a1 > a2) std::swap(a1, a2);
                    lo = std::max(lo, a1); hi = std::min(hi, a2);
                }
                double step = -zd / dd;
                step = std::min(std::max(step, lo), hi);
                if (lo <= hi && std::isfinite(step))
                    for (int c = 0; c < n; ++c) best_z[(size_t)c] += step * dz[c];
            }
        }
    }
    if (alpha_out) *alpha_out = std::min(std::max(best_z[(size_t)n], 0.0), 1.0);
    if (v_out) for (int c = 0; c < n; ++c) v_out[c] = best_z[(size_t)c];
    return 0;
}

// lib.rs:165-183 of optik-cpp: the joint velocities only, malloc'ed; NULL = no solution.
const double *optik_robot_diff_ik(const optik_robot *r, const double *x0, const double *V_WE, const double *v_max) {
    if (!r) panic("null robot");
    std::vector<double> v((size_t)r->n);
    double alpha = 0.0;
    const int rc = optik_robot_diff_ik_ex(r, x0, V_WE, v_max, nullptr, &alpha, v.data());
    if (rc < 0) panic(g_robot_err);
    if (rc != 0) return nullptr;
    return malloc_copy(v.data(), v.size());
}

int optik_robot_chain_tables_n(const optik_robot *r, int32_t capacity, int32_t *n_joints, double *origins7,
                               double *axes3, int32_t *types) {
    if (!r || !n_joints) return set_err(-1, "null argument");
    *n_joints = (int32_t)r->types.size();
    if ((origins7 || axes3 || types) && capacity < *n_joints)
        return set_err(-1, "chain_tables: buffers too small for the chain's joints");
    if (origins7) std::memcpy(origins7, r->origins.data(), sizeof(double) * r->origins.size());
    if (axes3) std::memcpy(axes3, r->axes.data(), sizeof(double) * r->axes.size());
    if (types) std::memcpy(types, r->types.data(), sizeof(int32_t) * r->types.size());
    return 0;
}

// (the original, in/out contract: with buffers, *n_joints holds their capacity in joints on entry -- a buffer that is
// too small is an error, never an overrun; without buffers it is written only)
int optik_robot_chain_tables(const optik_robot *r, int32_t *n_joints, double *origins7, double *axes3,
                             int32_t *types) {
    if (!r || !n_joints) return set_err(-1, "null argument");
    const int32_t capacity = (origins7 || axes3 || types) ? *n_joints : 0;
    return optik_robot_chain_tables_n(r, capacity, n_joints, origins7, axes3, types);
}

optik_hip_chain *optik_robot_hip_chain(const optik_robot *r) {
    if (!r) return nullptr;
    DeviceCtx *c = device_ctx(r);
    return c ? c->chain : nullptr;
}

}  // extern "C"
