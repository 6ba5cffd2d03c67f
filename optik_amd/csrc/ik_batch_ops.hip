// ik_batch_ops.hip -- the batched stand-alone operators of the kernel layer and the test probes.
//
//   eval_batch_kernel     objective + gradient (objective.rs:40-110) for a batch of configurations
//   fk_batch_kernel       end-effector pose (+ body Jacobian, kinematics.rs:123-196) for a batch
//   fk_general_kernel     forward kinematics of a chain with prismatic joints (kinematics.rs:243-255)
//   seed_batch_kernel     ChaCha8 restart seeds (lib.rs:358-370, 86-91)
//   probe_kernel, probe_math_kernel    elementary functions / the math.rs functions one at a time (test hooks)
// One configuration per lane, chain table staged in LDS, coalesced struct-of-arrays in and out.  All f64.
#include "ik_host.hpp"

using namespace optik;
using namespace optik::host;
using namespace optik::hostparams;

namespace {

struct EvalLaunch {
    const ChainDev *chain;
    EvalParams ep;
    double target[7];
    const double *q;  // [n][B]
    long long B;
    double *f;        // [B]
    double *g;        // [n][B] or null
};

template <int N, bool TIP>
__global__ __launch_bounds__(256) void eval_batch_kernel(const EvalLaunch a) {
    __shared__ ChainDev sch;
    stage_chain(sch, a.chain);
    const Pose target = load_pose(a.target);
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.B;
         b += (long long)gridDim.x * blockDim.x) {
        double q[N];
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = a.q[(size_t)i * a.B + b];
        // the gradient streams to its column as each component is known (as in eng_eval_kernel:
        // neither it nor the joint positions stay in registers across the evaluation)
        const double f = eval_fg_stream<N, TIP>(sch, a.ep, target, q, [&](int k, double v) {
            if (a.g) a.g[(size_t)k * a.B + b] = v;
        });
        a.f[b] = f;
    }
}

struct FkLaunch {
    const ChainDev *chain;
    EvalParams ep;  // only the ee_offset part is used
    const double *q;
    long long B;
    double *pose;  // [7][B]
    double *jac;   // [6n][B] or null
};

template <int N, bool TIP>
__global__ __launch_bounds__(256) void fk_batch_kernel(const FkLaunch a) {
    __shared__ ChainDev sch;
    stage_chain(sch, a.chain);
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.B;
         b += (long long)gridDim.x * blockDim.x) {
        double q[N];
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = a.q[(size_t)i * a.B + b];
        Kin<N, TIP> kin;
        forward_kinematics<N, TIP>(sch, a.ep, q, kin);
        const double p[7] = {kin.ee.t.x, kin.ee.t.y, kin.ee.t.z, kin.ee.q.i, kin.ee.q.j, kin.ee.q.k, kin.ee.q.w};
#pragma unroll
        for (int i = 0; i < 7; ++i) a.pose[(size_t)i * a.B + b] = p[i];
        if (a.jac) {
            // joint_jacobian, kinematics.rs:166-196
            const Q4 eeqc = qconj(kin.ee.q);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const V3 ax{sch.axis[k][0], sch.axis[k][1], sch.axis[k][2]};
                const V3 angular = qrot(kin.tf[k].q, ax);
                const V3 d{kin.ee.t.x - kin.tf[k].t.x, kin.ee.t.y - kin.tf[k].t.y, kin.ee.t.z - kin.tf[k].t.z};
                const V3 linear = cross(angular, d);
                const V3 al = qrot(eeqc, angular);
                const V3 ll = qrot(eeqc, linear);
                const double c6[6] = {ll.x, ll.y, ll.z, al.x, al.y, al.z};
#pragma unroll
                for (int r = 0; r < 6; ++r) a.jac[(size_t)(k * 6 + r) * a.B + b] = c6[r];
            }
        }
    }
}

// Chains with prismatic joints: the reference's FK handles them (kinematics.rs:243-255), its
// Jacobian -- and with it ik() -- does not (kinematics.rs:185: todo!()).  One generic kernel walks
// the joint table at run time: state = state * (origin_j * local_transform_j(q_j)).
struct FkGeneralLaunch {
    int32_t n_joints, n_pos;
    int32_t types[MAX_JOINTS];
    int32_t pad;
    double origin[MAX_JOINTS][7];
    double axis[MAX_JOINTS][3];
    double ee_offset[7];
    const double *q;  // [n][B]
    long long B;
    double *pose;     // [7][B]
};

__global__ __launch_bounds__(256) void fk_general_kernel(const FkGeneralLaunch a) {
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.B;
         b += (long long)gridDim.x * blockDim.x) {
        Pose state;
        state.t = V3{0.0, 0.0, 0.0};
        state.q = Q4{0.0, 0.0, 0.0, 1.0};
        int qi = 0;
        for (int j = 0; j < a.n_joints; ++j) {
            Pose local;
            local.t = V3{0.0, 0.0, 0.0};
            local.q = Q4{0.0, 0.0, 0.0, 1.0};
            if (a.types[j] == OPTIK_JOINT_REVOLUTE) {
                double s, c;
                sincos_dev(a.q[(size_t)qi * a.B + b] / 2.0, s, c);  // UnitQuaternion::from_axis_angle
                local.q = Q4{a.axis[j][0] * s, a.axis[j][1] * s, a.axis[j][2] * s, c};
                ++qi;
            } else if (a.types[j] == OPTIK_JOINT_PRISMATIC) {
                const double d = a.q[(size_t)qi * a.B + b];
                local.t = V3{a.axis[j][0] * d, a.axis[j][1] * d, a.axis[j][2] * d};
                ++qi;
            }
            const Pose jt = pose_mul(load_pose(a.origin[j]), local);
            state = pose_mul(state, jt);
        }
        const Pose ee = pose_mul(state, load_pose(a.ee_offset));
        const double p[7] = {ee.t.x, ee.t.y, ee.t.z, ee.q.i, ee.q.j, ee.q.k, ee.q.w};
#pragma unroll
        for (int i = 0; i < 7; ++i) a.pose[(size_t)i * a.B + b] = p[i];
    }
}

struct SeedLaunch {
    uint32_t key[8];
    double lb[MAX_DOF];
    double scale[MAX_DOF];
    unsigned long long first;
    long long count;
    double *q;  // [n][count]
};

template <int N>
__global__ __launch_bounds__(256) void seed_batch_kernel(const SeedLaunch a) {
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < a.count;
         b += (long long)gridDim.x * blockDim.x) {
        double q[N];
        restart_seed<N>(a.key, a.lb, a.scale, a.first + (unsigned long long)b, q);
#pragma unroll
        for (int i = 0; i < N; ++i) a.q[(size_t)i * a.count + b] = q[i];
    }
}

__global__ void probe_kernel(int op, const double *a, const double *b, long long count, double *out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (long long)gridDim.x * blockDim.x) {
        double r = 0.0, s, c;
        switch (op) {
        case 0: r = a[i] / b[i]; break;
        case 1: r = __builtin_sqrt(a[i]); break;
        case 2: sincos_dev(a[i], s, c); r = s; break;
        case 3: sincos_dev(a[i], s, c); r = c; break;
        default: r = atan2_q1(a[i], b[i]); break;
        }
        out[i] = r;
    }
}

// math.rs functions one at a time (test hook: compared with the reference's own golden vectors,
// /root/reference/crates/optik/tests/test_math.rs:14-61).  pose = t[3], quat[i,j,k,w]; matrices row-major.
// op 0 so3::log (3), 1 so3::right_jacobian(so3::log(q)) (9), 2 se3::log (6: V^-1 t, w),
// 3 se3::right_jacobian (36: [[J, Q], [0, J]]).
__global__ void probe_math_kernel(int op, const double *poses, long long count, double *out, int stride) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (long long)gridDim.x * blockDim.x) {
        const Pose X = load_pose(poses + i * 7);
        double *o = out + i * stride;
        const V3 w = so3_log(X.q);
        if (op == 0) { o[0] = w.x; o[1] = w.y; o[2] = w.z; continue; }
        const RotTerms rt = rot_terms(w);
        const M3 Jr = so3_right_jacobian(rt);
        if (op == 1) {
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[r * 3 + c] = Jr.m[r][c];
        } else if (op == 2) {
            const V3 lin = se3_log_linear(rt, X.t);
            o[0] = lin.x; o[1] = lin.y; o[2] = lin.z; o[3] = w.x; o[4] = w.y; o[5] = w.z;
        } else {
            const M3 Q = se3_q_matrix(rt, X.t, Jr);
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    o[r * 6 + c] = Jr.m[r][c];
                    o[r * 6 + 3 + c] = Q.m[r][c];
                    o[(r + 3) * 6 + c] = 0.0;
                    o[(r + 3) * 6 + 3 + c] = Jr.m[r][c];
                }
        }
    }
}

}  // namespace

extern "C" {

int optik_hip_eval_batch(const optik_hip_chain *ch, const optik_solver_config *cfg, const double *target7,
                         const double *ee_offset7, const double *d_q, int64_t B, double *d_f, double *d_g,
                         void *stream) {
    if (!ch || !cfg || !target7 || !d_q || !d_f || B < 0) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (B == 0) return 0;
    if (ch->prismatic)
        return fail(OPTIK_HIP_EUNSUPPORTED,
                    "prismatic joints: only forward kinematics is available (the reference's Jacobian panics, kinematics.rs:185)");
    BIND_DEVICE(ch);
    if (ch->wide) {
        WideBatchLaunch w;
        std::memset(&w, 0, sizeof w);
        w.chain = ch->wdev;
        make_eval_params(cfg->linear_weight, cfg->angular_weight, ee_offset7, w.ep);
        std::memcpy(w.target, target7, sizeof w.target);
        w.q = d_q; w.B = B; w.f = d_f; w.g = d_g;
        HIP_TRY(wide_batch_launch(0, grid_for(ch, B, 256, 8), (hipStream_t)stream, w));
        return 0;
    }
    EvalLaunch a;
    a.chain = ch->dev;
    make_eval_params(cfg->linear_weight, cfg->angular_weight, ee_offset7, a.ep);
    std::memcpy(a.target, target7, sizeof a.target);
    a.q = d_q; a.B = B; a.f = d_f; a.g = d_g;
    const int grid = grid_for(ch, B, 256, 8);
#define CALL(NN, TT) hipLaunchKernelGGL((eval_batch_kernel<NN, TT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a)
    OPTIK_DISPATCH(ch, CALL);
#undef CALL
    HIP_TRY(hipGetLastError());
    return 0;
}

int optik_hip_fk_batch(const optik_hip_chain *ch, const double *ee_offset7, const double *d_q, int64_t B,
                       double *d_pose, double *d_jac, void *stream) {
    if (!ch || !d_q || !d_pose || B < 0) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (B == 0) return 0;
    BIND_DEVICE(ch);
    if (ch->prismatic) {
        if (d_jac)
            return fail(OPTIK_HIP_EUNSUPPORTED,
                        "joint_jacobian: prismatic joints are not implemented (the reference panics: kinematics.rs:185 todo!())");
        FkGeneralLaunch g;
        std::memset(&g, 0, sizeof g);
        g.n_joints = ch->n_joints;
        g.n_pos = ch->n;
        std::memcpy(g.types, ch->types, sizeof g.types);
        std::memcpy(g.origin, ch->host.origin, sizeof g.origin);
        std::memcpy(g.axis, ch->axis_all, sizeof g.axis);
        const double ident[7] = {0, 0, 0, 0, 0, 0, 1};
        std::memcpy(g.ee_offset, ee_offset7 ? ee_offset7 : ident, sizeof ident);
        g.q = d_q; g.B = B; g.pose = d_pose;
        hipLaunchKernelGGL(fk_general_kernel, dim3(grid_for(ch, B, 256, 8)), dim3(256), 0, (hipStream_t)stream, g);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (ch->wide) {
        WideBatchLaunch w;
        std::memset(&w, 0, sizeof w);
        w.chain = ch->wdev;
        const double one_w[3] = {1, 1, 1};
        make_eval_params(one_w, one_w, ee_offset7, w.ep);
        w.q = d_q; w.B = B; w.pose = d_pose; w.jac = d_jac;
        HIP_TRY(wide_batch_launch(1, grid_for(ch, B, 256, 8), (hipStream_t)stream, w));
        return 0;
    }
    FkLaunch a;
    a.chain = ch->dev;
    const double one[3] = {1, 1, 1};
    make_eval_params(one, one, ee_offset7, a.ep);
    a.q = d_q; a.B = B; a.pose = d_pose; a.jac = d_jac;
    const int grid = grid_for(ch, B, 256, 8);
#define CALL(NN, TT) hipLaunchKernelGGL((fk_batch_kernel<NN, TT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a)
    OPTIK_DISPATCH(ch, CALL);
#undef CALL
    HIP_TRY(hipGetLastError());
    return 0;
}

int optik_hip_seed_batch(const optik_hip_chain *ch, uint64_t first, int64_t count, double *d_q, void *stream) {
    if (!ch || !d_q || count < 0) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (count == 0) return 0;
    for (int k = 0; k < ch->n; ++k)
        if (std::isnan(ch->scale[k]))
            return fail(OPTIK_HIP_EINVAL, "random restarts need finite joint limits (reference: random_range panics)");
    BIND_DEVICE(ch);
    if (ch->wide) {
        WideBatchLaunch w;
        std::memset(&w, 0, sizeof w);
        w.chain = ch->wdev;
        std::memcpy(w.key, ch->key, sizeof w.key);
        w.first = first; w.B = count; w.q_out = d_q;
        HIP_TRY(wide_batch_launch(2, grid_for(ch, count, 256, 8), (hipStream_t)stream, w));
        return 0;
    }
    SeedLaunch a;
    std::memcpy(a.key, ch->key, sizeof a.key);
    std::memcpy(a.lb, ch->host.lb, sizeof a.lb);
    std::memcpy(a.scale, ch->scale, sizeof a.scale);
    a.first = first; a.count = count; a.q = d_q;
    const int grid = grid_for(ch, count, 256, 8);
#define CALL(NN, TT) hipLaunchKernelGGL((seed_batch_kernel<NN>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a)
    OPTIK_DISPATCH(ch, CALL);
#undef CALL
    HIP_TRY(hipGetLastError());
    return 0;
}

int optik_hip_probe(int32_t op, const double *a, const double *b, int64_t count, double *out) {
    if (!a || !out || count < 0) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (int rc = ensure_device()) return rc;
    if (count == 0) return 0;
    double *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
    const size_t bytes = sizeof(double) * (size_t)count;
    HIP_TRY(hipMalloc(&d_a, bytes));
    HIP_TRY(hipMalloc(&d_b, bytes));
    HIP_TRY(hipMalloc(&d_o, bytes));
    HIP_TRY(hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_b, b ? b : a, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_kernel, dim3(1024), dim3(256), 0, nullptr, op, d_a, d_b, (long long)count, d_o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost));
    hipFree(d_a); hipFree(d_b); hipFree(d_o);
    return 0;
}

int optik_hip_probe_math(int32_t op, const double *poses7, int64_t count, double *out) {
    if (!poses7 || !out || count < 0 || op < 0 || op > 3) return fail(OPTIK_HIP_EINVAL, "bad argument");
    if (int rc = ensure_device()) return rc;
    if (count == 0) return 0;
    const int stride = op == 0 ? 3 : (op == 1 ? 9 : (op == 2 ? 6 : 36));
    // (both buffers are released on every path out)
    struct DevBuf {
        double *p = nullptr;
        ~DevBuf() { if (p) (void)hipFree(p); }
    } d_p, d_o;
    HIP_TRY(hipMalloc(&d_p.p, sizeof(double) * 7 * (size_t)count));
    HIP_TRY(hipMalloc(&d_o.p, sizeof(double) * (size_t)stride * (size_t)count));
    HIP_TRY(hipMemcpy(d_p.p, poses7, sizeof(double) * 7 * (size_t)count, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_math_kernel, dim3(64), dim3(64), 0, nullptr, op, d_p.p, (long long)count, d_o.p, stride);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_o.p, sizeof(double) * (size_t)stride * (size_t)count, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
