// ik_eval.hpp -- fused objective + gradient for one joint configuration per lane.
//
// Restates, in one pass, what the reference does per NLopt callback
// (/root/reference/crates/optik/src/lib.rs:305-337):
//   forward_kinematics_mut   kinematics.rs:123-164
//   joint_jacobian           kinematics.rs:166-196   (body frame, 6 x n)
//   objective_grad           objective.rs:60-110     g = 2 (W^2 e)' Jlog6(X) J
//   objective                objective.rs:40-57      f = ||W e||^2
// The chain table is read from LDS (wave-uniform addresses -> broadcast reads);
// the per-lane state (q, joint transforms, g) stays in VGPRs.  Chains are N
// revolute joints plus an optional trailing fixed joint -- the only shape the
// reference's fixed-joint folding (kinematics.rs:64-97) produces, and prismatic
// joints panic in its Jacobian (kinematics.rs:185).
#pragma once

#include "ik_math.hpp"

// Keeps the machine scheduler from interleaving the iterations of the fully
// unrolled per-joint loops: without it every joint's temporaries are live at once
// (2x the registers of the rolled loop) and the restart kernel spills.
#ifdef OPTIK_NO_SCHED_FENCE
#define OPTIK_SCHED_FENCE()
#else
#define OPTIK_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// (by file, for experiments: -DOPTIK_NOFENCE_EVAL / _SLSQP drop the fences of ik_eval.hpp / ik_slsqp.hpp alone: the
// first costs the lane kernel a quarter of its rate, the second ~0.5 %)
#ifdef OPTIK_NOFENCE_EVAL
#define OPTIK_SCHED_FENCE_EVAL()
#else
#define OPTIK_SCHED_FENCE_EVAL() OPTIK_SCHED_FENCE()
#endif
#ifdef OPTIK_NOFENCE_SLSQP
#define OPTIK_SCHED_FENCE_SLSQP()
#else
#define OPTIK_SCHED_FENCE_SLSQP() OPTIK_SCHED_FENCE()
#endif
// (the thirteen fences between the phases of ik_lane64.hpp's trip are OFF since the lane kernel is scheduled with the
// iterative-ILP strategy: 29.16 -> 29.48 M restarts/s without them, tools/ab_kernel_path.sh; -DOPTIK_FENCE_LANE64 puts
// them back)
#ifdef OPTIK_FENCE_LANE64
#define OPTIK_SCHED_FENCE_LANE64() OPTIK_SCHED_FENCE()
#else
#define OPTIK_SCHED_FENCE_LANE64()
#endif

namespace optik {

constexpr int MAX_DOF = 8;
constexpr int MAX_JOINTS = MAX_DOF + 1;

// Flat chain table.  Lives in HBM once per robot; every kernel stages it into
// LDS in its prologue.
struct ChainDev {
    int32_t n_pos;     // n
    int32_t has_tip;   // trailing fixed joint present
    int32_t pad0, pad1;
    double origin[MAX_JOINTS][7];  // per chain joint: t[3], quat[i,j,k,w]
    double axis[MAX_DOF][3];       // unit axis of revolute joint k
    double lb[MAX_DOF];
    double ub[MAX_DOF];
};

// Objective parameters, wave-uniform (kernel arguments -> SGPRs).
struct EvalParams {
    double w_lin[3], w_ang[3];    // objective weights          (objective.rs:52)
    double w_lin2[3], w_ang2[3];  // squared, for the gradient  (objective.rs:102-104)
    int32_t skip_lin, skip_ang;   // nalgebra is_identity(w) (quirk Q2), decided on the host
    int32_t skip_lin2, skip_ang2;
    int32_t grad_same_as_value;   // w2 == w bitwise: weighted e is shared
    int32_t has_ee_offset;
    double ee_offset[7];
};

OPTIK_DEV Pose load_pose(const double *p) {
    Pose o;
    o.t = V3{p[0], p[1], p[2]};
    o.q = Q4{p[3], p[4], p[5], p[6]};
    return o;
}

template <int N, bool TIP>
struct Kin {
    Pose tf[N];  // T_w_j after joint j's own rotation (kinematics.rs:153-156)
    Pose ee;     // kinematics.rs:163
};

// forward_kinematics_mut, kinematics.rs:123-164.
template <int N, bool TIP>
OPTIK_DEV void forward_kinematics(const ChainDev &ch, const EvalParams &ep, const double (&q)[N],
                                  Kin<N, TIP> &kin) {
    Pose state;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double s, c;
        sincos_dev(q[j] / 2.0, s, c);  // UnitQuaternion::from_axis_angle
        const Q4 local{ch.axis[j][0] * s, ch.axis[j][1] * s, ch.axis[j][2] * s, c};
        Pose jt;  // joint.origin * local_transform(q): the translation part is exact
        jt.t = V3{ch.origin[j][0], ch.origin[j][1], ch.origin[j][2]};
        jt.q = qmul(Q4{ch.origin[j][3], ch.origin[j][4], ch.origin[j][5], ch.origin[j][6]}, local);
        state = (j == 0) ? jt : pose_mul(state, jt);  // identity * jt is exact
        kin.tf[j] = state;
        OPTIK_SCHED_FENCE_EVAL();
    }
    if (TIP) state = pose_mul(state, load_pose(ch.origin[N]));
    kin.ee = ep.has_ee_offset ? pose_mul(state, load_pose(ep.ee_offset)) : state;
}

// Value and gradient at q; gradient component k is handed to gsink(k, g_k) as soon as it is
// known (a caller that stores it right away never holds the gradient).  Returns f.
//
// Register diet: only the orientation of every joint frame is kept from the forward pass
// (4 doubles per joint instead of 7); the Jacobian loop walks the chain again for the
// positions, t_k = t_(k-1) + R_(k-1) * origin_k.t -- the same operations on the same operands
// as the forward pass, hence the same bits, for 33 flops per joint.
template <int N, bool TIP, class GSink>
OPTIK_DEV double eval_fg_stream(const ChainDev &ch, const EvalParams &ep, const Pose target,
                                const double (&q)[N], GSink &&gsink) {
    Q4 tfq[N];  // orientation of T_w_j after joint j's own rotation (kinematics.rs:153-156)
    Pose ee;    // kinematics.rs:163
    {
        Pose state;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double s, c;
            sincos_dev(q[j] / 2.0, s, c);  // UnitQuaternion::from_axis_angle
            const Q4 local{ch.axis[j][0] * s, ch.axis[j][1] * s, ch.axis[j][2] * s, c};
            Pose jt;  // joint.origin * local_transform(q): the translation part is exact
            jt.t = V3{ch.origin[j][0], ch.origin[j][1], ch.origin[j][2]};
            jt.q = qmul(Q4{ch.origin[j][3], ch.origin[j][4], ch.origin[j][5], ch.origin[j][6]}, local);
            state = (j == 0) ? jt : pose_mul(state, jt);  // identity * jt is exact
            tfq[j] = state.q;
            OPTIK_SCHED_FENCE_EVAL();
        }
        if (TIP) state = pose_mul(state, load_pose(ch.origin[N]));
        ee = ep.has_ee_offset ? pose_mul(state, load_pose(ep.ee_offset)) : state;
    }

    // X = T_target^-1 T_ee  (objective.rs:69-70)
    const Pose X = pose_inv_mul(target, ee);
    const V3 w = so3_log(X.q);
    const RotTerms rt = rot_terms(w);
    const M3 Jr = so3_right_jacobian(rt);          // math.rs:195
    const M3 Qm = se3_q_matrix(rt, X.t, Jr);       // math.rs:196 (E = Jr, math.rs:167)
    const V3 elin = se3_log_linear(rt, X.t);       // math.rs:120-122

    // weighted error for the value (objective.rs:52) and for the gradient (:104)
    V3 fl = elin, fa = w;
    if (!ep.skip_lin) fl = weight_block(target.q, elin, ep.w_lin);
    if (!ep.skip_ang) fa = weight_block(target.q, w, ep.w_ang);
    V3 gl = fl, ga = fa;
    if (!ep.grad_same_as_value) {
        gl = elin; ga = w;
        if (!ep.skip_lin2) gl = weight_block(target.q, elin, ep.w_lin2);
        if (!ep.skip_ang2) ga = weight_block(target.q, w, ep.w_ang2);
    }
    const double e2[6] = {2.0 * gl.x, 2.0 * gl.y, 2.0 * gl.z, 2.0 * ga.x, 2.0 * ga.y, 2.0 * ga.z};
    // f = ||e||^2 (objective.rs:56)
    const double ef[6] = {fl.x, fl.y, fl.z, fa.x, fa.y, fa.z};
    double f = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) f += ef[i] * ef[i];
    OPTIK_SCHED_FENCE_EVAL();

    // per joint: body-frame Jacobian column (kinematics.rs:173-184), then
    // Jtask = Jlog6 * J (objective.rs:81) and g = (2 e') Jtask (objective.rs:106-109)
    const Q4 eeqc = qconj(ee.q);
    V3 tk{ch.origin[0][0], ch.origin[0][1], ch.origin[0][2]};  // position of joint frame 0
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k > 0) {  // pose_mul(state_(k-1), jt_k).t
            const V3 sft = qrot(tfq[k - 1], V3{ch.origin[k][0], ch.origin[k][1], ch.origin[k][2]});
            tk = V3{tk.x + sft.x, tk.y + sft.y, tk.z + sft.z};
        }
        const V3 ax{ch.axis[k][0], ch.axis[k][1], ch.axis[k][2]};
        const V3 angular = qrot(tfq[k], ax);
        const V3 d{ee.t.x - tk.x, ee.t.y - tk.y, ee.t.z - tk.z};
        const V3 linear = cross(angular, d);
        const V3 al = qrot(eeqc, angular);
        const V3 ll = qrot(eeqc, linear);
        const double lin[3] = {ll.x, ll.y, ll.z};
        const double ang[3] = {al.x, al.y, al.z};
        double jt[6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m) acc += Jr.m[r][m] * lin[m];
#pragma unroll
            for (int m = 0; m < 3; ++m) acc += Qm.m[r][m] * ang[m];
            jt[r] = acc;
            double acc2 = 0.0;  // lower-left block of Jlog6 is zero
#pragma unroll
            for (int m = 0; m < 3; ++m) acc2 += Jr.m[r][m] * ang[m];
            jt[r + 3] = acc2;
        }
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) acc += e2[r] * jt[r];
        gsink(k, acc);
        OPTIK_SCHED_FENCE_EVAL();
    }
    return f;
}

// Value and gradient at q.  Returns f, writes g[N].
template <int N, bool TIP>
OPTIK_DEV double eval_fg(const ChainDev &ch, const EvalParams &ep, const Pose target,
                         const double (&q)[N], double (&g)[N]) {
    return eval_fg_stream<N, TIP>(ch, ep, target, q, [&](int k, double v) {
#pragma unroll
        for (int i = 0; i < N; ++i) g[i] = (i == k) ? v : g[i];
    });
}

}  // namespace optik
