/*
 * optik_hip.h -- C ABI of the MI355X kernel layer for optik's random-restart IK.
 *
 * This is the boundary a host written in the reference's own language binds
 * (Rust `extern "C"` / cgo-style FFI): plain pointers and
 * sizes, integer return codes, no C++ or torch types.  It replaces, behind
 * `Robot::ik` (/root/reference/crates/optik/src/lib.rs:241-415),
 *   - the rayon fan-out over restart indices        lib.rs:297-301, 393-395
 *   - the per-restart NLopt SLSQP solve             lib.rs:302-356, 372 (nlopt crate)
 *   - the ChaCha8 restart seeds                     lib.rs:358-370, 86-91 (rand crates)
 *   - objective / gradient / FK / Jacobian          objective.rs:40-110, kinematics.rs:123-196
 *   - classification and winner selection           lib.rs:376-390, 397-413
 *
 * Conventions: poses are 7 doubles [tx, ty, tz, qi, qj, qk, qw]; `d_` pointers
 * are device (HBM) memory, everything else is host memory; batched per-restart
 * arrays are struct-of-arrays ([component][batch]) so that lane-consecutive
 * accesses coalesce.  All calls are stream-ordered on `stream` (a hipStream_t
 * passed as void*, NULL = default stream) and return 0 on success or a negative
 * OPTIK_HIP_E* code; optik_hip_last_error() describes the last failure of the
 * calling thread.  There is no CPU fallback: without a usable GPU every compute
 * entry point fails with OPTIK_HIP_ENODEVICE.
 */
#ifndef OPTIK_HIP_H
#define OPTIK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPTIK_HIP_MAX_DOF 16 /* joint positions of a chain (9 .. 16: the general kernels, see below) */

enum {
    OPTIK_HIP_OK = 0,
    OPTIK_HIP_EINVAL = -1,     /* bad argument */
    OPTIK_HIP_EUNSUPPORTED = -2, /* chain shape the kernels do not cover */
    OPTIK_HIP_ENODEVICE = -3,  /* no HIP device / HIP runtime error */
    OPTIK_HIP_ENOMEM = -4
};

/* JointType, kinematics.rs:227-241 */
enum { OPTIK_JOINT_FIXED = 0, OPTIK_JOINT_REVOLUTE = 1, OPTIK_JOINT_PRISMATIC = 2 };

/* Per-restart status = NLopt's nlopt_result for the SLSQP run (lib.rs:372-379). */
enum {
    OPTIK_RES_FAILURE = -1,
    OPTIK_RES_ROUNDOFF_LIMITED = -4,
    OPTIK_RES_FORCED_STOP = -5,   /* abandoned: timeout or a lower index succeeded */
    OPTIK_RES_ITER_CAP = -100,
    OPTIK_RES_NOT_RUN = 0,
    OPTIK_RES_STOPVAL_REACHED = 2,
    OPTIK_RES_FTOL_REACHED = 3,
    OPTIK_RES_XTOL_REACHED = 4
};

/* SolverConfig with the layout of CSolverConfig
 * (crates/optik-cpp/src/lib.rs:10-20; config.rs:22-50): 96 bytes on LP64. */
typedef struct optik_solver_config {
    int32_t solution_mode; /* 1 = Quality, 2 = Speed (config.rs:3-8) */
    int32_t _pad;
    double max_time;       /* seconds, 0 = unlimited */
    uint64_t max_restarts; /* 0 = unlimited */
    double tol_f;
    double tol_df;
    double tol_dx;
    double linear_weight[3];
    double angular_weight[3];
} optik_solver_config;

/* Flat kinematic chain (the output of KinematicChain::from_urdf,
 * kinematics.rs:18-105, uploaded once per robot). */
typedef struct optik_hip_chain optik_hip_chain;

int optik_hip_device_count(void);
const char *optik_hip_last_error(void);

/* origins: n_joints x 7 poses (Joint::origin), axes: n_joints x 3 (unit axis of
 * each chain joint, ignored for fixed), types: OPTIK_JOINT_*, lb/ub: n limits
 * (Robot::joint_limits, lib.rs:78-84).  Supported: 1 <= n <= 16 positional joints plus
 * an optional trailing fixed joint (what from_urdf's folding produces).  n <= 8 runs on the
 * tuned solvers (the lane-per-restart form for n <= 7 from one full load of the chip on, the
 * quad solver otherwise); 9 <= n <= 16 runs on one general kernel per entry point (joint count at run
 * time; the solver keeps one restart per wave in LDS with the wave's 64 lanes working on it
 * together: the same results as the CPU oracle bit for bit, a fraction of the tuned solvers'
 * rate) -- the reference itself has no limit
 * (kinematics.rs:107-110).  Prismatic joints (n <= 8): forward kinematics only (the reference's
 * Jacobian is todo!(), kinematics.rs:185). */
int optik_hip_chain_create(const double *origins, const double *axes, const int32_t *types,
                           int32_t n_joints, const double *lb, const double *ub, int32_t n,
                           optik_hip_chain **out);
void optik_hip_chain_destroy(optik_hip_chain *chain);
int32_t optik_hip_chain_num_positions(const optik_hip_chain *chain);

/* Which rand 0.9.2 code path the restart seeds restate for `rng.random_range(lb..=ub)`
 * (lib.rs:89).  SINGLE_INCLUSIVE (default): UniformFloat::sample_single_inclusive, which is
 * what Rng::random_range dispatches to for a RangeInclusive<f64> -- scale = ub - lb.
 * NEW_INCLUSIVE: Uniform::new_inclusive(lb, ub).sample(rng) -- scale = (ub - lb) / (1 - eps),
 * decreased by ulps until scale * (1 - eps) + lb <= ub.  The two differ in the last bits
 * of every seed; the kernels only see the precomputed scale.  The environment variable
 * OPTIK_RANDOM_RANGE_RULE=new_inclusive selects the second rule for chains created after
 * it is set.  Not to be called while a launch on the chain is in flight. */
enum { OPTIK_HIP_RANGE_SINGLE_INCLUSIVE = 0, OPTIK_HIP_RANGE_NEW_INCLUSIVE = 1 };
int optik_hip_chain_set_range_rule(optik_hip_chain *chain, int32_t rule);
int32_t optik_hip_chain_range_rule(const optik_hip_chain *chain);

/* objective + objective_grad (objective.rs:40-110) for B configurations.
 * d_q [n][B] -> d_f [B], d_g [n][B] (d_g may be NULL).  Only the weights of
 * `cfg` are read.  ee_offset7 may be NULL (identity). */
int optik_hip_eval_batch(const optik_hip_chain *chain, const optik_solver_config *cfg,
                         const double *target7, const double *ee_offset7, const double *d_q,
                         int64_t B, double *d_f, double *d_g, void *stream);

/* forward_kinematics + joint_jacobian (kinematics.rs:123-196) for B
 * configurations: d_pose [7][B] (EE pose), d_jac [6n][B] column-major 6 x n per
 * configuration (may be NULL). */
int optik_hip_fk_batch(const optik_hip_chain *chain, const double *ee_offset7, const double *d_q,
                       int64_t B, double *d_pose, double *d_jac, void *stream);

/* Restart seeds: ChaCha8Rng::seed_from_u64(42), set_stream(i), one uniform draw
 * per joint (lib.rs:358-370, 86-91) for i = first .. first+count-1 -> d_q [n][count]. */
int optik_hip_seed_batch(const optik_hip_chain *chain, uint64_t first, int64_t count, double *d_q,
                         void *stream);

/* flags of optik_hip_ik_batch */
#define OPTIK_HIP_IK_EARLY_EXIT 1u /* Speed: abandon restarts above a known success (lib.rs:382-384) */
/* With EARLY_EXIT: abandon every other restart of a target as soon as ANY restart of it has
 * succeeded -- the reference's behaviour with more than one rayon thread (find_any, lib.rs:409-412;
 * README.md:17, 96: "non-deterministic").  The winner is then the lowest index among the restarts
 * that happened to finish successfully: a valid solution -- its x / f are what that restart
 * returns on its own, bit for bit -- but which one depends on timing.  Without it (the default)
 * only restarts ABOVE a success are abandoned and the answer is the reference's 1-thread one. */
#define OPTIK_HIP_IK_FIND_ANY 2u
/* optik_hip_ik_batch / _ik_host with T > 1 targets: hand the (target, restart) work items out
 * restart-major (every target's restart 0, then every target's restart 1, ...) instead of
 * target-major; with EARLY_EXIT about eight restarts per target are kept in flight and most
 * higher indices are never started.  Scheduling only: per-restart results and winners are the
 * same (an abandoned restart's status is FORCED_STOP either way). */
#define OPTIK_HIP_IK_RESTART_MAJOR 4u
/* Outputs of optik_hip_ik_batch; any pointer may be NULL to skip that output.
 * R = restart_end - restart_begin. */
typedef struct optik_hip_ik_outputs {
    /* per restart, struct-of-arrays over T*R columns, column = t*R + (i - restart_begin) */
    double *d_x;        /* [n][T*R] returned point (NLopt's best-so-far x)   */
    double *d_f;        /* [T*R]    returned objective value                  */
    int32_t *d_status;  /* [T*R]    OPTIK_RES_*                               */
    int32_t *d_evals;   /* [T*R]    objective evaluations (NLopt's count)     */
    /* per target: the selection of lib.rs:397-413 */
    double *d_win_x;      /* [T][n]                                           */
    double *d_win_f;      /* [T]                                              */
    uint64_t *d_win_idx;  /* [T] winning restart index, UINT64_MAX if none    */
    double *d_win_key;    /* [T] Quality: ||x - x0||_2 ; Speed: (double)index */
} optik_hip_ik_outputs;

/* The hot path: for each of T targets run restarts restart_begin..restart_end-1
 * (index 0 = the caller's seed x0, index i > 0 = ChaCha8 stream i), classify
 * (lib.rs:376-379) and select (Speed: lowest successful index = the reference's
 * 1-thread order; Quality: min ||x - x0||_2, ties to the lower index).
 * d_targets [T][7], d_x0 [T][n].  deadline_s > 0 abandons restarts still running
 * that many seconds after the kernel starts (max_time, lib.rs:260-264, 308). */
int optik_hip_ik_batch(optik_hip_chain *chain, const optik_solver_config *cfg,
                       const double *d_targets, const double *d_x0, int32_t T,
                       const double *ee_offset7, uint64_t restart_begin, uint64_t restart_end,
                       uint32_t flags, double deadline_s, const optik_hip_ik_outputs *out,
                       void *stream);

/* Tuning options of the kernel layer (diagnostics: tests and tools; the defaults are what the product runs with).
 * Each option's default comes from the environment variable named with it, read ONCE when the library first needs
 * an option; afterwards only this call changes it.  Not synchronised with calls in flight.
 *   solve_kernel        OPTIK_SOLVE_KERNEL = quad | lane64 | general   0 auto (by launch size), 1 quad solver
 *                                                                       (ik_quad.hpp), 2 lane-per-restart form
 *                                                                       (ik_lane64.hpp), 3 general solver (ik_wide.hpp)
 *   wide_form           OPTIK_WIDE_FORM = lds | hbm   form of the general solver (0 lds, 1 hbm)
 *   range_rule          OPTIK_RANDOM_RANGE_RULE = new_inclusive   OPTIK_HIP_RANGE_* of chains created afterwards
 *   stop_x_legacy       (none)                   1: nlopt_stop_x of NLopt 2.5 (no zero-step rule)
 * The host layer (optik.h) reads OPTIK_HOST_THREADS and OPTIK_DEVICES; nothing else in the library reads the
 * environment.  Returns 0, or OPTIK_HIP_EINVAL for an unknown name; optik_hip_get_option returns -1 for one. */
int optik_hip_set_option(const char *name, long long value);
long long optik_hip_get_option(const char *name);

/* Host-buffer convenience over optik_hip_ik_batch (what Robot::ik calls): copies
 * targets/x0 in, runs, synchronises, copies the per-target winners out.
 * win_x [T][n], win_f [T], win_idx [T] (UINT64_MAX = no solution), win_key [T]
 * (the selection key; any of them may be NULL).  Calls on one chain are serialised.
 * optik_hip_ik_batch itself is stream-ordered and keeps its launch workspace in the
 * chain handle: use one stream per chain handle. */
int optik_hip_ik_host(optik_hip_chain *chain, const optik_solver_config *cfg,
                      const double *targets, const double *x0, int32_t T, const double *ee_offset7,
                      uint64_t restart_begin, uint64_t restart_end, uint32_t flags,
                      double deadline_s, double *win_x, double *win_f, uint64_t *win_idx,
                      double *win_key);

/* Test hook: elementary functions the kernels use, evaluated on the device.
 * op 0: a/b, 1: sqrt(a), 2: sin(a), 3: cos(a), 4: atan2(a, b) for a > 0, b >= 0. */
int optik_hip_probe(int32_t op, const double *a, const double *b, int64_t count, double *out);
/* Test hook: the math.rs functions one at a time on the device (the reference pins them with golden
 * vectors, crates/optik/tests/test_math.rs:14-61).  poses7 = count x {t[3], quat[i,j,k,w]}; row-major
 * results, `count` x: op 0 so3::log (3), 1 so3::right_jacobian(so3::log(q)) (9), 2 se3::log (6: linear,
 * angular), 3 se3::right_jacobian (36). */
int optik_hip_probe_math(int32_t op, const double *poses7, int64_t count, double *out);

/* Last launch geometry / timing of optik_hip_ik_batch on this chain (for bench.py). */
typedef struct optik_hip_launch_info {
    int32_t grid, block, lds_bytes, tiles;
    float kernel_ms; /* HIP-event time of the solve kernel when timing is enabled */
} optik_hip_launch_info;
/* Enabling timing records a HIP event pair around every solve-kernel launch on the
 * launch stream (and resets the recorded set); optik_hip_timing_mean synchronises
 * on them and returns the mean kernel duration over the launches recorded since
 * (at most 256 are kept). */
void optik_hip_set_timing(optik_hip_chain *chain, int32_t enabled);
int optik_hip_last_launch(const optik_hip_chain *chain, optik_hip_launch_info *info);
int optik_hip_timing_mean(optik_hip_chain *chain, double *mean_ms, int32_t *count);
/* Diagnostic builds (-DOPTIK_PROFILE) accumulate s_memtime cycles per solver phase:
 * out8 = {refill, eval, update, publish, bfgs, lsq, nnls, trips}; all zero in normal builds. */
int optik_hip_phase_profile(optik_hip_chain *chain, unsigned long long *out8);

#ifdef __cplusplus
}
#endif
#endif
