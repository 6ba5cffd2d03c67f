/*
 * optik_oracle.h -- CPU oracle for the random-restart IK hot path of kylc/optik.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a plain-C, f64,
 * one-restart-at-a-time restatement of the reference algorithm.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the
 * product path (optik_amd/csrc) never links, imports or executes anything here.
 *
 * Parity status (see DESIGN.md section "Oracle"):
 *   - math.rs / kinematics.rs / objective.rs restatement: PINNED against the
 *     reference's own golden vectors (tests/golden/reference/ JSON files, copied data
 *     files of /root/reference/crates/optik/tests/data).
 *   - NLopt SLSQP (nlopt 0.8.1 @ kylc/rust-nlopt 8e731e3, not vendored in the
 *     reference tree), rand_chacha 0.9.0 / rand 0.9.2 (not vendored): restated
 *     from the published algorithms.  PARITY UNPINNED against the reference (the
 *     generated fixtures under tests/golden/generated/ and the Rust program of
 *     INTEGRATION.md section 5 are what a maintainer with cargo runs to pin it); the
 *     SLSQP restatement is cross-checked iterate-by-iterate against scipy
 *     1.15.3's Fortran build of the same Kraft code, the ChaCha core against the
 *     published 8-round known-answer and libsodium at 20 rounds.
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef OPTIK_ORACLE_H
#define OPTIK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OK_MAX_JOINTS 17 /* chain joints incl. trailing fixed */
#define OK_MAX_DOF 16

/* Pose = translation + unit quaternion stored [i, j, k, w]  (SURVEY app. A). */
typedef struct {
    double t[3];
    double q[4];
} ok_pose;

/* kinematics.rs:219-241 -- Joint / JointType, flattened. */
enum { OK_JOINT_FIXED = 0, OK_JOINT_REVOLUTE = 1, OK_JOINT_PRISMATIC = 2 };

typedef struct {
    int32_t n_joints;                 /* J: chain.joints.len() */
    int32_t n_pos;                    /* n: num_positions(), kinematics.rs:108-110 */
    int32_t type[OK_MAX_JOINTS];      /* OK_JOINT_* */
    int32_t _pad;
    ok_pose origin[OK_MAX_JOINTS];    /* Joint::origin (fixed joints already folded) */
    double axis[OK_MAX_JOINTS][3];    /* unit axis (revolute / prismatic) */
    double lb[OK_MAX_DOF];            /* joint_limits(), lib.rs:78-84 */
    double ub[OK_MAX_DOF];
} ok_chain;

/* config.rs:22-50 / optik-cpp/src/lib.rs:10-20 (CSolverConfig, 96 bytes LP64). */
typedef struct {
    int32_t solution_mode; /* 1 = Quality, 2 = Speed (config.rs:3-8) */
    int32_t _pad;
    double max_time;
    uint64_t max_restarts;
    double tol_f;
    double tol_df;
    double tol_dx;
    double linear_weight[3];
    double angular_weight[3];
} ok_config;

/* Per-restart result codes: NLopt's nlopt_result values where one applies. */
enum {
    OK_RES_FAILURE = -1,          /* NLOPT_FAILURE (LSQ modes 3,4,9) */
    OK_RES_INVALID_ARGS = -2,
    OK_RES_ROUNDOFF_LIMITED = -4, /* NLOPT_ROUNDOFF_LIMITED */
    OK_RES_FORCED_STOP = -5,      /* NLOPT_FORCED_STOP (timeout / early exit) */
    OK_RES_ITER_CAP = -100,       /* oracle-only safety cap, never hit in tests */
    OK_RES_STOPVAL_REACHED = 2,   /* NLOPT_STOPVAL_REACHED */
    OK_RES_FTOL_REACHED = 3,      /* NLOPT_FTOL_REACHED */
    OK_RES_XTOL_REACHED = 4       /* NLOPT_XTOL_REACHED */
};

typedef struct {
    int32_t result;   /* OK_RES_* (raw optimiser return) */
    int32_t success;  /* lib.rs:376-379 classification */
    int32_t n_evals;  /* objective evaluations (value or value+grad) */
    int32_t n_iters;  /* SLSQP major iterations started */
    double f;         /* minf (best value seen) */
    double x[OK_MAX_DOF]; /* best point */
} ok_restart_result;

/* ---- elementary functions shared (as an operation sequence) with the kernel */
void ok_sincos(double x, double *s, double *c);
double ok_atan2_q1(double y, double x); /* y > 0, x >= 0 */

/* ---- math.rs ---------------------------------------------------------- */
void ok_so3_log(const double q[4], double w[3]);                 /* math.rs:40-63 */
void ok_so3_right_jacobian(const double w[3], double J[9]);      /* math.rs:72-94, col-major */
void ok_se3_log(const ok_pose *X, double e[6]);                  /* math.rs:107-124 */
void ok_se3_right_jacobian(const ok_pose *X, double J[36]);      /* math.rs:191-203, col-major */

/* ---- kinematics.rs ---------------------------------------------------- */
void ok_pose_from_rpy(const double xyz[3], const double rpy[3], ok_pose *out); /* :263-267 */
/* 3x3 rotation block (column-major) -> unit quaternion [i, j, k, w]: iterative != 0 as the reference's C binding
 * (optik-cpp/src/lib.rs:141-142, nalgebra UnitQuaternion::from_matrix), 0 as its Python binding
 * (optik-py/src/lib.rs:8-15, from_rotation_matrix). */
void ok_quat_from_matrix(const double m[9], int iterative, double q[4]);
void ok_fk(const ok_chain *c, const double *q, const ok_pose *ee_offset,
           ok_pose *joint_tfms /* [J] */, ok_pose *ee_tfm);      /* :123-164 */
void ok_joint_jacobian(const ok_chain *c, const ok_pose *joint_tfms,
                       const ok_pose *ee_tfm, double *J6n);      /* :166-196, 6 x n col-major */

/* ---- objective.rs ----------------------------------------------------- */
double ok_objective(const ok_chain *c, const ok_pose *target, const ok_pose *joint_tfms,
                    const ok_pose *ee_tfm, const double w_lin[3], const double w_ang[3]);
void ok_objective_grad(const ok_chain *c, const ok_pose *target, const ok_pose *joint_tfms,
                       const ok_pose *ee_tfm, const double w_lin[3], const double w_ang[3],
                       double *g);
/* value + gradient at q, as the NLopt callback does (lib.rs:305-337). */
double ok_eval(const ok_chain *c, const ok_pose *target, const ok_pose *ee_offset,
               const double w_lin[3], const double w_ang[3], const double *q, double *g_or_null);

/* ---- rand_core / rand_chacha / rand  ([EXT]) ----------- */
void ok_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds,
                     uint32_t out[16]);
void ok_seed_from_u64(uint64_t seed, uint32_t key[8]);           /* rand_core PCG32 expansion */
/* which rand 0.9.2 code path `random_range(lb..=ub)` (lib.rs:89) is restated; process-wide,
 * set before solving.  See the comment at ok_uniform_scale in optik_oracle.c. */
enum { OK_RANGE_SINGLE_INCLUSIVE = 0, /* UniformFloat::sample_single_inclusive: scale = hi - lo (default) */
       OK_RANGE_NEW_INCLUSIVE = 1 };  /* Uniform::new_inclusive(..).sample(..): scale = (hi - lo)/(1 - eps), decreased */
void ok_set_range_rule(int rule);
int ok_get_range_rule(void);
double ok_uniform_scale(double lo, double hi, int rule);
double ok_uniform_inclusive(double lo, double hi, uint64_t bits);/* rand UniformFloat<f64>, current rule */
/* lib.rs:358-370 + 86-91: seed of restart i (i >= 1); restart 0 uses the caller's x0. */
void ok_restart_seed(const ok_chain *c, uint64_t restart_index, double *q0);

/* ---- NLopt SLSQP restatement + lib.rs:301-391 ------------------------- */
/* nlopt_stop_x rule: 1 (default) = NLopt >= 2.6.2, a zero step is x-converged; 0 = NLopt 2.5. */
void ok_set_stop_x_zero(int on);
/* One restart: builds the seed (x0 if index==0), runs SLSQP with NLopt's
 * stopping rules, classifies.  trace (optional, may be NULL) receives one row
 * of n+1 doubles [x..., f] per objective evaluation, up to trace_cap rows. */
void ok_solve_restart(const ok_chain *c, const ok_config *cfg, const ok_pose *target,
                      const ok_pose *ee_offset, const double *x0, uint64_t restart_index,
                      ok_restart_result *out, double *trace, int trace_cap, int *trace_len);

/* lib.rs:241-415 with max_time == 0 semantics on a restart range
 * [restart_begin, restart_end): Speed -> lowest successful index (the 1-thread
 * order, quirk Q6), Quality -> min ||x - x0||_2 (ties: lowest index).
 * Returns 1 and fills winner/x/f when any restart succeeded, else 0.
 * per_restart (optional) receives every restart's result (no early exit is
 * taken when it is non-NULL or when early_exit == 0).  n_threads >= 1: restart
 * indices are handed out dynamically from a shared counter (rayon analogue).
 * early_exit == 2 (Speed): the reference's own multi-thread rule, find_any (lib.rs:409-412): the first success in
 * TIME wins and every other restart stops at its next objective call (lib.rs:308) -- a valid answer, not a fixed
 * one; used by bench.py's CPU baseline only, never by a parity test. */
int ok_ik(const ok_chain *c, const ok_config *cfg, const ok_pose *target,
          const ok_pose *ee_offset, const double *x0, uint64_t restart_begin,
          uint64_t restart_end, int n_threads, int early_exit, uint64_t *winner,
          double *x_out, double *f_out, ok_restart_result *per_restart,
          uint64_t *n_restarts_run);

/* T independent single-threaded ik() calls (restart indices 0 .. n_restarts-1 each, Speed early exit) spread over
 * n_threads threads; found[t] = solved, xs (optional) [T][n].  bench.py's CPU figure for BASELINE config 5. */
int ok_ik_many(const ok_chain *c, const ok_config *cfg, const ok_pose *targets, const ok_pose *ee_offset,
               const double *x0s, int T, uint64_t n_restarts, int n_threads, int32_t *found, double *xs);

/* Persistent worker threads for ok_ik calls with n_threads == n (what rayon's pool is to Robot::ik); 0 stops them. */
int ok_pool_start(int n_threads);
void ok_pool_stop(void);

/* Direction sub-problem of one SLSQP major iteration, exposed for tests:
 * min 1/2 s'LDL's + g's  s.t. lo <= s <= hi   via Kraft's LSQ->LSI->LDP->NNLS.
 * l: packed LDL' (n(n+1)/2), returns the LSQ mode (1 = ok). */
int ok_lsq_direction(int n, const double *l, const double *g, const double *lo, const double *hi,
                     double *s);

#ifdef __cplusplus
}
#endif
#endif
