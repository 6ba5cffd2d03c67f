/*
 * optik.h -- the reference's C ABI (crates/optik-cpp/src/lib.rs:26-183), exported by
 * liboptik_amd.so so that the reference's C++ wrapper (crates/optik-cpp/src/lib.cpp,
 * include/optik.hpp) and any other binding of `liboptikcpp` can link against the
 * MI355X implementation unchanged.  All compute goes through the HIP kernels of
 * include/optik_hip.h; there is no CPU fallback.
 *
 * Ownership (as in the reference): a `robot*` is released only by optik_robot_free;
 * every returned `double*` is a malloc'ed buffer the caller releases with free()
 * (lib.cpp:72, 87, 100, 115, 129).  Matrices are column-major.  Invalid input
 * aborts the process with the reference's panic message on stderr (a Rust panic
 * across `extern "C"` aborts too); "no solution" is NULL.
 */
#ifndef OPTIK_H
#define OPTIK_H

#include <stdint.h>

#include "optik_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct optik_robot optik_robot;           /* Box<Robot>, optik-cpp/src/lib.rs:27-57 */
typedef optik_solver_config CSolverConfig;        /* optik-cpp/src/lib.rs:10-20 */

/* ---- the 11 symbols of the reference ------------------------------------------ */
optik_robot *optik_robot_from_urdf_file(const char *path, const char *base_link,
                                        const char *ee_link);                 /* lib.rs:27-37 */
optik_robot *optik_robot_from_urdf_str(const char *urdf, const char *base_link,
                                       const char *ee_link);                  /* lib.rs:40-50 */
void optik_robot_free(optik_robot *robot);                                    /* lib.rs:53-57 */
void optik_robot_set_parallelism(optik_robot *robot, unsigned int n);         /* lib.rs:60-65 */
unsigned int optik_robot_num_positions(const optik_robot *robot);             /* lib.rs:68-73 */
const double *optik_robot_joint_limits(const optik_robot *robot);             /* lib.rs:76-88: [lb.., ub..] */
const double *optik_robot_joint_jacobian(const optik_robot *robot, const double *x); /* lib.rs:91-104: 6 x n */
const double *optik_robot_fk(const optik_robot *robot, const double *x);      /* lib.rs:107-116: 4 x 4 */
const double *optik_robot_random_configuration(const optik_robot *robot);     /* lib.rs:119-125 */
const double *optik_robot_ik(const optik_robot *robot, const CSolverConfig *config,
                             const double *target, const double *x0);         /* lib.rs:128-162 */
/* Differential IK (optik-cpp lib.rs:165-183; Robot::diff_ik, lib.rs:123-239): the joint
 * velocities v_n[n] realising alpha * V_WE for the largest feasible 0 <= alpha <= 1 under
 * |v_i| <= v_max_i.  The LP (<= 9 unknowns) is solved exactly on the host; FK and the Jacobian
 * come from the HIP kernels.  NULL = no solution.  Chains of more than 8 joint positions are
 * refused (the reference's own diff_ik only runs for n = 6: lib.rs:196-197). */
const double *optik_robot_diff_ik(const optik_robot *robot, const double *x0, const double *V_WE,
                                  const double *v_max);

/* ---- extensions used by the Python front end and bench.py --------------------- */
/* Error-returning constructors / solver: 0 on success, negative on failure with the
 * message in optik_robot_last_error() (what a pyo3-style binding turns into an
 * exception instead of aborting). */
int optik_robot_try_from_urdf_str(const char *urdf, const char *base_link, const char *ee_link,
                                  optik_robot **out);
const char *optik_robot_last_error(void);
/* Robot::ik with the full signature of lib.rs:241-247: ee_offset (4x4 col-major, NULL
 * = identity), returns x and the residual c.  rc 0 = solution, 1 = none, < 0 = error. */
int optik_robot_ik_ex(const optik_robot *robot, const CSolverConfig *config, const double *target16,
                      const double *x0, const double *ee_offset16, double *x_out, double *f_out,
                      uint64_t *winner_out);
/* How a 4x4 target becomes the (translation, unit quaternion) pose Robot::ik takes.  The reference has two readings:
 * its C path (optik-cpp/src/lib.rs:137-144) runs nalgebra's ITERATIVE UnitQuaternion::from_matrix on the 3x3 block,
 * its Python path (optik-py/src/lib.rs:8-15) nalgebra's try_convert = the closed-form from_rotation_matrix.  Both give
 * the same rotation for a rotation matrix, last bits apart -- and last bits decide which restart wins (DESIGN.md
 * section 2).  optik_robot_ik (the reference's C symbol) uses the iterative form, as liboptikcpp does;
 * optik_robot_ik_ex (what the Python front end calls; ee_offset only exists there) the closed form;
 * optik_robot_ik_pose takes the choice as a flag: OPTIK_POSE_FROM_MATRIX = the iterative form.
 * optik_pose_from_matrix: the conversion alone (host code, no GPU): m16 column-major -> pose7 [t, qi, qj, qk, qw]. */
#define OPTIK_POSE_FROM_MATRIX 4u
int optik_robot_ik_pose(const optik_robot *robot, const CSolverConfig *config, const double *target16,
                        uint32_t flags, const double *x0, const double *ee_offset16, double *x_out,
                        double *f_out, uint64_t *winner_out);
int optik_pose_from_matrix(const double *m16, uint32_t flags, double *pose7);
/* T independent ik() calls with one SolverConfig: targets16 [T][16] (4x4 col-major each),
 * x0 [T][n] -> x_out [T][n], f_out [T], found_out [T] (0/1).  Every target gets Robot::ik's
 * semantics (config 5 of BASELINE.json: many targets x a few hundred restarts).  Scheduling
 * (robot_host.cpp:ik_batch_on_device): every round is ONE launch of optik_hip_ik_batch over the
 * targets still unsolved.  Speed: a first round of 128 restart indices per target, restart-major
 * with early exit (64 when the round would exceed ~4 M work items, down to 8 for batches of more
 * than 65 536 targets); what it leaves unsolved runs in rounds four times as long each.  Quality:
 * every restart of every target, as many indices per round as ~4 M work items allow, on the
 * lane-per-restart form.
 * rc 0 = ran, < 0 = error. */
int optik_robot_ik_batch_ex(const optik_robot *robot, const CSolverConfig *config, int32_t T,
                            const double *targets16, const double *x0, const double *ee_offset16,
                            double *x_out, double *f_out, int32_t *found_out);
/* The same with the targets as the Python binding receives them: OPTIK_BATCH_ROW_MAJOR = each
 * 4x4 is row-major (optik-py/src/lib.rs:8-15 reads nested rows), OPTIK_BATCH_VALIDATE_POSES =
 * apply parse_pose's isometry test to every target (bottom row exactly 0 0 0 1, R'R = I within
 * 100 * DBL_EPSILON per entry, det R > 0) and return -3 with "invalid target transform
 * specified" before any GPU work if one fails; OPTIK_POSE_FROM_MATRIX = convert every target with the iterative
 * UnitQuaternion::from_matrix (the C path's reading; optik_robot_ik_batch_ex passes it, the Python front end does
 * not). */
#define OPTIK_BATCH_ROW_MAJOR 1u
#define OPTIK_BATCH_VALIDATE_POSES 2u
int optik_robot_ik_batch_poses(const optik_robot *robot, const CSolverConfig *config, int32_t T,
                               const double *targets16, uint32_t flags, const double *x0,
                               const double *ee_offset16, double *x_out, double *f_out,
                               int32_t *found_out);
/* GPUs of this node the robot spreads its work over (restarts shard trivially: lib.rs:297-300
 * hands the same index range to rayon workers).  optik_robot_ik / _ik_ex: after the
 * latency-sized first launch every round's restart range is cut into one contiguous part per
 * GPU and the host keeps the minimum of the per-GPU (key, index) records -- the 16-byte
 * reduction SURVEY 8e calls for; _ik_batch_ex: the targets are cut into one contiguous part
 * per GPU, no reduction.  Results do not depend on the device list.  Must be called before
 * the robot's first GPU call; count = 0 restores the default (the HIP device current at first
 * use); the environment variable OPTIK_DEVICES ("all", a count, or "0,1,...") sets the list
 * for robots created after it.  A device may be listed more than once. */
int optik_robot_set_devices(optik_robot *robot, const int32_t *device_ids, int32_t count);
int32_t optik_robot_num_devices(const optik_robot *robot);
/* Over how many of the robot's devices the last _ik / _ik_batch call was actually cut (its widest round): 1 for a
 * call that ended with its latency-sized first launch or whose range was too short to be worth cutting. */
int32_t optik_robot_last_parts(const optik_robot *robot);
/* Robot::diff_ik with its full signature: ee_offset and alpha.  rc 0 = solved, 1 = none. */
int optik_robot_diff_ik_ex(const optik_robot *robot, const double *x0, const double *V_WE6,
                           const double *v_max, const double *ee_offset16, double *alpha_out,
                           double *v_out);
int optik_robot_fk_ex(const optik_robot *robot, const double *x, const double *ee_offset16,
                      double *pose16_out);
int optik_robot_joint_jacobian_ex(const optik_robot *robot, const double *x,
                                  const double *ee_offset16, double *jac6n_out);
/* Flat chain table (what KinematicChain::from_urdf produced): n_joints poses, axes, types.
 * optik_robot_chain_tables: *n_joints is IN/OUT -- with non-NULL buffers it holds their capacity in joints on
 * entry (x 7 / x 3 / x 1 elements each; OPTIK_MAX_JOINTS covers every chain of at most 8 joint positions) and the
 * chain's joint count on return; fewer than the chain has: error, nothing is written.  With NULL buffers it just
 * reports the count.  optik_robot_chain_tables_n: the same with the capacity as its own argument. */
#define OPTIK_MAX_JOINTS 9
int optik_robot_chain_tables(const optik_robot *robot, int32_t *n_joints, double *origins7,
                             double *axes3, int32_t *types);
int optik_robot_chain_tables_n(const optik_robot *robot, int32_t capacity, int32_t *n_joints,
                               double *origins7, double *axes3, int32_t *types);
/* The device-side chain of this robot on the current HIP device (created on first
 * use; owned by the robot). */
optik_hip_chain *optik_robot_hip_chain(const optik_robot *robot);

#ifdef __cplusplus
}
#endif
#endif
