/* examples/c_api.c -- the reference's C ABI (crates/optik-cpp/src/lib.rs:26-183, declared in
 * include/optik.h) from plain C: load a URDF, solve one pose, check it with forward kinematics.
 * Every returned double* is a malloc'ed buffer the caller frees (the reference's ownership rule).
 *
 *   gcc -std=c11 -Iinclude examples/c_api.c -Loptik_amd/csrc -loptik_amd -Wl,-rpath,$PWD/optik_amd/csrc -lm -o c_api
 *   ./c_api optik_amd/robots/panda.urdf panda_link0 panda_link8
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "optik.h"

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <urdf> <base link> <ee link>\n", argv[0]);
        return 2;
    }
    optik_robot *robot = optik_robot_from_urdf_file(argv[1], argv[2], argv[3]);
    const unsigned n = optik_robot_num_positions(robot);
    const double *lim = optik_robot_joint_limits(robot); /* [lb.., ub..] */
    double *q = malloc(sizeof(double) * n), *x0 = malloc(sizeof(double) * n);
    for (unsigned i = 0; i < n; ++i) {
        q[i] = lim[i] + 0.3 * (lim[n + i] - lim[i]);   /* a reachable pose: FK of a point inside the limits */
        x0[i] = lim[i] + 0.6 * (lim[n + i] - lim[i]);  /* the seed */
    }
    const double *target = optik_robot_fk(robot, q); /* 4 x 4, column-major */

    CSolverConfig cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.solution_mode = 2; /* Speed */
    cfg.max_time = 0.1;
    cfg.max_restarts = 0;  /* unbounded */
    cfg.tol_f = 1e-12;
    cfg.tol_df = -1.0;
    cfg.tol_dx = -1.0;
    for (int k = 0; k < 3; ++k) cfg.linear_weight[k] = cfg.angular_weight[k] = 1.0;

    const double *x = optik_robot_ik(robot, &cfg, target, x0);
    if (!x) {
        printf("no solution\n");
        return 1;
    }
    const double *reached = optik_robot_fk(robot, x);
    double err = 0.0;
    for (int k = 0; k < 16; ++k) err = fmax(err, fabs(reached[k] - target[k]));
    printf("solved %u joints, max |FK(x) - target| = %.3e\n", n, err);
    free((void *)reached);
    free((void *)x);
    free((void *)target);
    free((void *)lim);
    free(q);
    free(x0);
    optik_robot_free(robot);
    return err < 1e-5 ? 0 : 1;
}
