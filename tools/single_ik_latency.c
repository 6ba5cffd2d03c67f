/* tools/single_ik_latency.c -- the loop of the reference's examples/example.rs through the C ABI (no Python
 * in the timed path): random reachable targets, random seeds, default SolverConfig; average time of
 * optik_robot_ik per solved call.
 *   gcc -O2 -std=c11 -Iinclude tools/single_ik_latency.c -Loptik_amd/csrc -loptik_amd -Wl,-rpath,$PWD/optik_amd/csrc -lm -o /tmp/lat
 *   /tmp/lat optik_amd/robots/panda.urdf panda_link0 panda_link8 [calls] [parallelism, 0 = unset] [gap: 1 = an fk call between two ik calls, 0 = back to back]
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "optik.h"

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const int calls = argc > 4 ? atoi(argv[4]) : 1000;
    optik_robot *robot = optik_robot_from_urdf_file(argv[1], argv[2], argv[3]);
    if (argc > 5 && atoi(argv[5]) > 0) optik_robot_set_parallelism(robot, (unsigned)atoi(argv[5]));  /* 0: leave unset */
    const unsigned n = optik_robot_num_positions(robot);
    const double *lim = optik_robot_joint_limits(robot);
    CSolverConfig cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.solution_mode = 2; cfg.max_time = 0.1; cfg.max_restarts = 0;
    cfg.tol_f = 1e-6; cfg.tol_df = -1.0; cfg.tol_dx = -1.0;
    for (int k = 0; k < 3; ++k) cfg.linear_weight[k] = cfg.angular_weight[k] = 1.0;
    double *q = malloc(sizeof(double) * n), *x0 = malloc(sizeof(double) * n);
    srand(42);
    double total = 0.0;
    int solved = 0;
    /* The targets are formed before the timed loop (the reference's example forms them on the CPU; here
     * optik_robot_fk is a GPU call of its own): the second figure is the loop with nothing between two ik calls --
     * a call that returned on its first success leaves a launch behind, and the next call queues behind it. */
    double *targets = malloc(sizeof(double) * 16 * (size_t)(calls + 1)), *seeds = malloc(sizeof(double) * n * (size_t)(calls + 1));
    for (int c = 0; c <= calls; ++c) {
        for (unsigned i = 0; i < n; ++i) {
            q[i] = lim[i] + (lim[n + i] - lim[i]) * ((double)rand() / RAND_MAX);
            seeds[(size_t)c * n + i] = lim[i] + (lim[n + i] - lim[i]) * ((double)rand() / RAND_MAX);
        }
        const double *target = optik_robot_fk(robot, q);
        memcpy(targets + 16 * (size_t)c, target, sizeof(double) * 16);
        free((void *)target);
    }
    const int gap = argc > 6 ? atoi(argv[6]) : 1;  /* 1: a GPU call (fk) between two ik calls, as in rounds 1-3; 0: back to back */
    for (int c = -1; c < calls; ++c) {  /* (call -1 warms the device up) */
        const double *target = targets + 16 * (size_t)(c + 1);
        if (gap) free((void *)optik_robot_fk(robot, seeds + (size_t)(c + 1) * n));
        const double t0 = now_s();
        const double *x = optik_robot_ik(robot, &cfg, target, seeds + (size_t)(c + 1) * n);
        const double dt = now_s() - t0;
        if (x && c >= 0) { total += dt; solved += 1; }
        free((void *)x);
    }
    printf("Average time: %.0fus   Success rate: %.1f%%   (%d calls through the C ABI, %u joints, parallelism %s, %s)\n",
           1e6 * total / (solved > 0 ? solved : 1), 100.0 * solved / calls, calls, n,
           (argc > 5 && atoi(argv[5]) > 0) ? argv[5] : "unset", gap ? "an fk call between calls" : "back to back");
    free(targets); free(seeds);
    free((void *)lim); free(q); free(x0);
    optik_robot_free(robot);
    return 0;
}
