#!/usr/bin/env python3
"""One ik() call at a time, as a motion planner's inner loop issues them (the workload of the
reference's examples/example.py / example.rs, same command line):

    python examples/single_ik.py <robot.urdf> <base_link> <ee_link> [calls]

Targets are poses of random configurations (reachable by construction), seeds are random,
SolverConfig is the default (Speed, 0.1 s budget).  With the reference's default thread pool a
Speed call returns the first solution any restart finds; call robot.set_parallelism(1) for the
deterministic lowest-restart answer (about 1 ms instead of 0.4 ms on MI355X, Panda)."""
import sys
import time

import numpy as np

from optik_amd import Robot, SolverConfig


def main():
    urdf, base, ee = sys.argv[1:4]
    calls = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
    robot = Robot.from_urdf_file(urdf, base, ee)
    config = SolverConfig()
    lb, ub = (np.array(v) for v in robot.joint_limits())
    rng = np.random.default_rng()
    robot.ik(config, np.array(robot.fk(rng.uniform(lb, ub))), rng.uniform(lb, ub))  # first call: set-up
    spent, solved = 0.0, 0
    for _ in range(calls):
        seed = rng.uniform(lb, ub)
        target = np.array(robot.fk(rng.uniform(lb, ub)))
        t0 = time.perf_counter()
        sol = robot.ik(config, target, seed)
        dt = time.perf_counter() - t0
        if sol is not None:
            spent += dt
            solved += 1
    print(f"{solved} of {calls} solved, {1e6 * spent / max(solved, 1):.0f} us per solved call")


if __name__ == "__main__":
    main()
