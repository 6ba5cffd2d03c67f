#!/usr/bin/env python3
"""Many independent ik() calls in one go -- what a GPU is for:

    python examples/many_targets.py <robot.urdf> <base_link> <ee_link> [targets]

Robot.ik_batch_arrays takes [T, 4, 4] poses and [T, n] seeds and returns (x [T, n], c [T],
found [T]); every target gets exactly the answer Robot.ik would give it alone (MI355X, Panda:
1.1 M ik() calls/s at 4 096 targets, 5 M at 262 144)."""
import sys
import time

import numpy as np

from optik_amd import Robot, SolverConfig


def main():
    urdf, base, ee = sys.argv[1:4]
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
    robot = Robot.from_urdf_file(urdf, base, ee)
    lb, ub = (np.array(v) for v in robot.joint_limits())
    rng = np.random.default_rng(0)
    # reachable targets: forward kinematics of random configurations (a few hundred distinct ones)
    distinct = np.array([robot.fk(rng.uniform(lb, ub)) for _ in range(min(T, 512))])
    targets = distinct[rng.integers(0, len(distinct), size=T)]
    seeds = rng.uniform(lb, ub, size=(T, len(lb)))
    config = SolverConfig(solution_mode="speed", max_time=0.0, max_restarts=256)
    robot.ik_batch_arrays(config, targets, seeds)  # first call: set-up
    t0 = time.perf_counter()
    x, c, found = robot.ik_batch_arrays(config, targets, seeds)
    dt = time.perf_counter() - t0
    worst = float(c[found].max()) if found.any() else float("nan")
    print(f"{int(found.sum())} of {T} solved in {1e3 * dt:.1f} ms ({T / dt:,.0f} ik() calls/s), worst residual {worst:.1e}")


if __name__ == "__main__":
    main()
