#!/usr/bin/env python3
"""Differential IK: joint velocities that realise as much of a Cartesian twist as the velocity
limits allow (Robot.diff_ik, the call of the reference's examples/example_diff_ik.py):

    python examples/diff_ik.py <robot.urdf> <base_link> <ee_link>"""
import sys

import numpy as np

from optik_amd import Robot


def main():
    urdf, base, ee = sys.argv[1:4]
    robot = Robot.from_urdf_file(urdf, base, ee)
    lb, ub = (np.array(v) for v in robot.joint_limits())
    rng = np.random.default_rng(1)
    q = rng.uniform(lb, ub)
    twist = np.array([0.1, 0.0, 0.05, 0.0, 0.2, 0.0])  # world-frame [v; w] of the end effector
    v_max = np.ones(len(q))
    out = robot.diff_ik(q, twist, v_max)
    if out is None:
        print("no solution")
        return
    alpha, v = out
    fk = np.array(robot.fk(q))
    J = np.array(robot.joint_jacobian(q))  # body frame
    R = fk[:3, :3]
    achieved = np.concatenate([R @ (J[:3] @ v), R @ (J[3:] @ v)])
    print(f"alpha = {alpha:.4f}; |J v - alpha V| = {np.linalg.norm(achieved - alpha * twist):.2e}; max |v| = {np.abs(v).max():.3f}")


if __name__ == "__main__":
    main()
