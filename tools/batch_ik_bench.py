#!/usr/bin/env python3
"""Config 5 of BASELINE.json as the user sees it: T independent ik() calls (reachable random
targets, random seeds, SolutionMode::Speed, up to 256 restarts each) through Robot.ik_batch.
Prints ik() calls/s.  (The CPU oracle's figure for the same workload: tests/perf/cpu_ik_batch_rate.py.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    from optik_amd import Robot, SolverConfig
    robot = Robot.from_urdf_file(os.path.join(ROOT, "optik_amd", "robots", "panda.urdf"), "panda_link0", "panda_link8")
    rng = np.random.default_rng(0)
    lb, ub = (np.array(v) for v in robot.joint_limits())
    hc = robot.hip_chain("cuda:0")
    import torch
    q = rng.uniform(lb, ub, size=(T, 7))
    pose = hc.fk_batch(torch.tensor(q.T.copy(), device="cuda:0")).T.cpu().numpy()  # [T, 7]
    x0s = rng.uniform(lb, ub, size=(T, 7))

    def mat(p):
        i, j, k, w = p[3:]
        Rm = np.array([[w*w+i*i-j*j-k*k, 2*(i*j-w*k), 2*(w*j+i*k)], [2*(w*k+i*j), w*w-i*i+j*j-k*k, 2*(j*k-w*i)],
                       [2*(i*k-w*j), 2*(w*i+j*k), w*w-i*i-j*j+k*k]])
        m = np.eye(4); m[:3, :3] = Rm; m[:3, 3] = p[:3]
        return m
    targets = np.array([mat(p) for p in pose])
    cfg = SolverConfig(solution_mode="speed", max_time=0.0, max_restarts=R)
    robot.ik_batch(cfg, targets, x0s)  # warm-up at full size (the first call allocates the launch workspace)
    dt = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        res = robot.ik_batch(cfg, targets, x0s)
        dt = min(dt, time.perf_counter() - t0)
    solved = sum(r is not None for r in res)
    print(f"GPU  Robot.ik_batch: {T} targets x <= {R} restarts: {dt*1e3:.1f} ms -> {T/dt:,.0f} ik() calls/s, "
          f"{100.0*solved/T:.1f} % solved")
    dt = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        _, _, ok = robot.ik_batch_arrays(cfg, targets, x0s)
        dt = min(dt, time.perf_counter() - t0)
    print(f"GPU  Robot.ik_batch_arrays (numpy in, numpy out): {dt*1e3:.1f} ms -> {T/dt:,.0f} ik() calls/s, "
          f"{100.0*ok.mean():.1f} % solved")


if __name__ == "__main__":
    main()
