#!/usr/bin/env python3
"""examples/example.rs on this implementation: random reachable targets, random seeds, default
SolverConfig (Speed, 0.1 s budget); prints average time per ik() call and the success rate."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from optik_amd import Robot, SolverConfig  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
robot = Robot.from_urdf_file(os.path.join(ROOT, "optik_amd", "robots", "panda.urdf"), "panda_link0", "panda_link8")
lb, ub = (np.array(v) for v in robot.joint_limits())
rng = np.random.default_rng(42)
cfg = SolverConfig()
if len(sys.argv) > 2:
    robot.set_parallelism(int(sys.argv[2]))  # > 1: Speed stops at the first success of any restart (find_any)
robot.ik(cfg, np.array(robot.fk(rng.uniform(lb, ub))), rng.uniform(lb, ub).tolist())  # warm-up
tot, ok = 0.0, 0
for _ in range(N):
    x0 = rng.uniform(lb, ub).tolist()
    tgt = np.array(robot.fk(rng.uniform(lb, ub)))
    t0 = time.perf_counter()
    sol = robot.ik(cfg, tgt, x0)
    dt = time.perf_counter() - t0
    if sol is not None:
        tot += dt
        ok += 1
print(f"Average time: {1e6 * tot / max(ok, 1):.0f}us   Success rate: {100.0 * ok / N:.1f}%   ({N} calls, Panda, default config, "
      f"parallelism {sys.argv[2] if len(sys.argv) > 2 else 'unset'})")
