#!/usr/bin/env python3
"""Host-scheduling fuzz: Robot.ik_batch_arrays under random shapes (targets, restart budgets, mode,
share of unreachable targets, robot) must return, for every sampled target, exactly what Robot.ik
returns for it alone (set_parallelism(1): the deterministic Speed rule) -- whatever rounds, kernels
and pool sizes the batch went through (robot_host.cpp:ik_batch_on_device / optik_robot_ik_ex).
Usage: python tools/batch_fuzz.py [rounds] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optik_amd import Robot, SolverConfig  # noqa: E402

ROBOTS = os.path.join(ROOT, "optik_amd", "robots")
SPECS = {"panda": ("panda.urdf", "panda_link0", "panda_link8"), "ur10": ("ur10.urdf", "base_link", "ee_link"),
         "panda5": ("panda.urdf", "panda_link0", "panda_link5")}


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    robots = {}
    for k, (f, b, e) in SPECS.items():
        robots[k] = Robot.from_urdf_file(os.path.join(ROBOTS, f), b, e)
        robots[k].set_parallelism(1)
    far = np.eye(4)
    far[:3, 3] = 50.0
    bad = 0
    for it in range(rounds):
        name = str(rng.choice(list(SPECS)))
        r = robots[name]
        lb, ub = (np.array(v) for v in r.joint_limits())
        n = len(lb)
        T = int(rng.choice([1, 3, 17, 100, 700, 3000, 70000]))
        mode = str(rng.choice(["speed", "quality"]))
        R = int(rng.choice([1, 7, 100, 257, 1500, 9000]))
        if mode == "quality" and T * R > 3_000_000:
            R = max(1, 3_000_000 // T)
        n_far = int(rng.choice([0, 0, 1, 3])) if T > 3 else 0
        if mode == "speed" and n_far * R > 30000:
            n_far = 1 if R <= 30000 else 0
        distinct = np.array([r.fk(rng.uniform(lb, ub)) for _ in range(min(T, 64))])
        targets = distinct[rng.integers(0, len(distinct), size=T)]
        far_at = rng.choice(T, size=n_far, replace=False) if n_far else np.array([], dtype=int)
        targets[far_at] = far
        x0s = rng.uniform(lb, ub, size=(T, n))
        cfg = SolverConfig(solution_mode=mode, max_time=0.0, max_restarts=R)
        x, f, ok = r.ik_batch_arrays(cfg, targets, x0s)
        sample = list(rng.choice(T, size=min(T, 6), replace=False)) + [int(t) for t in far_at[:1]]
        good = True
        for t in sample:
            single = r.ik(cfg, targets[t], x0s[t])
            if (single is None) != (not ok[t]):
                good = False
            elif single is not None and (single[0] != x[t].tolist() or single[1] != f[t]):
                good = False
        print(f"round {it} {name} {mode} T={T} R={R} unreachable={n_far} solved={int(ok.sum())} -> {'ok' if good else 'MISMATCH'}",
              flush=True)
        bad += 0 if good else 1
    print("fuzz ok" if bad == 0 else f"fuzz FAILED: {bad} rounds")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
