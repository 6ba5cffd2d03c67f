#!/usr/bin/env python3
"""Soak of the host API: a few thousand mixed calls (single ik() short and long, Speed / Quality
batches of several sizes, fk) on one robot -- answers identical from cycle to cycle, device memory
flat (job buffers, staging blocks and launch workspaces are allocated once or freed per call).
Usage: python tools/robot_soak.py [cycles]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optik_amd import Robot, SolverConfig  # noqa: E402


def main():
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    r = Robot.from_urdf_file(os.path.join(ROOT, "optik_amd", "robots", "panda.urdf"), "panda_link0", "panda_link8")
    r.set_parallelism(1)
    rng = np.random.default_rng(0)
    lb, ub = (np.array(v) for v in r.joint_limits())
    targets = np.array([r.fk(rng.uniform(lb, ub)) for _ in range(64)])
    x0s = rng.uniform(lb, ub, size=(64, 7))
    big = targets[rng.integers(0, 64, size=50000)]
    big_x0 = rng.uniform(lb, ub, size=(50000, 7))
    far = np.eye(4)
    far[:3, 3] = 50.0
    ref, free0 = None, None
    for c in range(cycles):
        out = []
        for t in range(8):
            out.append(r.ik(SolverConfig(max_time=0.0, max_restarts=2000), targets[t], x0s[t]))
        out.append(r.ik(SolverConfig(solution_mode="quality", max_time=0.0, max_restarts=150_000), targets[0], x0s[0]))
        out.append(r.ik(SolverConfig(max_time=0.0, max_restarts=120_000), far, x0s[1]))
        out.append(r.ik_batch(SolverConfig(max_time=0.0, max_restarts=64), targets, x0s))
        out.append(r.ik_batch(SolverConfig(solution_mode="quality", max_time=0.0, max_restarts=3000), targets[:40], x0s[:40]))
        x, f, ok = r.ik_batch_arrays(SolverConfig(max_time=0.0, max_restarts=64), big, big_x0)
        out.append((x.tobytes(), f.tobytes(), ok.tobytes()))
        out.append([r.fk(x0s[t]) for t in range(16)])
        if ref is None:
            ref = out
        assert out == ref, f"cycle {c}: answers changed"
        free, _ = torch.cuda.mem_get_info()
        if c == 1:
            free0 = free  # (cycle 0 allocates the pools and staging blocks)
        if c % 5 == 0:
            print(f"cycle {c}: free {free / 2**30:.3f} GiB", flush=True)
    assert free0 is None or abs(free - free0) < 64 * 2**20, (free0, free)
    print("soak ok")


if __name__ == "__main__":
    main()
