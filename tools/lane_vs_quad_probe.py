#!/usr/bin/env python3
"""Launch time of one optik_hip_ik_batch call of R restarts (one target) on the quad solver and on the lane-per-restart
solver (option solve_kernel = quad / lane64): where the crossover between the two lies.
Usage: python tools/lane_vs_quad_probe.py [robot]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from optik_amd import Robot  # noqa: E402
from optik_amd import _native as nat  # noqa: E402
import conftest  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "panda"
path, base, ee = conftest.ROBOT_SPECS[name]
rb = Robot.from_urdf_file(path, base, ee)
hc = rb.hip_chain("cuda:0")
n = rb.num_positions()
rng = np.random.default_rng(0)
lb, ub = (np.array(v) for v in rb.joint_limits())
tgt = hc.fk_batch(torch.tensor(rng.uniform(lb, ub, size=(1, n)).T.copy(), device="cuda:0")).T.contiguous()
x0 = torch.tensor(rng.uniform(lb, ub, size=(1, n)), device="cuda:0")
cfg = nat.make_config("speed")
for R in (1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 1048576):
    row = []
    for kern in ("quad", "lane64"):
        nat.set_option("solve_kernel", kern)
        bufs = hc.alloc_ik_buffers(1, R)
        hc.ik_batch(cfg, tgt, x0, 0, R, bufs=bufs)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            hc.ik_batch(cfg, tgt, x0, 0, R, bufs=bufs)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        row.append(best)
    print(f"{name} R={R:8d}  quad {row[0] * 1e3:8.3f} ms ({R / row[0] / 1e6:6.2f} M/s)   lane64 {row[1] * 1e3:8.3f} ms "
          f"({R / row[1] / 1e6:6.2f} M/s)   lane64 / quad time {row[1] / row[0]:.2f}")
