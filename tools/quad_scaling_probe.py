#!/usr/bin/env python3
"""Restarts/s of the solve kernel for chains of 2..8 joints (sub-chains of the Panda, UR10, arm8): one
launch of T targets x R restarts each.  With OPTIK_AMD_LIB pointing at build variants this separates
what a chain costs from what the build costs (scratch grows with n in the two-waves-per-SIMD build).
Usage: python tools/quad_scaling_probe.py [T] [R]"""
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

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
for name in ("panda2", "panda3", "panda4", "panda5", "ur10", "panda", "arm8"):
    path, base, ee = conftest.ROBOT_SPECS[name]
    rb = Robot.from_urdf_file(path, base, ee)
    hc = rb.hip_chain("cuda:0")
    n = rb.num_positions()
    rng = np.random.default_rng(0)
    lb, ub = (np.array(v) for v in rb.joint_limits())
    tgt = hc.fk_batch(torch.tensor(rng.uniform(lb, ub, size=(T, n)).T.copy(), device="cuda:0")).T.contiguous()
    x0 = torch.tensor(rng.uniform(lb, ub, size=(T, n)), device="cuda:0")
    cfg = nat.make_config("speed")
    bufs = hc.alloc_ik_buffers(T, R)
    hc.ik_batch(cfg, tgt, x0, 0, R, bufs=bufs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hc.ik_batch(cfg, tgt, x0, 0, R, bufs=bufs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev = float(bufs["evals"].double().mean())
    print(f"{name:8s} n={n}  {T * R / dt / 1e6:7.2f} M restarts/s  {dt * 1e3:7.2f} ms  evals/restart {ev:5.1f}  "
          f"{T * R * ev / dt / 1e6:8.1f} M evals/s")
