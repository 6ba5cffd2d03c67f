#!/usr/bin/env python3
"""One job of R restarts (one Panda target, every restart run to the end) on the quad solve kernel and on
the streaming engine: where the engine's ~10 ms floor is amortised (the host API's switch-over size,
robot_host.cpp).  Usage: python tools/kernel_vs_engine_probe.py [robot]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from optik_amd import Robot  # noqa: E402
from optik_amd import _native as nat  # noqa: E402

robot = sys.argv[1] if len(sys.argv) > 1 else "panda"
spec = {"panda": ("panda.urdf", "panda_link0", "panda_link8"), "ur10": ("ur10.urdf", "base_link", "ee_link")}[robot]
rb = Robot.from_urdf_file(os.path.join(ROOT, "optik_amd", "robots", spec[0]), spec[1], spec[2])
hc = rb.hip_chain("cuda:0")
n = rb.num_positions()
rng = np.random.default_rng(0)
lb, ub = (np.array(v) for v in rb.joint_limits())
tgt = hc.fk_batch(torch.tensor(rng.uniform(lb, ub, size=(1, n)).T.copy(), device="cuda:0")).T.contiguous()
x0 = torch.tensor(rng.uniform(lb, ub, size=(1, n)), device="cuda:0")
cfg = nat.make_config("quality")
hc.engine_reserve()
for R in (16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152):
    bufs = hc.alloc_ik_buffers(1, R)
    res = {}
    for path in ("kernel", "engine"):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if path == "kernel":
                hc.ik_batch(cfg, tgt, x0, 0, R, bufs=bufs)
            else:
                hc.engine_submit(cfg, tgt, x0, 0, R, bufs=bufs)
                hc.engine_run()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        res[path] = best
    print(f"R={R:8d}  kernel {res['kernel'] * 1e3:8.2f} ms ({R / res['kernel'] / 1e6:6.2f} M/s)   "
          f"engine {res['engine'] * 1e3:8.2f} ms ({R / res['engine'] / 1e6:6.2f} M/s)")
