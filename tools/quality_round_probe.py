#!/usr/bin/env python3
"""Round 5: a long SolutionMode::Quality Robot.ik() call at 2 / 4 / 6 M restarts -- the host runs it in big rounds (robot_host.cpp:
big_batch), each one launch with its own ~3.5 ms drain; compare a -DOPTIK_QUALITY_BATCH_LOG2=20 build through OPTIK_AMD_LIB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optik_amd import Robot, SolverConfig
r = Robot.from_urdf_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'optik_amd', 'robots', 'panda.urdf'), 'panda_link0', 'panda_link8')
rng = np.random.default_rng(4)
lb, ub = (np.array(v) for v in r.joint_limits())
tgt = np.array(r.fk(rng.uniform(lb, ub))); x0 = rng.uniform(lb, ub).tolist()
for R in (1 << 21, 1 << 22, 3 << 21):
    cfg = SolverConfig(solution_mode='quality', max_time=0.0, max_restarts=R)
    a = r.ik(cfg, tgt, x0, return_index=True)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); b = r.ik(cfg, tgt, x0, return_index=True); ts.append(time.perf_counter() - t0)
    assert a == b
    print(os.environ.get('OPTIK_AMD_LIB', 'product')[-12:], R, f"{min(ts)*1e3:.1f} ms", f"{R/min(ts)/1e6:.2f} M/s", a[2], flush=True)
