#!/usr/bin/env python3
"""Wave-trips and direction-search passes of the quad kernel on the bench workload (diagnostic;
needs a -DOPTIK_PROFILE build: OPTIK_PROF_LIB=<so>)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optik_amd import _native as nat
nat.LIB_PATH = os.path.abspath(os.environ["OPTIK_PROF_LIB"])
import numpy as np
import torch
from bench import load_chain

dev = torch.device("cuda", 0)
robot = load_chain("panda")
hc = robot.hip_chain(dev)
n = robot.num_positions()
rng = np.random.default_rng(0)
lb, ub = (np.array(v) for v in robot.joint_limits())
K, R = 20, 65536
q = rng.uniform(lb, ub, size=(K, n))
x0 = torch.tensor(rng.uniform(lb, ub, size=(K, n)), device=dev)
targets = hc.fk_batch(torch.tensor(q.T.copy(), device=dev)).T.contiguous()
cfg = nat.make_config(solution_mode="speed", tol_f=1e-6)
bufs = hc.alloc_ik_buffers(K, R)
hc.ik_batch(cfg, targets, x0, 0, R, bufs=bufs, per_restart=True)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
nat.lib().optik_hip_phase_profile.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
nat.check(nat.lib().optik_hip_phase_profile(hc._h, out))
v = list(out)
print(f"wave-trips {v[7]}, direction passes {v[2]} ({v[2] / max(v[7], 1):.3f} per trip), evals/restart {float(bufs['evals'].double().mean()):.1f}")
tot = v[0] + v[1] + v[3] + v[4] + v[5]
for name, c in zip(["refill", "eval", "-", "publish", "bookkeeping+bfgs", "direction", "  nnls"], v[:7]):
    if name != "-":
        print(f"  {name:18s} {c / max(v[7], 1):9.0f} cycles/trip {100.0 * c / tot:5.1f} %")
