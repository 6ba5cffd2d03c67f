#!/usr/bin/env python3
"""How the restarts of the bench workload end: histogram of NLopt result codes and evaluation counts
(diagnostic; one launch of the solve kernel, Panda, 65 536 restarts of a few targets)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from optik_amd import _native as nat
from bench import load_chain

dev = torch.device("cuda", 0)
robot = load_chain(sys.argv[1] if len(sys.argv) > 1 else "panda")
hc = robot.hip_chain(dev)
n = robot.num_positions()
rng = np.random.default_rng(0)
lb, ub = (np.array(v) for v in robot.joint_limits())
K, R = 4, 65536
q = rng.uniform(lb, ub, size=(K, n))
x0 = torch.tensor(rng.uniform(lb, ub, size=(K, n)), device=dev)
targets = hc.fk_batch(torch.tensor(q.T.copy(), device=dev)).T.contiguous()
cfg = nat.make_config(solution_mode="speed", tol_f=1e-6)
b = hc.ik_batch(cfg, targets, x0, 0, R, per_restart=True)
torch.cuda.synchronize()
st = b["status"].cpu().numpy()
ev = b["evals"].cpu().numpy()
names = {v: k for k, v in vars(nat).items() if k.startswith("RES_") and isinstance(v, int)}
for code in sorted(set(st.tolist())):
    m = st == code
    print(f"status {code:3d} {names.get(code, '?'):24s} {m.mean() * 100:6.2f} %   evals mean {ev[m].mean():6.1f}  p50 {np.percentile(ev[m], 50):5.0f}  "
          f"p99 {np.percentile(ev[m], 99):5.0f}  max {ev[m].max():5d}")
print(f"all: evals mean {ev.mean():.1f} p99 {np.percentile(ev, 99):.0f} p99.9 {np.percentile(ev, 99.9):.0f} max {ev.max()}")
