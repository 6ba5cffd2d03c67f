#!/usr/bin/env python3
"""Wave cycles per part of a trip of the lane-per-restart form (ik_lane64.hpp: LANE_PROF) on the bench workload; needs a
-DOPTIK_PROFILE build of ik_capi.o and ik_lane_kernel.o:
  python tools/build_lib_variant.py prof2 -DOPTIK_PROFILE --only=ik_capi.o,ik_lane_kernel.o
  OPTIK_PROF_LIB=optik_amd/csrc/variants/prof2.so python tools/lane_phase_profile.py"""
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
robot = load_chain(sys.argv[1] if len(sys.argv) > 1 else "panda")
hc = robot.hip_chain(dev)
n = robot.num_positions()
rng = np.random.default_rng(0)
lb, ub = (np.array(v) for v in robot.joint_limits())
K, R = 8, 65536
q = rng.uniform(lb, ub, size=(K, n))
x0 = torch.tensor(rng.uniform(lb, ub, size=(K, n)), device=dev)
targets = hc.fk_batch(torch.tensor(q.T.copy(), device=dev)).T.contiguous()
cfg = nat.make_config(solution_mode="speed", tol_f=1e-6)
bufs = hc.alloc_ik_buffers(K, R)
hc.ik_batch(cfg, targets, x0, 0, R, bufs=bufs, per_restart=True)
torch.cuda.synchronize()
assert hc.last_launch()["lds_bytes"] > 30000
out = (C.c_ulonglong * 8)()
nat.lib().optik_hip_phase_profile.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
nat.check(nat.lib().optik_hip_phase_profile(hc._h, out))
v = list(out)
names = {0: "refill", 1: "evaluation", 4: "bookkeeping + BFGS", 5: "LSQ factor + records", 2: "first NNLS pass (per lane)",
         6: "ranking + NNLS (quads)", 3: "LDP tail .. trial point, publish"}
trips = max(v[7], 1)
tot = sum(v[k] for k in names)
print(f"{robot and sys.argv[1] if len(sys.argv) > 1 else 'panda'}: {trips} wave-trips, {tot / trips:.0f} cycles per trip")
for k in (0, 1, 4, 5, 2, 6, 3):
    print(f"  {names[k]:36s} {v[k] / trips:9.0f} cycles/trip  {100.0 * v[k] / tot:5.1f} %")
