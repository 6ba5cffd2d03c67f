#!/usr/bin/env python3
"""How full the lane-per-restart form's NNLS loop runs (diagnostic; a -DOPTIK_PROFILE build of ik_lane_kernel.o:
python tools/build_lib_variant.py prof_lane -DOPTIK_PROFILE -DOPTIK_LANE_ONLY_N=7 --only=ik_lane_kernel.o, then
OPTIK_PROF_LIB=optik_amd/csrc/variants/prof_lane.so python tools/lane_nnls_hist.py): loop trips by the number of quads
still solving, on the bench workload."""
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
K, R = 8, 65536
q = rng.uniform(lb, ub, size=(K, n))
x0 = torch.tensor(rng.uniform(lb, ub, size=(K, n)), device=dev)
targets = hc.fk_batch(torch.tensor(q.T.copy(), device=dev)).T.contiguous()
cfg = nat.make_config(solution_mode="speed", tol_f=1e-6)
bufs = hc.alloc_ik_buffers(K, R)
h = (C.c_ulonglong * 66)()
nat.lib().optik_hip_lane_nnls_hist(h)  # (reset)
cyc = (C.c_ulonglong * 26)()
nat.lib().optik_hip_lane_nnls_cycles(cyc)  # (reset)
hc.ik_batch(cfg, targets, x0, 0, R, bufs=bufs, per_restart=True)
torch.cuda.synchronize()
assert hc.last_launch()["lds_bytes"] > 30000
nat.lib().optik_hip_lane_nnls_hist(h)
h = list(h)
trips = h[:17]
tot = max(sum(trips), 1)
print("loop trips by quads still solving (0..16), %: " + " ".join(f"{100.0 * c / tot:.1f}" for c in trips))
util = sum(k * c for k, c in enumerate(trips)) / (16.0 * tot)
print(f"mean quads solving per loop trip {16 * util:.2f} of 16 = {100 * util:.1f} %")
calls = max(sum(h[17:49]), 1)
print(f"calls {calls}, loop trips per call {tot / calls:.2f}")
print("calls by loop trips (0..31+), %: " + " ".join(f"{100.0 * c / calls:.1f}" for c in h[17:49]))
nat.lib().optik_hip_lane_nnls_cycles(cyc)
cyc = list(cyc)
tc = cyc[:17]
print("wave cycles per loop trip by quads solving (1..16): " + " ".join(f"{(tc[k] / trips[k]) if trips[k] else 0:.0f}" for k in range(1, 17)))
ct = max(sum(tc), 1)
print("share of the loop's cycles by quads solving (1..16), %: " + " ".join(f"{100.0 * tc[k] / ct:.1f}" for k in range(1, 17)))
print(f"loop cycles per call {ct / calls:.0f}; parts (cycles per call): steps 2-4 {cyc[18] / calls:.0f}, step 5 {cyc[19] / calls:.0f}, "
      f"steps 6-10 {cyc[20] / calls:.0f}, step 11 {cyc[21] / calls:.0f}, hand-over / loop head {cyc[25] / calls:.0f}")
