#!/usr/bin/env python3
"""Wave cycles per step of the cooperative NNLS (ik_nnls_coop.hpp) in the latency configuration
(one restart per wave on the cooperative solve kernel): builds a -DOPTIK_PROFILE_NNLS copy of the
library, runs R restarts of one Panda target and prints cycles per NNLS call and per loop trip.
Diagnostic; not part of the product path.  Usage: python tools/nnls_step_profile.py [R]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "optik_amd", "csrc")
LIB = os.path.join(ROOT, "gpurun_out", "liboptik_amd_nnlsprof.so")
os.makedirs(os.path.dirname(LIB), exist_ok=True)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                       "-Wno-unused-value", "-pthread", "-DOPTIK_PROFILE_NNLS", "-x", "hip",
                       os.path.join(CSRC, "ik_kernels.hip"), os.path.join(CSRC, "robot_host.cpp"), "-o", LIB])
from optik_amd import _native as nat  # noqa: E402
nat.LIB_PATH = LIB
import numpy as np  # noqa: E402
import torch  # noqa: E402
from optik_amd import Robot  # noqa: E402
rb = Robot.from_urdf_file(os.path.join(ROOT, "optik_amd", "robots", "panda.urdf"), "panda_link0", "panda_link8")
hc = rb.hip_chain("cuda:0")
rng = np.random.default_rng(0)
lb, ub = (np.array(v) for v in rb.joint_limits())
tgt = hc.fk_batch(torch.tensor(rng.uniform(lb, ub, size=(1, 7)).T.copy(), device="cuda:0")).T.contiguous()
x0 = torch.tensor(rng.uniform(lb, ub, size=(1, 7)), device="cuda:0")
out = (C.c_ulonglong * 8)()
L = nat.lib()
L.optik_hip_nnls_step_profile.argtypes = [C.POINTER(C.c_ulonglong)]
hc.ik_batch(nat.make_config("speed"), tgt, x0, 0, R)
L.optik_hip_nnls_step_profile(out)
hc.ik_batch(nat.make_config("speed"), tgt, x0, 0, R)
L.optik_hip_nnls_step_profile(out)
v = list(out)
calls, trips = max(v[7], 1), max(v[6], 1)
names = ["steps 2-3 (duals, argmax)", "step 5 construction", "step 5 applied to columns", "step 6 (solve)",
         "steps 7-10 (step length)", "step 11 (remove)"]
print(f"R={R}: {calls} NNLS calls by wave leaders, {trips / calls:.2f} loop trips per call")
for n_, c in zip(names, v[:6]):
    print(f"  {n_:28s} {c / calls:9.0f} cycles/call {c / trips:9.0f} cycles/trip")
print(f"  {'sum':28s} {sum(v[:6]) / calls:9.0f} cycles/call")
