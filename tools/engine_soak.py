"""Soak: repeated engine runs in one process -- results identical run to run, device memory flat."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from optik_amd import Robot, _native as nat
rb = Robot.from_urdf_file("optik_amd/robots/panda.urdf", "panda_link0", "panda_link8")
hc = rb.hip_chain("cuda:0")
rng = np.random.default_rng(0)
lb, ub = (np.array(v) for v in rb.joint_limits()); n = len(lb)
K, R = 8, 65536
tg = hc.fk_batch(torch.tensor(rng.uniform(lb, ub, size=(K, n)).T.copy(), device="cuda:0")).T.contiguous()
x0 = torch.tensor(rng.uniform(lb, ub, size=(K, n)), device="cuda:0")
cfg = nat.make_config("speed")
bufs = [hc.alloc_ik_buffers(1, R) for _ in range(K)]
ref = None
for it in range(25):
    for k in range(K):
        hc.engine_submit(cfg, tg[k:k+1], x0[k:k+1], 0, R, bufs=bufs[k])
    hc.engine_run(); torch.cuda.synchronize()
    chk = tuple(int(b["win_idx"][0]) for b in bufs) + (int(sum(int(b["status"].sum()) for b in bufs)),)
    if ref is None: ref = chk
    assert chk == ref, (it, chk, ref)
    if it % 6 == 0:
        free, total = torch.cuda.mem_get_info()
        print(it, "free GB", round(free / 2**30, 3))
print("soak ok", ref[:4])
