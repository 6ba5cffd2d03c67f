import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from optik_amd import _native as nat
from optik_amd.parallel import select_winner
from bench import load_chain
dev = torch.device("cuda", 0)
robot = load_chain("panda"); hc = robot.hip_chain(dev); n = robot.num_positions()
K, R = 20, 65536
rng = np.random.default_rng(0)
lb, ub = (np.array(v) for v in robot.joint_limits())
q = rng.uniform(lb, ub, size=(K, n)); x0 = torch.tensor(rng.uniform(lb, ub, size=(K, n)), device=dev)
targets = hc.fk_batch(torch.tensor(q.T.copy(), device=dev)).T.contiguous()
cfg = nat.make_config(solution_mode="speed", tol_f=1e-6)
bufs = [hc.alloc_ik_buffers(1, R) for _ in range(K)]
win_idx_all = torch.zeros((K, 1), dtype=torch.int64, device=dev); win_key_all = torch.zeros((K, 1), dtype=torch.float64, device=dev)
for k, b in enumerate(bufs): b["win_idx"] = win_idx_all[k]; b["win_key"] = win_key_all[k]
hc.engine_reserve()
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K): hc.engine_submit(cfg, targets[k:k+1], x0[k:k+1], 0, R, bufs=bufs[k])
    t1 = time.perf_counter()
    hc.engine_run(); t2 = time.perf_counter()
    w = select_winner({"win_idx": win_idx_all.reshape(-1), "win_key": win_key_all.reshape(-1)}, "speed", False)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"submit {1e3*(t1-t0):.2f} ms  run {1e3*(t2-t1):.2f} ms  select+sync {1e3*(t3-t2):.2f} ms  total {1e3*(t3-t0):.2f}")
