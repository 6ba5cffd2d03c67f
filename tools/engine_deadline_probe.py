import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from optik_amd import _native as nat
from bench import load_chain
dev = torch.device("cuda", 0)
robot = load_chain("panda"); hc = robot.hip_chain(dev); n = robot.num_positions()
rng = np.random.default_rng(29)
lb, ub = (np.array(v) for v in robot.joint_limits())
q = rng.uniform(lb, ub, size=(1, n)); x0 = torch.tensor(rng.uniform(lb, ub, size=(1, n)), device=dev)
tg = hc.fk_batch(torch.tensor(q.T.copy(), device=dev)).T.contiguous()
cfg = nat.make_config(solution_mode="speed")
hc.engine_reserve(); hc.engine_submit(cfg, tg, x0, 0, 1 << 17); hc.engine_run()
for dl in (0.002, 0.005, 0.010, 0.020):
    for rep in range(3):
        out = hc.engine_submit(cfg, tg, x0, 0, 1 << 20)
        t0 = time.perf_counter(); hc.engine_run(deadline_s=dl); took = time.perf_counter() - t0
        torch.cuda.synchronize()
        st = out["status"].cpu().numpy(); ev = out["evals"].cpu().numpy()
        print(f"deadline {dl*1e3:.0f} ms: took {took*1e3:.2f} ms, finished {(st != nat.RES_FORCED_STOP).sum()}, never started {((st == nat.RES_FORCED_STOP) & (ev == 0)).sum()}")
