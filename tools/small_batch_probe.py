import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optik_amd import Robot, SolverConfig
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rb = Robot.from_urdf_file(os.path.join(ROOT, "optik_amd/robots/panda.urdf"), "panda_link0", "panda_link8")
if len(sys.argv) > 1: rb.set_parallelism(int(sys.argv[1]))
lb, ub = (np.array(v) for v in rb.joint_limits())
rng = np.random.default_rng(0)
cfg = SolverConfig(max_time=0.0, max_restarts=256)
for T in (1, 4, 16, 64, 256, 1024):
    tg = [np.array(rb.fk(rng.uniform(lb, ub))) for _ in range(T)]
    x0 = rng.uniform(lb, ub, size=(T, 7))
    rb.ik_batch(cfg, tg, x0)
    t0 = time.perf_counter(); n = 5
    for _ in range(n): out = rb.ik_batch(cfg, tg, x0)
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for t in range(min(T, 64)): rb.ik(cfg, tg[t], x0[t].tolist())
    ds = (time.perf_counter() - t0) / min(T, 64)
    print(f"T={T:5d}: ik_batch {dt*1e3:7.2f} ms ({dt/T*1e6:8.1f} us per target), single ik() {ds*1e6:7.1f} us each, solved {sum(o is not None for o in out)}")
