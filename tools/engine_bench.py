import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from optik_amd import Robot, _native as nat
robot = sys.argv[1] if len(sys.argv) > 1 else "panda"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
R = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
spec = {"panda": ("panda.urdf", "panda_link0", "panda_link8"), "ur10": ("ur10.urdf", "base_link", "ee_link")}[robot]
rb = Robot.from_urdf_file(os.path.join("optik_amd/robots", spec[0]), spec[1], spec[2])
hc = rb.hip_chain("cuda:0")
rng = np.random.default_rng(0)
lb, ub = (np.array(v) for v in rb.joint_limits()); n = len(lb)
tg = hc.fk_batch(torch.tensor(rng.uniform(lb, ub, size=(K, n)).T.copy(), device="cuda:0")).T.contiguous()
x0 = torch.tensor(rng.uniform(lb, ub, size=(K, n)), device="cuda:0")
cfg = nat.make_config("speed")
bufs = [hc.alloc_ik_buffers(1, R) for _ in range(K)]
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): hc.engine_submit(cfg, tg[i:i+1], x0[i:i+1], 0, R, bufs=bufs[i])
    trips = hc.engine_run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"engine {robot}: K={K} R={R} slots={os.environ.get('OPTIK_ENGINE_SLOTS','262144')} trips={trips} {dt*1e3:.1f} ms -> {K*R/dt/1e6:.2f} M restarts/s, {dt/trips*1e6:.1f} us/trip")
