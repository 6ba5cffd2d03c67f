"""Two engine instances on two streams / host threads, splitting the steps (diagnostic)."""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from optik_amd import Robot, _native as nat
from optik_amd.device import HipChain
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NE = int(sys.argv[2]) if len(sys.argv) > 2 else 2
R = 65536
rb = Robot.from_urdf_file("optik_amd/robots/panda.urdf", "panda_link0", "panda_link8")
tabs = rb.chain_tables()
hcs = [HipChain(**tabs) for _ in range(NE)]
streams = [torch.cuda.Stream() for _ in range(NE)]
rng = np.random.default_rng(0)
lb, ub = (np.array(v) for v in rb.joint_limits()); n = len(lb)
tg = hcs[0].fk_batch(torch.tensor(rng.uniform(lb, ub, size=(K, n)).T.copy(), device="cuda:0")).T.contiguous()
x0 = torch.tensor(rng.uniform(lb, ub, size=(K, n)), device="cuda:0")
cfg = nat.make_config("speed")
bufs = [hcs[0].alloc_ik_buffers(1, R) for _ in range(K)]
torch.cuda.synchronize()
def work(e):
    with torch.cuda.stream(streams[e]):
        for i in range(e, K, NE):
            hcs[e].engine_submit(cfg, tg[i:i+1], x0[i:i+1], 0, R, bufs=bufs[i])
        hcs[e].engine_run()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(e,)) for e in range(NE)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"engines={NE} K={K}: {dt*1e3:.1f} ms -> {K*R/dt/1e6:.2f} M restarts/s")
