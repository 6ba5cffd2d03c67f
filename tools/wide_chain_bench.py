import time, numpy as np, torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from optik_amd import device, _native as nat
from oracle import binding as ob, urdf_chain
from gpu_util import make_targets
for name, base, ee in (("arm8","l0","l9"),("arm9","l0","l9"),("arm10","l0","l11"),("arm12","l0","l12"),("arm16","l0","l17")):
    d = urdf_chain.chain_from_urdf(open(f"tests/golden/robots/{name}.urdf").read(), base, ee)
    ch = ob.make_chain(**d)
    hc = device.HipChain(**d)
    rng = np.random.default_rng(0)
    tg, x0 = make_targets(ob, d, ch, rng, 1)
    cfg = nat.make_config(solution_mode="speed", tol_f=1e-6)
    tgd, x0d = torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda")
    R = 262144
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        out = hc.ik_batch(cfg, tgd, x0d, 0, R)
        torch.cuda.synchronize(); dt = time.time() - t
    ok = (out["status"] > 0).float().mean().item()
    print(f"{name}: n={len(d['lb'])} {R/dt/1e6:.3f} M restarts/s ({dt*1e3:.1f} ms), success {ok:.3f}, evals/restart {out['evals'].float().mean().item():.1f}", flush=True)
