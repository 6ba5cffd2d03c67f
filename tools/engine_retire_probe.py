#!/usr/bin/env python3
"""Round 5: does the streaming engine still win anywhere the host API uses it?  Robot.ik_batch_arrays (config 5's
shape) with and without OPTIK_BATCH_NO_ENGINE, Speed and Quality, per batch size.  One process per setting."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(T, R, mode):
    import numpy as np
    import torch
    from optik_amd import Robot, SolverConfig
    robot = Robot.from_urdf_file(os.path.join(ROOT, "optik_amd", "robots", "panda.urdf"), "panda_link0", "panda_link8")
    rng = np.random.default_rng(0)
    lb, ub = (np.array(v) for v in robot.joint_limits())
    hc = robot.hip_chain("cuda:0")
    q = rng.uniform(lb, ub, size=(T, 7))
    pose = hc.fk_batch(torch.tensor(q.T.copy(), device="cuda:0")).T.cpu().numpy()
    x0s = rng.uniform(lb, ub, size=(T, 7))
    i, j, k, w = pose[:, 3], pose[:, 4], pose[:, 5], pose[:, 6]
    targets = np.zeros((T, 4, 4))
    targets[:, 0, 0] = w*w+i*i-j*j-k*k; targets[:, 0, 1] = 2*(i*j-w*k); targets[:, 0, 2] = 2*(w*j+i*k)
    targets[:, 1, 0] = 2*(w*k+i*j); targets[:, 1, 1] = w*w-i*i+j*j-k*k; targets[:, 1, 2] = 2*(j*k-w*i)
    targets[:, 2, 0] = 2*(i*k-w*j); targets[:, 2, 1] = 2*(w*i+j*k); targets[:, 2, 2] = w*w-i*i-j*j+k*k
    targets[:, :3, 3] = pose[:, :3]; targets[:, 3, 3] = 1.0
    cfg = SolverConfig(solution_mode=mode, max_time=0.0, max_restarts=R)
    robot.ik_batch_arrays(cfg, targets, x0s)
    dts = []
    for _ in range(4):
        t0 = time.perf_counter()
        x, f, ok = robot.ik_batch_arrays(cfg, targets, x0s)
        dts.append(time.perf_counter() - t0)
    import hashlib
    h = hashlib.sha1(np.ascontiguousarray(x).tobytes() + np.ascontiguousarray(ok).tobytes()).hexdigest()[:12]
    print(f"{mode:8s} T={T:7d} R={R:4d} engine={'off' if os.environ.get('OPTIK_BATCH_NO_ENGINE') else 'on '}: "
          f"min {min(dts)*1e3:8.2f} ms  ({T/min(dts):12,.0f} ik/s)  solved {100*ok.mean():.2f} %  sha {h}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    cases = [(4096, 256, "speed"), (32768, 256, "speed"), (49152, 256, "speed"), (131072, 256, "speed"),
             (4096, 256, "quality"), (16384, 256, "quality"), (1024, 4096, "quality")]
    for T, R, mode in cases:
        for off in (False, True):
            env = dict(os.environ)
            env.pop("OPTIK_BATCH_NO_ENGINE", None)
            if off:
                env["OPTIK_BATCH_NO_ENGINE"] = "1"
            subprocess.run([sys.executable, __file__, str(T), str(R), mode], env=env, timeout=600)
