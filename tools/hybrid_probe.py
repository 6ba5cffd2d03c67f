#!/usr/bin/env python3
"""Probe: do the streaming engine and the quad kernel run faster TOGETHER than either alone?

The engine's phase kernels wait on HBM for much of their life (VALU-busy 0.14 - 0.42), the quad
kernel is issue-bound and touches no memory; the engine also leaves the chip half empty while its
slot pool drains.  This splits the bench workload (K steps x R restarts) between the two on two
streams of one device and times the whole.  Diagnostic only (not the product path).

usage: hybrid_probe.py [K=20] [R=65536]   (prints one line per split)
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    from optik_amd import _native as nat
    from bench import load_chain
    dev = torch.device("cuda", 0)
    robot = load_chain("panda")
    hc_e = robot.hip_chain(dev)
    hc_q = load_chain("panda").hip_chain(dev)
    n = robot.num_positions()
    rng = np.random.default_rng(0)
    lb, ub = (np.array(v) for v in robot.joint_limits())
    q_star = rng.uniform(lb, ub, size=(K, n))
    x0 = torch.tensor(rng.uniform(lb, ub, size=(K, n)), device=dev)
    targets = hc_e.fk_batch(torch.tensor(q_star.T.copy(), device=dev)).T.contiguous()
    cfg = nat.make_config(solution_mode="speed", tol_f=1e-6)
    ebufs = [hc_e.alloc_ik_buffers(1, R) for _ in range(K)]
    qbufs = {}
    hc_e.engine_reserve()
    s_e, s_q = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()

    def run(ke, waves, delay_ms):
        """ke steps on the engine, K - ke on the quad kernel with `waves` resident waves per CU."""
        kq = K - ke
        if kq and kq not in qbufs:
            qbufs[kq] = hc_q.alloc_ik_buffers(kq, R)
        if waves:
            os.environ["OPTIK_SOLVE_WAVES_PER_CU"] = str(waves)
        else:
            os.environ.pop("OPTIK_SOLVE_WAVES_PER_CU", None)

        def eng():
            with torch.cuda.stream(s_e):
                for k in range(ke):
                    hc_e.engine_submit(cfg, targets[k:k + 1], x0[k:k + 1], 0, R, bufs=ebufs[k])
                hc_e.engine_run()

        def quad():
            if delay_ms:
                time.sleep(delay_ms * 1e-3)
            with torch.cuda.stream(s_q):
                hc_q.ik_batch(cfg, targets[ke:], x0[ke:], 0, R, bufs=qbufs[kq], per_restart=True)
                s_q.synchronize()

        best = None
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = []
            if ke:
                th.append(threading.Thread(target=eng))
            if kq:
                th.append(threading.Thread(target=quad))
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            best = el if best is None or el < best else best
        print(f"engine {ke:2d} steps + quad {kq:2d} steps ({waves or 'all'} waves/CU, quad starts +{delay_ms} ms): "
              f"{best * 1e3:7.2f} ms  {K * R / best / 1e6:6.2f} M restarts/s", flush=True)

    run(K, 0, 0)
    run(0, 0, 0)
    for ke, waves, delay in [(16, 4, 0), (15, 4, 0), (14, 4, 0), (16, 2, 0), (17, 4, 20), (16, 4, 20), (17, 8, 30),
                             (16, 8, 25), (18, 8, 35), (12, 4, 0), (10, 4, 0)]:
        if ke < K:
            run(ke, waves, delay)


if __name__ == "__main__":
    main()
