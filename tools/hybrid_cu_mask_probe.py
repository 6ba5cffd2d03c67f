#!/usr/bin/env python3
"""Probe: the streaming engine and the quad kernel on DISJOINT sets of CUs (hipExtStreamCreateWithCUMask).

tools/hybrid_probe.py found both at once on the whole chip slower than either alone: the persistent quad waves
take registers and LDS the engine's 240-VGPR kernels need for their second wave.  With CU masks nothing is
shared but HBM and the fabric: the engine (its sub-pool streams through OPTIK_ENG_CU_MASK, the caller's stream
masked alike, the pool sized for its share) on the first E CUs, the quad kernel on the rest, the bench workload's
K steps split statically between them.  Diagnostic only (not the product path).

usage: hybrid_cu_mask_probe.py [K=20] [R=65536]
"""
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

hip = ctypes.CDLL("libamdhip64.so")


def mask_words(lo, hi, total=256):
    bits = 0
    for i in range(lo, hi):
        bits |= 1 << i
    return [(bits >> (32 * w)) & 0xffffffff for w in range((total + 31) // 32)]


def masked_stream(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    from optik_amd import _native as nat
    from bench import load_chain
    dev = torch.device("cuda", 0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = np.random.default_rng(0)

    def one(e_cus, ke):
        """engine on CUs [0, e_cus) with ke steps, quad kernel on [e_cus, cus) with K - ke steps"""
        kq = K - ke
        q_cus = cus - e_cus
        if ke:
            os.environ["OPTIK_ENG_CU_MASK"] = ",".join(f"{w:x}" for w in mask_words(0, e_cus, cus))
            os.environ["OPTIK_ENGINE_SLOTS"] = str(int(393216 * e_cus / cus) // 3072 * 3072)
        else:
            os.environ.pop("OPTIK_ENG_CU_MASK", None)
            os.environ.pop("OPTIK_ENGINE_SLOTS", None)
        # resident quad waves: 8 per CU of its share, expressed per CU of the whole chip (the launch multiplies by 256)
        os.environ["OPTIK_SOLVE_WAVES_PER_CU"] = str(max(1, round(8 * q_cus / cus))) if kq and e_cus else "8"
        robot = load_chain("panda")
        hc_e = robot.hip_chain(dev)
        hc_q = load_chain("panda").hip_chain(dev)
        n = robot.num_positions()
        lb, ub = (np.array(v) for v in robot.joint_limits())
        r2 = np.random.default_rng(0)
        q_star = r2.uniform(lb, ub, size=(K, n))
        x0 = torch.tensor(r2.uniform(lb, ub, size=(K, n)), device=dev)
        targets = hc_e.fk_batch(torch.tensor(q_star.T.copy(), device=dev)).T.contiguous()
        cfg = nat.make_config(solution_mode="speed", tol_f=1e-6)
        ebufs = [hc_e.alloc_ik_buffers(1, R) for _ in range(ke)]
        qb = hc_q.alloc_ik_buffers(kq, R) if kq else None
        s_e = masked_stream(mask_words(0, e_cus, cus)) if (ke and e_cus < cus) else torch.cuda.Stream()
        s_q = masked_stream(mask_words(e_cus, cus, cus)) if (kq and e_cus > 0) else torch.cuda.Stream()
        torch.cuda.synchronize()

        def eng():
            with torch.cuda.stream(s_e):
                for k in range(ke):
                    hc_e.engine_submit(cfg, targets[k:k + 1], x0[k:k + 1], 0, R, bufs=ebufs[k])
                hc_e.engine_run()
                s_e.synchronize()

        def quad():
            with torch.cuda.stream(s_q):
                hc_q.ik_batch(cfg, targets[ke:], x0[ke:], 0, R, bufs=qb, per_restart=True)
                s_q.synchronize()

        best, t_e, t_q = None, 0.0, 0.0
        for _ in range(4):
            torch.cuda.synchronize()
            done = {}

            def timed(name, fn):
                t = time.perf_counter()
                fn()
                done[name] = time.perf_counter() - t
            th = []
            if ke:
                th.append(threading.Thread(target=timed, args=("e", eng)))
            if kq:
                th.append(threading.Thread(target=timed, args=("q", quad)))
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if best is None or el < best:
                best, t_e, t_q = el, done.get("e", 0.0), done.get("q", 0.0)
        print(f"engine {ke:2d} steps on {e_cus:3d} CUs ({t_e * 1e3:6.2f} ms) + quad {kq:2d} steps on {q_cus:3d} CUs "
              f"({t_q * 1e3:6.2f} ms): {best * 1e3:7.2f} ms  {K * R / best / 1e6:6.2f} M restarts/s", flush=True)
        del hc_e, hc_q

    one(cus, K)      # engine alone, whole chip
    one(0, 0)        # quad alone, whole chip
    for e_cus, ke in [(192, 15), (192, 16), (192, 14), (224, 17), (224, 18), (160, 13), (128, 10)]:
        one(e_cus, ke)


if __name__ == "__main__":
    main()
