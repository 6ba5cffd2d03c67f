#!/usr/bin/env python3
"""Where an update-kernel wave spends its cycles (diagnostic; -DOPTIK_PROFILE build).

s_memtime probes (after draining outstanding memory operations) around the phases of
eng_update_body, summed over the waves that ran a direction search.  One sub-pool, the
default bench workload.  Not part of the product path."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "optik_amd", "csrc")
LIB = os.path.join(ROOT, "gpurun_out", "liboptik_amd_prof.so")


def main():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                           "-shared", "-Wno-unused-value", "-pthread", "-DOPTIK_PROFILE", "-x", "hip",
                           os.path.join(CSRC, "ik_kernels.hip"), os.path.join(CSRC, "robot_host.cpp"), "-o", LIB])
    os.environ["OPTIK_ENG_POOLS"] = "1"  # (read once, at the library's first use)
    from optik_amd import _native as nat
    nat.LIB_PATH = LIB
    import numpy as np
    import torch
    from optik_amd import Robot
    rb = Robot.from_urdf_file(os.path.join(ROOT, "optik_amd", "robots", "panda.urdf"), "panda_link0", "panda_link8")
    hc = rb.hip_chain("cuda:0")
    rng = np.random.default_rng(0)
    lb, ub = (np.array(v) for v in rb.joint_limits())
    K, R = 8, 65536
    tgt = hc.fk_batch(torch.tensor(rng.uniform(lb, ub, size=(K, len(lb))).T.copy(), device="cuda:0")).T.contiguous()
    x0 = torch.tensor(rng.uniform(lb, ub, size=(K, len(lb))), device="cuda:0")
    cfg = nat.make_config("speed")
    bufs = [hc.alloc_ik_buffers(1, R) for _ in range(K)]
    for k in range(K):
        hc.engine_submit(cfg, tgt[k:k + 1], x0[k:k + 1], 0, R, bufs=bufs[k])
    hc.engine_run()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    nat.lib().optik_hip_phase_profile.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    nat.lib().optik_hip_phase_profile(hc._h, out)
    waves = max(out[7], 1)
    names = ["entry + refill", "loads of an accepted step", "BFGS update", "direction search + stores", "whole body"]
    print(f"update kernel, {waves} waves with a direction search; mean wave cycles per phase:")
    for k, nm in enumerate(names):
        print(f"  {nm:28s} {out[k] / waves:10.0f}")
    print(f"  inside the direction search: factor {out[5] / waves:.0f}, bound rows + record stores {out[6] / waves:.0f}, "
          f"rest (listing, back-substitution, plane stores) {(out[3] - out[5] - out[6]) / waves:.0f}")


if __name__ == "__main__":
    main()
