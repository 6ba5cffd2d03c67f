#!/usr/bin/env python3
"""Per-phase cycle breakdown of the solve kernel (diagnostic).

Builds a second copy of the library with -DOPTIK_PROFILE (s_memtime counters around
the phases of a wave's trip), runs one bench-sized launch and prints where the wave
cycles go.  Not part of the product path.
"""
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
    robot = sys.argv[1] if len(sys.argv) > 1 else "panda"
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    lib = os.environ.get("OPTIK_PROF_LIB")  # a -DOPTIK_PROFILE build made beforehand (hipcc cross-compiles without a GPU)
    if not lib:
        lib = LIB
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                               "-shared", "-Wno-unused-value", "-pthread", "-DOPTIK_PROFILE", "-x", "hip",
                               os.path.join(CSRC, "ik_kernels.hip"), os.path.join(CSRC, "ik_quad_kernel.hip"),
                               os.path.join(CSRC, "robot_host.cpp"), "-o", lib])
    from optik_amd import _native as nat
    nat.LIB_PATH = os.path.abspath(lib)
    import numpy as np
    import torch
    from optik_amd import Robot
    spec = {"panda": ("panda.urdf", "panda_link0", "panda_link8"), "ur10": ("ur10.urdf", "base_link", "ee_link")}[robot]
    rb = Robot.from_urdf_file(os.path.join(ROOT, "optik_amd", "robots", spec[0]), spec[1], spec[2])
    hc = rb.hip_chain("cuda:0")
    rng = np.random.default_rng(0)
    lb, ub = (np.array(v) for v in rb.joint_limits())
    tgt = hc.fk_batch(torch.tensor(rng.uniform(lb, ub, size=(1, len(lb))).T.copy(), device="cuda:0")).T.contiguous()
    x0 = torch.tensor(rng.uniform(lb, ub, size=(1, len(lb))), device="cuda:0")
    cfg = nat.make_config("speed")
    bufs = hc.alloc_ik_buffers(1, R)
    qn = (C.c_ulonglong * 8)()
    have_qn = hasattr(nat.lib(), "optik_hip_quad_nnls_profile")
    for _ in range(2):
        if have_qn:
            nat.lib().optik_hip_quad_nnls_profile(qn)  # (reset)
        if have_qn and hasattr(nat.lib(), "optik_hip_quad_nnls_hist"):
            nat.lib().optik_hip_quad_nnls_hist((C.c_ulonglong * 66)())  # (reset)
        hc.ik_batch(cfg, tgt, x0, 0, R, bufs=bufs)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    nat.lib().optik_hip_phase_profile.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    nat.check(nat.lib().optik_hip_phase_profile(hc._h, out))
    names = ["refill", "eval", "(unused)", "publish", "bookkeeping+bfgs", "direction(total)", "  nnls", "trips"]
    v = list(out)
    total = v[0] + v[1] + v[2] + v[3] + v[4] + v[5]
    trips = max(v[7], 1)
    print(f"{robot}: R={R}, wave-trips={trips}, mean evals/restart={float(bufs['evals'].double().mean()):.1f}")
    for n_, c in zip(names[:7], v[:7]):
        print(f"  {n_:16s} {c / trips:10.0f} cycles/trip  {100.0 * c / total:5.1f} %")
    print(f"  total            {total / trips:10.0f} cycles/trip")
    if have_qn and os.environ.get("OPTIK_SOLVE_KERNEL") in (None, "quad"):
        nat.lib().optik_hip_quad_nnls_profile(qn)
        q = list(qn)
        calls, loops = max(q[5], 1), max(q[4], 1)
        print(f"  quad NNLS: {calls} wave-level calls ({calls / trips:.2f} per trip), {loops / calls:.2f} loop trips per call, "
              f"{q[6] / calls:.2f} Givens steps per call")
        for n_, c in zip(["steps 2-4 (duals, argmax)", "step 5 (Householder)", "steps 6-10 (solve, step)", "step 11 (remove)"], q[:4]):
            print(f"    {n_:26s} {c / calls:9.0f} cycles/call {c / loops:9.0f} cycles/loop trip")
        if hasattr(nat.lib(), "optik_hip_quad_nnls_hist"):
            h = (C.c_ulonglong * 66)()
            nat.lib().optik_hip_quad_nnls_hist(h)
            h = list(h)
            tot = max(sum(h[:17]), 1)
            print("    loop trips by quads still solving (0..16): " + " ".join(f"{100.0 * c / tot:.1f}" for c in h[:17]))
            totc = max(sum(h[17:49]), 1)
            print("    calls by loop trips (0..31+): " + " ".join(f"{100.0 * c / totc:.1f}" for c in h[17:49]))
            totp = max(sum(h[49:66]), 1)
            print("    calls by quads taking part (0..16): " + " ".join(f"{100.0 * c / totp:.1f}" for c in h[49:66]))


if __name__ == "__main__":
    main()
