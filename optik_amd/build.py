"""In-tree build of the gfx950 shared library (hipcc cross-compiles without a GPU).

``liboptik_amd.so`` = HIP kernels + the kernel-layer C ABI (include/optik_hip.h) +
the reference-compatible host API (include/optik.h, ``optik_robot_*``).
-ffp-contract=off is part of the numerical contract (see csrc/ik_math.hpp).
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "liboptik_amd.so")
SOURCES = ["ik_kernels.hip", "robot_host.cpp"]
HEADERS = ["ik_math.hpp", "ik_eval.hpp", "ik_slsqp.hpp", "ik_solve.hpp", "ik_nnls_coop.hpp", "ik_engine.hpp",
           "ik_tail.hpp",
           "urdf_chain.hpp",
           os.path.join("..", "..", "include", "optik_hip.h"),
           os.path.join("..", "..", "include", "optik.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-value", "-pthread"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [_hipcc(), *FLAGS, "-x", "hip", *srcs, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
