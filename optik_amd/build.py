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
SOURCES = ["ik_kernels.hip", "ik_quad_kernel.hip", "ik_wide_kernel.hip", "robot_host.cpp"]
# translation units: (source, object, extra flags).  ik_quad_kernel.hip is compiled twice -- its
# throughput form (two waves per SIMD) without the machine-LICM pass, which otherwise hoists constants and
# LDS addresses out of the solver loop only for the register allocator to spill them to scratch
# (csrc/ik_quad_kernel.hip), its latency forms and the launch function with the default pipeline.
UNITS = [("ik_kernels.hip", "ik_kernels.o", []),
         # (the quad solver is issue-bound: the max-ILP scheduling strategy is worth +3 % restarts/s on the
         # throughput form and -3 % on a single ik()'s latency; on the engine's kernels it costs 6 %)
         ("ik_quad_kernel.hip", "ik_quad_latency.o",
          ["-DOPTIK_QUAD_PART=1", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
         ("ik_quad_kernel.hip", "ik_quad_throughput.o",
          ["-DOPTIK_QUAD_PART=2", "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-use-amdgpu-trackers=1",
           "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
         # chains with 9 .. 16 joint positions: one run-time-n body per kernel (ik_wide.hpp)
         ("ik_wide_kernel.hip", "ik_wide_kernel.o", []),
         ("robot_host.cpp", "robot_host.o", [])]
HEADERS = ["ik_math.hpp", "ik_eval.hpp", "ik_slsqp.hpp", "ik_solve.hpp", "ik_nnls_coop.hpp", "ik_engine.hpp",
           "ik_tail.hpp", "ik_coop.hpp", "device_scope.hpp", "ik_host_params.hpp", "ik_launch.hpp",
           "urdf_chain.hpp", "ik_wide_launch.hpp",
           os.path.join("..", "..", "include", "optik_hip.h"),
           os.path.join("..", "..", "include", "optik.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wno-unused-value", "-pthread"]
# headers each translation unit depends on (a source is recompiled when one of them is newer than
# its object file; the objects are build artefacts, git-ignored like the library)
QUAD_HEADERS = ["ik_math.hpp", "ik_eval.hpp", "ik_slsqp.hpp", "ik_solve.hpp", "ik_nnls_coop.hpp", "ik_nnls_quad.hpp", "ik_lane.hpp",
                "ik_quad.hpp", "ik_launch.hpp"]
WIDE_HEADERS = ["ik_math.hpp", "ik_eval.hpp", "ik_slsqp.hpp", "ik_solve.hpp", "ik_wide_launch.hpp", "ik_wide.hpp"]
DEPS = {"ik_kernels.hip": HEADERS,
        "ik_wide_kernel.hip": WIDE_HEADERS,
        "ik_quad_kernel.hip": QUAD_HEADERS,
        "robot_host.cpp": ["urdf_chain.hpp", "device_scope.hpp", os.path.join("..", "..", "include", "optik_hip.h"),
                           os.path.join("..", "..", "include", "optik.h")]}


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS + QUAD_HEADERS + WIDE_HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _newer(path, than):
    return os.path.exists(path) and os.path.getmtime(path) > than


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    hipcc = _hipcc()
    objs = []
    for src, objname, extra in UNITS:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(CSRC, objname)
        t = os.path.getmtime(obj) if os.path.exists(obj) else -1.0
        deps = [sp] + [os.path.join(CSRC, d) for d in DEPS.get(src, HEADERS)]
        cmd = [hipcc, *FLAGS, *extra, "-x", "hip", "-c", sp, "-o", obj]
        # (an object is also stale when it was compiled with other flags: they are kept next to it)
        flags_file = obj + ".flags"
        same_flags = os.path.exists(flags_file) and open(flags_file).read() == " ".join(cmd[1:])
        if force or t < 0 or not same_flags or any(_newer(d, t) for d in deps):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=CSRC)
            with open(flags_file, "w") as fh:
                fh.write(" ".join(cmd[1:]))
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
