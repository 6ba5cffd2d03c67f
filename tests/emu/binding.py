"""ctypes binding of the host emulation of the quad solver (tests/emu/quad_emu.cpp).

Test infrastructure: built with the ROCm clang as plain host C++ (-ffp-contract=off), the device
headers of optik_amd/csrc compiled under OPTIK_LANE_EMU.  Nothing in the product imports this."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "optik_amd", "csrc")
LIB = os.path.join(HERE, "libquad_emu.so")
SRC = os.path.join(HERE, "quad_emu.cpp")
_lib = None


def _clang():
    for cand in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("amdclang++"), shutil.which("clang++")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("no clang++ found (the emulation needs ext_vector_type)")


def build(force=False):
    deps = [SRC, os.path.join(HERE, "lane_emu.hpp")] + [
        os.path.join(CSRC, h) for h in ("ik_quad.hpp", "ik_lane64.hpp", "ik_lane.hpp", "ik_platform.hpp", "ik_math.hpp", "ik_eval.hpp", "ik_slsqp.hpp",
                                        "ik_solve.hpp", "ik_nnls_quad.hpp", "ik_nnls_first.hpp", "ik_host_params.hpp")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    subprocess.check_call([_clang(), "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
                           "-I", HERE, "-I", CSRC, "-Wno-unused-value", "-Wno-psabi",
                           # (the emulated waves are a few lanes wide: the per-lane first NNLS pass of ik_nnls_first.hpp, which the
                           # device only runs on trips with more than 24 problems, runs on every trip here)
                           "-DOPTIK_LANE_FIRST_PASS_MIN=0", *os.environ.get("OPTIK_EMU_EXTRA_FLAGS", "").split(), SRC, "-o", LIB])
    return LIB


def lib():
    global _lib
    if _lib is None:
        from optik_amd import _native as nat
        L = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.quad_emu_solve.argtypes = [dp, dp, C.c_int, C.c_int, dp, dp, C.POINTER(nat.SolverConfigC), dp, dp, dp,
                                     C.c_uint64, C.c_uint64, C.c_int, C.c_int, dp, dp, dp, ip, ip, C.c_int]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def solve(chain, cfg, target7, x0, begin, end, quads=1, range_rule=0, ee_offset7=None, lane64=False):
    """chain: dict(types, origins[J,7], axes[J,3], lb, ub).  Returns dict(x[R,n], f, key, status, evals)."""
    origins = np.ascontiguousarray(chain["origins"], dtype=np.float64)
    n = len(chain["lb"])
    axes = np.ascontiguousarray(np.asarray(chain["axes"], dtype=np.float64)[:n])
    lb, ub = (np.ascontiguousarray(chain[k], dtype=np.float64) for k in ("lb", "ub"))
    tg = np.ascontiguousarray(target7, dtype=np.float64)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    ee = np.ascontiguousarray(ee_offset7, dtype=np.float64) if ee_offset7 is not None else None
    R = end - begin
    out_x = np.zeros((n, R))
    out_f, out_key = np.zeros(R), np.zeros(R)
    status, evals = np.zeros(R, dtype=np.int32), np.zeros(R, dtype=np.int32)
    rc = lib().quad_emu_solve(_dp(origins), _dp(axes), n, origins.shape[0], _dp(lb), _dp(ub), C.byref(cfg), _dp(tg),
                              _dp(x0), _dp(ee) if ee is not None else None, begin, end, quads, range_rule,
                              _dp(out_x), _dp(out_f), _dp(out_key), status.ctypes.data_as(C.POINTER(C.c_int32)),
                              evals.ctypes.data_as(C.POINTER(C.c_int32)), 1 if lane64 else 0)
    if rc:
        raise RuntimeError(f"quad_emu_solve rc={rc}")
    return dict(x=out_x.T.copy(), f=out_f, key=out_key, status=status, evals=evals)
