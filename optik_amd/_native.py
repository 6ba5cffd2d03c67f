"""ctypes view of the kernel-layer C ABI (``include/optik_hip.h``) in ``liboptik_amd.so``.

The shared library is built in-tree by ``optik_amd.build`` (hipcc, gfx950).  There is
no CPU fallback anywhere in this package: if the library is missing, or no GPU is
usable, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liboptik_amd.so")
# diagnostics only (tools/: tuning variants and -DOPTIK_PROFILE builds of the same sources)
if os.environ.get("OPTIK_AMD_LIB"):
    LIB_PATH = os.path.abspath(os.environ["OPTIK_AMD_LIB"])

MAX_DOF = 8
IK_EARLY_EXIT = 1
IK_FIND_ANY = 2
IK_RESTART_MAJOR = 4
UINT64_MAX = 0xFFFFFFFFFFFFFFFF

RES_FAILURE, RES_ROUNDOFF, RES_FORCED_STOP, RES_ITER_CAP = -1, -4, -5, -100
RES_STOPVAL, RES_FTOL, RES_XTOL = 2, 3, 4
# rand 0.9.2 random_range(lb..=ub) reading (include/optik_hip.h: OPTIK_HIP_RANGE_*)
RANGE_SINGLE_INCLUSIVE, RANGE_NEW_INCLUSIVE = 0, 1


class OptikHipError(RuntimeError):
    pass


class SolverConfigC(C.Structure):
    """CSolverConfig (optik-cpp/src/lib.rs:10-20), 96 bytes."""
    _fields_ = [
        ("solution_mode", C.c_int32),
        ("_pad", C.c_int32),
        ("max_time", C.c_double),
        ("max_restarts", C.c_uint64),
        ("tol_f", C.c_double),
        ("tol_df", C.c_double),
        ("tol_dx", C.c_double),
        ("linear_weight", C.c_double * 3),
        ("angular_weight", C.c_double * 3),
    ]


class IkOutputs(C.Structure):
    _fields_ = [
        ("d_x", C.c_void_p), ("d_f", C.c_void_p), ("d_status", C.c_void_p), ("d_evals", C.c_void_p),
        ("d_win_x", C.c_void_p), ("d_win_f", C.c_void_p), ("d_win_idx", C.c_void_p),
        ("d_win_key", C.c_void_p),
    ]


class LaunchInfo(C.Structure):
    _fields_ = [("grid", C.c_int32), ("block", C.c_int32), ("lds_bytes", C.c_int32),
                ("tiles", C.c_int32), ("kernel_ms", C.c_float)]


_lib = None


def lib():
    """Load liboptik_amd.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OptikHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  optik_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.optik_hip_device_count.restype = C.c_int
    L.optik_hip_set_option.argtypes = [C.c_char_p, C.c_longlong]
    L.optik_hip_get_option.argtypes = [C.c_char_p]
    L.optik_hip_get_option.restype = C.c_longlong
    L.optik_hip_last_error.restype = C.c_char_p
    L.optik_hip_chain_create.argtypes = [dp, dp, ip, C.c_int32, dp, dp, C.c_int32, C.POINTER(vp)]
    L.optik_hip_chain_destroy.argtypes = [vp]
    L.optik_hip_chain_num_positions.argtypes = [vp]
    L.optik_hip_chain_set_range_rule.argtypes = [vp, C.c_int32]
    L.optik_hip_chain_range_rule.argtypes = [vp]
    L.optik_hip_eval_batch.argtypes = [vp, C.POINTER(SolverConfigC), dp, dp, vp, C.c_int64, vp, vp, vp]
    L.optik_hip_fk_batch.argtypes = [vp, dp, vp, C.c_int64, vp, vp, vp]
    L.optik_hip_seed_batch.argtypes = [vp, C.c_uint64, C.c_int64, vp, vp]
    L.optik_hip_ik_batch.argtypes = [vp, C.POINTER(SolverConfigC), vp, vp, C.c_int32, dp,
                                     C.c_uint64, C.c_uint64, C.c_uint32, C.c_double,
                                     C.POINTER(IkOutputs), vp]
    L.optik_hip_ik_host.argtypes = [vp, C.POINTER(SolverConfigC), dp, dp, C.c_int32, dp,
                                    C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, dp, dp,
                                    C.POINTER(C.c_uint64), dp]
    L.optik_hip_probe.argtypes = [C.c_int32, dp, dp, C.c_int64, dp]
    L.optik_hip_probe_math.argtypes = [C.c_int32, dp, C.c_int64, dp]
    L.optik_hip_set_timing.argtypes = [vp, C.c_int32]
    L.optik_hip_last_launch.argtypes = [vp, C.POINTER(LaunchInfo)]
    L.optik_hip_timing_mean.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        msg = lib().optik_hip_last_error()
        raise OptikHipError(f"optik_hip error {rc}: {msg.decode() if msg else '?'}")


def make_config(solution_mode="speed", max_time=0.0, max_restarts=0, tol_f=1e-6, tol_df=-1.0,
                tol_dx=-1.0, linear_weight=(1.0, 1.0, 1.0), angular_weight=(1.0, 1.0, 1.0)):
    cfg = SolverConfigC()
    cfg.solution_mode = {"quality": 1, "speed": 2}[solution_mode]
    cfg.max_time = float(max_time)
    cfg.max_restarts = int(max_restarts)
    cfg.tol_f, cfg.tol_df, cfg.tol_dx = float(tol_f), float(tol_df), float(tol_dx)
    cfg.linear_weight[:] = [float(v) for v in linear_weight]
    cfg.angular_weight[:] = [float(v) for v in angular_weight]
    return cfg


SOLVE_KERNELS = {"auto": 0, "quad": 1, "lane64": 2, "general": 3}
WIDE_FORMS = {"lds": 0, "hbm": 1}


def set_option(name, value):
    """Tuning option of the kernel layer (include/optik_hip.h: optik_hip_set_option) -- tests and tools.  `value`: an
    integer, or for solve_kernel / wide_form one of their names."""
    if isinstance(value, str):
        value = {"solve_kernel": SOLVE_KERNELS, "wide_form": WIDE_FORMS}[name][value]
    check(lib().optik_hip_set_option(name.encode(), int(value)))


def get_option(name):
    return int(lib().optik_hip_get_option(name.encode()))


class options:
    """``with options(solve_kernel="lane64"): ...`` -- set for the block, restored after it."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.prev = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_option(k, v)
        return False
