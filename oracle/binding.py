"""ctypes binding of the C oracle (``liboptik_oracle.so``) -- TEST INFRASTRUCTURE ONLY.

See ``optik_oracle.h`` for what each entry point restates.  Nothing under
``optik_amd/`` imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboptik_oracle.so")

MAX_JOINTS = 17
MAX_DOF = 16

RES_STOPVAL, RES_FTOL, RES_XTOL = 2, 3, 4
RES_FAILURE, RES_ROUNDOFF, RES_FORCED, RES_ITER_CAP = -1, -4, -5, -100
# how rand 0.9.2's random_range(lb..=ub) is restated (optik_oracle.c:ok_uniform_scale)
RANGE_SINGLE_INCLUSIVE, RANGE_NEW_INCLUSIVE = 0, 1


class Pose(C.Structure):
    _fields_ = [("t", C.c_double * 3), ("q", C.c_double * 4)]

    @staticmethod
    def make(t=(0, 0, 0), q=(0, 0, 0, 1)):
        p = Pose()
        p.t[:] = [float(v) for v in t]
        p.q[:] = [float(v) for v in q]
        return p

    def as7(self):
        return np.array(list(self.t) + list(self.q))


class Chain(C.Structure):
    _fields_ = [
        ("n_joints", C.c_int32),
        ("n_pos", C.c_int32),
        ("type", C.c_int32 * MAX_JOINTS),
        ("_pad", C.c_int32),
        ("origin", Pose * MAX_JOINTS),
        ("axis", (C.c_double * 3) * MAX_JOINTS),
        ("lb", C.c_double * MAX_DOF),
        ("ub", C.c_double * MAX_DOF),
    ]


class Config(C.Structure):
    """CSolverConfig layout (optik-cpp/src/lib.rs:10-20)."""
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


class RestartResult(C.Structure):
    _fields_ = [
        ("result", C.c_int32),
        ("success", C.c_int32),
        ("n_evals", C.c_int32),
        ("n_iters", C.c_int32),
        ("f", C.c_double),
        ("x", C.c_double * MAX_DOF),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc if the shared object is missing or stale."""
    src = os.path.join(_HERE, "optik_oracle.c")
    hdr = os.path.join(_HERE, "optik_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboptik_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_lib_path = _LIB_PATH  # which build lib() loads (use_native_build() switches it)

NATIVE_FLAGS = ["-O3", "-march=native", "-flto", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-pthread"]


def use_native_build() -> str:
    """Compile the oracle for THIS host (-O3 -march=native -flto: the stand-in for the reference's
    release profile, Cargo.toml:36-39 / SURVEY 8d) into liboptik_oracle_native.so and make lib()
    load it.  Same arithmetic (-ffp-contract=off, no fast-math): bit-identical results, checked by
    tests/test_oracle_native_build.py.  Used by bench.py's cpu_baseline leg only: the portable
    build is what travels to the GPU box with the repository, an -march=native build cannot.
    Falls back to the portable build when the compiler is missing; returns the flags in use."""
    global _lib, _lib_path
    out = os.path.join(_HERE, "liboptik_oracle_native.so")
    src = os.path.join(_HERE, "optik_oracle.c")
    try:
        subprocess.check_call(["gcc", *NATIVE_FLAGS, "-shared", "-o", out, src, "-lm", "-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        return "portable: -O3 -ffp-contract=off (native build failed)"
    pool_stop()
    _lib, _lib_path = None, out
    return "gcc " + " ".join(NATIVE_FLAGS)


def use_libm_build() -> str:
    """The libm-sensitivity study only: the oracle with ok_sincos / ok_atan2_q1 routed to glibc's sincos / atan2
    (-DOK_PLATFORM_LIBM: what f64::sin_cos / f64::atan2 bind on Linux, math.rs:54,76,113,144) into
    liboptik_oracle_libm.so; lib() loads it.  NOT what any parity test compares the GPU with."""
    global _lib, _lib_path
    out = os.path.join(_HERE, "liboptik_oracle_libm.so")
    src = os.path.join(_HERE, "optik_oracle.c")
    deps = (src, os.path.join(_HERE, "optik_oracle.h"))
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["gcc", *NATIVE_FLAGS, "-DOK_PLATFORM_LIBM", "-shared", "-o", out, src, "-lm", "-lpthread"])
    pool_stop()
    _lib, _lib_path = None, out
    return "gcc " + " ".join(NATIVE_FLAGS) + " -DOK_PLATFORM_LIBM"


def use_portable_build():
    global _lib, _lib_path
    pool_stop()
    _lib, _lib_path = None, _LIB_PATH


FLOPS_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fpermissive", "-w", "-pthread"]


def use_flops_build() -> str:
    """Compile optik_oracle_flops.cpp (the oracle with a counting double: every f64 + - * / sqrt it executes is
    counted) into liboptik_oracle_flops.so and make lib() load it.  Same results bit for bit, ~10 x slower, counters
    process-wide: call ik(..., n_threads=1) between flop_reset() and flop_counts().  Raises if g++ is missing."""
    global _lib, _lib_path
    out = os.path.join(_HERE, "liboptik_oracle_flops.so")
    src = os.path.join(_HERE, "optik_oracle_flops.cpp")
    deps = (src, os.path.join(_HERE, "optik_oracle.c"), os.path.join(_HERE, "optik_oracle.h"))
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", *FLOPS_FLAGS, "-shared", "-o", out, src, "-lm", "-lpthread"], cwd=_HERE)
    pool_stop()
    _lib, _lib_path = None, out
    return "g++ " + " ".join(FLOPS_FLAGS)


FLOP_NAMES = ("add_sub", "mul", "div", "sqrt", "compare", "sign_abs_minmax", "int_conversions")


def flop_reset():
    lib().ok_flop_reset()


def flop_counts() -> dict:
    """Counters of the counting build since the last flop_reset(); `flops` = add_sub + mul + div + sqrt."""
    out = (C.c_ulonglong * 8)()
    lib().ok_flop_counts(out)
    d = {k: int(out[i]) for i, k in enumerate(FLOP_NAMES)}
    d["flops"] = d["add_sub"] + d["mul"] + d["div"] + d["sqrt"]
    return d


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_lib_path):
            build()
        L = C.CDLL(_lib_path)
        dp = C.POINTER(C.c_double)
        L.ok_so3_log.argtypes = [dp, dp]
        L.ok_so3_right_jacobian.argtypes = [dp, dp]
        L.ok_se3_log.argtypes = [C.POINTER(Pose), dp]
        L.ok_se3_right_jacobian.argtypes = [C.POINTER(Pose), dp]
        L.ok_pose_from_rpy.argtypes = [dp, dp, C.POINTER(Pose)]
        L.ok_quat_from_matrix.argtypes = [dp, C.c_int, dp]
        L.ok_fk.argtypes = [C.POINTER(Chain), dp, C.POINTER(Pose), C.POINTER(Pose), C.POINTER(Pose)]
        L.ok_joint_jacobian.argtypes = [C.POINTER(Chain), C.POINTER(Pose), C.POINTER(Pose), dp]
        L.ok_eval.argtypes = [C.POINTER(Chain), C.POINTER(Pose), C.POINTER(Pose), dp, dp, dp, dp]
        L.ok_eval.restype = C.c_double
        L.ok_chacha_block.argtypes = [C.POINTER(C.c_uint32), C.c_uint64, C.c_uint64, C.c_int,
                                      C.POINTER(C.c_uint32)]
        L.ok_seed_from_u64.argtypes = [C.c_uint64, C.POINTER(C.c_uint32)]
        L.ok_uniform_inclusive.argtypes = [C.c_double, C.c_double, C.c_uint64]
        L.ok_uniform_inclusive.restype = C.c_double
        L.ok_uniform_scale.argtypes = [C.c_double, C.c_double, C.c_int]
        L.ok_uniform_scale.restype = C.c_double
        L.ok_set_range_rule.argtypes = [C.c_int]
        L.ok_set_stop_x_zero.argtypes = [C.c_int]
        L.ok_get_range_rule.restype = C.c_int
        L.ok_restart_seed.argtypes = [C.POINTER(Chain), C.c_uint64, dp]
        L.ok_solve_restart.argtypes = [C.POINTER(Chain), C.POINTER(Config), C.POINTER(Pose),
                                       C.POINTER(Pose), dp, C.c_uint64,
                                       C.POINTER(RestartResult), dp, C.c_int, C.POINTER(C.c_int)]
        L.ok_ik.argtypes = [C.POINTER(Chain), C.POINTER(Config), C.POINTER(Pose), C.POINTER(Pose),
                            dp, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                            C.POINTER(C.c_uint64), dp, dp, C.POINTER(RestartResult),
                            C.POINTER(C.c_uint64)]
        L.ok_ik.restype = C.c_int
        L.ok_lsq_direction.argtypes = [C.c_int, dp, dp, dp, dp, dp]
        L.ok_lsq_direction.restype = C.c_int
        L.ok_pool_start.argtypes = [C.c_int]
        L.ok_pool_start.restype = C.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def make_chain(types, origins, axes, lb, ub) -> Chain:
    """Flat chain table (as produced by ``urdf_chain.chain_from_urdf``) -> ok_chain."""
    types = np.asarray(types, dtype=np.int32)
    origins = _f64(origins).reshape(-1, 7)
    axes = _f64(axes).reshape(-1, 3)
    J = len(types)
    assert J <= MAX_JOINTS
    ch = Chain()
    ch.n_joints = J
    ch.n_pos = int(np.sum(types != 0))
    assert ch.n_pos <= MAX_DOF and ch.n_pos == len(lb) == len(ub)
    for j in range(J):
        ch.type[j] = int(types[j])
        ch.origin[j].t[:] = origins[j, :3].tolist()
        ch.origin[j].q[:] = origins[j, 3:].tolist()
        ch.axis[j][:] = axes[j].tolist()
    for k in range(ch.n_pos):
        ch.lb[k] = float(lb[k])
        ch.ub[k] = float(ub[k])
    return ch


def make_config(solution_mode="speed", max_time=0.0, max_restarts=0, tol_f=1e-6, tol_df=-1.0,
                tol_dx=-1.0, linear_weight=(1, 1, 1), angular_weight=(1, 1, 1)) -> Config:
    cfg = Config()
    cfg.solution_mode = {"quality": 1, "speed": 2}[solution_mode]
    cfg.max_time = max_time
    cfg.max_restarts = max_restarts
    cfg.tol_f, cfg.tol_df, cfg.tol_dx = tol_f, tol_df, tol_dx
    cfg.linear_weight[:] = [float(v) for v in linear_weight]
    cfg.angular_weight[:] = [float(v) for v in angular_weight]
    return cfg


# ---- thin numpy wrappers -------------------------------------------------

def so3_log(q):
    q = _f64(q); out = np.zeros(3)
    lib().ok_so3_log(_dp(q), _dp(out))
    return out


def so3_right_jacobian(w):
    """Returns the 3x3 matrix (row, col)."""
    w = _f64(w); out = np.zeros(9)
    lib().ok_so3_right_jacobian(_dp(w), _dp(out))
    return out.reshape(3, 3).T


def se3_log(t, q):
    p = Pose.make(t, q); out = np.zeros(6)
    lib().ok_se3_log(C.byref(p), _dp(out))
    return out


def se3_right_jacobian(t, q):
    p = Pose.make(t, q); out = np.zeros(36)
    lib().ok_se3_right_jacobian(C.byref(p), _dp(out))
    return out.reshape(6, 6).T


def quat_from_matrix(R, iterative: bool):
    """3x3 rotation (row, col) -> [i, j, k, w] by the reference's C-binding (iterative) or Python-binding reading."""
    m = _f64(np.asarray(R, dtype=np.float64).T).ravel()  # column-major
    q = np.zeros(4)
    lib().ok_quat_from_matrix(_dp(m), int(bool(iterative)), _dp(q))
    return q


def fk(chain: Chain, q, ee_offset=None):
    """Returns (joint_tfms[J,7], ee[7]) with rows [t, quat(i,j,k,w)]."""
    q = _f64(q)
    ee_off = ee_offset if ee_offset is not None else Pose.make()
    jt = (Pose * MAX_JOINTS)()
    ee = Pose()
    lib().ok_fk(C.byref(chain), _dp(q), C.byref(ee_off), jt, C.byref(ee))
    return np.array([jt[j].as7() for j in range(chain.n_joints)]), ee.as7()


def joint_jacobian(chain: Chain, q, ee_offset=None):
    """6 x n body-frame Jacobian at q."""
    q = _f64(q)
    ee_off = ee_offset if ee_offset is not None else Pose.make()
    jt = (Pose * MAX_JOINTS)()
    ee = Pose()
    lib().ok_fk(C.byref(chain), _dp(q), C.byref(ee_off), jt, C.byref(ee))
    out = np.zeros(6 * chain.n_pos)
    lib().ok_joint_jacobian(C.byref(chain), jt, C.byref(ee), _dp(out))
    return out.reshape(chain.n_pos, 6).T


def eval_fg(chain: Chain, target7, q, w_lin=(1, 1, 1), w_ang=(1, 1, 1), ee_offset=None,
            grad=True):
    q = _f64(q)
    tgt = Pose.make(target7[:3], target7[3:])
    ee_off = ee_offset if ee_offset is not None else Pose.make()
    wl, wa = _f64(w_lin), _f64(w_ang)
    g = np.zeros(chain.n_pos)
    f = lib().ok_eval(C.byref(chain), C.byref(tgt), C.byref(ee_off), _dp(wl), _dp(wa), _dp(q),
                      _dp(g) if grad else None)
    return (f, g) if grad else f


class range_rule:
    """``with range_rule(RANGE_NEW_INCLUSIVE): ...`` -- process-wide rule switch, restored on exit."""

    def __init__(self, rule):
        self.rule = rule

    def __enter__(self):
        self.prev = lib().ok_get_range_rule()
        lib().ok_set_range_rule(self.rule)
        return self

    def __exit__(self, *exc):
        lib().ok_set_range_rule(self.prev)
        return False


def restart_seed(chain: Chain, index: int):
    out = np.zeros(chain.n_pos)
    lib().ok_restart_seed(C.byref(chain), index, _dp(out))
    return out


def solve_restart(chain: Chain, cfg: Config, target7, x0, index: int, ee_offset=None,
                  trace_cap: int = 0):
    tgt = Pose.make(target7[:3], target7[3:])
    ee_off = ee_offset if ee_offset is not None else Pose.make()
    x0 = _f64(x0)
    res = RestartResult()
    n = chain.n_pos
    if trace_cap:
        trace = np.zeros((trace_cap, n + 1))
        tl = C.c_int(0)
        lib().ok_solve_restart(C.byref(chain), C.byref(cfg), C.byref(tgt), C.byref(ee_off),
                               _dp(x0), index, C.byref(res), _dp(trace), trace_cap, C.byref(tl))
        return res, trace[:tl.value]
    lib().ok_solve_restart(C.byref(chain), C.byref(cfg), C.byref(tgt), C.byref(ee_off), _dp(x0),
                           index, C.byref(res), None, 0, None)
    return res


def ik(chain: Chain, cfg: Config, target7, x0, restart_begin: int, restart_end: int,
       n_threads: int = 1, early_exit: bool = True, per_restart: bool = False, ee_offset=None):
    """Returns dict(found, winner, x, f, n_run[, status, success, xs, fs, evals])."""
    tgt = Pose.make(target7[:3], target7[3:])
    ee_off = ee_offset if ee_offset is not None else Pose.make()
    x0 = _f64(x0)
    n = chain.n_pos
    cnt = restart_end - restart_begin
    winner = C.c_uint64(0)
    n_run = C.c_uint64(0)
    x = np.zeros(n)
    f = C.c_double(0.0)
    pr = (RestartResult * cnt)() if per_restart else None
    # (early_exit: False / True = the deterministic reading; "find_any" or 2 = the reference's multi-thread rule)
    ee_mode = 2 if early_exit in (2, "find_any") else int(bool(early_exit))
    found = lib().ok_ik(C.byref(chain), C.byref(cfg), C.byref(tgt), C.byref(ee_off), _dp(x0),
                        restart_begin, restart_end, n_threads, ee_mode,
                        C.byref(winner), _dp(x), C.byref(f), pr, C.byref(n_run))
    out = dict(found=bool(found), winner=winner.value, x=x, f=f.value, n_run=n_run.value)
    if per_restart:
        raw = np.frombuffer(pr, dtype=np.dtype([
            ("result", "<i4"), ("success", "<i4"), ("n_evals", "<i4"), ("n_iters", "<i4"),
            ("f", "<f8"), ("x", "<f8", (MAX_DOF,))]))
        out.update(status=raw["result"].copy(), success=raw["success"].copy(),
                   evals=raw["n_evals"].copy(), iters=raw["n_iters"].copy(),
                   fs=raw["f"].copy(), xs=raw["x"][:, :n].copy())
    return out


def ik_many(chain: Chain, cfg: Config, targets7, x0s, n_restarts: int, n_threads: int):
    """T independent ik() calls in C (one thread per call, targets handed out to n_threads): found [T] (bool), xs [T, n]."""
    targets7 = _f64(targets7)
    x0s = _f64(x0s)
    T, n = targets7.shape[0], chain.n_pos
    tg = (Pose * T)()
    for t in range(T):
        tg[t].t[:] = targets7[t, :3].tolist()
        tg[t].q[:] = targets7[t, 3:].tolist()
    ee = Pose.make()
    found = np.zeros(T, dtype=np.int32)
    xs = np.zeros((T, n))
    L = lib()
    L.ok_ik_many.argtypes = [C.POINTER(Chain), C.POINTER(Config), C.POINTER(Pose), C.POINTER(Pose), C.POINTER(C.c_double),
                             C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    t0 = __import__("time").perf_counter()
    L.ok_ik_many(C.byref(chain), C.byref(cfg), tg, C.byref(ee), _dp(x0s), T, int(n_restarts), int(n_threads),
                 found.ctypes.data_as(C.POINTER(C.c_int32)), _dp(xs))
    dt = __import__("time").perf_counter() - t0
    return found != 0, xs, dt


def pool_start(n_threads: int) -> int:
    """Persistent worker threads for ik(..., n_threads == n) (rayon's pool); pool_stop() / a build switch ends them."""
    return int(lib().ok_pool_start(int(n_threads)))


def pool_stop():
    if _lib is not None:
        _lib.ok_pool_stop()


def lsq_direction(l_packed, g, lo, hi):
    l_packed, g, lo, hi = _f64(l_packed), _f64(g), _f64(lo), _f64(hi)
    n = len(g)
    s = np.zeros(n)
    mode = lib().ok_lsq_direction(n, _dp(l_packed), _dp(g), _dp(lo), _dp(hi), _dp(s))
    return mode, s
