"""Python front end with the surface of the reference's `optik` module
(/root/reference/optik.pyi:9-49; implementation crates/optik-py/src/lib.rs:17-155),
bound with ctypes to the `optik_robot_*` C ABI of liboptik_amd.so (include/optik.h).

Poses are 4x4 homogeneous matrices given as nested lists / arrays in row-major
order (optik-py/src/lib.rs:8-15); results come back as Python lists, as in the
reference (`diff_ik` included: a small LP solved exactly on the host).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat

U64_MAX = 0xFFFFFFFFFFFFFFFF


def _bind(L):
    if getattr(L, "_robot_bound", False):
        return L
    vp, dp = C.c_void_p, C.POINTER(C.c_double)
    L.optik_robot_try_from_urdf_str.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.optik_robot_last_error.restype = C.c_char_p
    L.optik_robot_free.argtypes = [vp]
    L.optik_robot_set_parallelism.argtypes = [vp, C.c_uint]
    L.optik_robot_num_positions.argtypes = [vp]
    L.optik_robot_num_positions.restype = C.c_uint
    L.optik_robot_ik_ex.argtypes = [vp, C.POINTER(nat.SolverConfigC), dp, dp, dp, dp, dp,
                                    C.POINTER(C.c_uint64)]
    L.optik_robot_ik_batch_ex.argtypes = [vp, C.POINTER(nat.SolverConfigC), C.c_int32, dp, dp, dp, dp, dp,
                                          C.POINTER(C.c_int32)]
    L.optik_robot_ik_batch_poses.argtypes = [vp, C.POINTER(nat.SolverConfigC), C.c_int32, dp, C.c_uint32, dp, dp,
                                             dp, dp, C.POINTER(C.c_int32)]
    L.optik_robot_fk_ex.argtypes = [vp, dp, dp, dp]
    L.optik_robot_diff_ik_ex.argtypes = [vp, dp, dp, dp, dp, C.POINTER(C.c_double), dp]
    L.optik_robot_joint_jacobian_ex.argtypes = [vp, dp, dp, dp]
    L.optik_robot_set_devices.argtypes = [vp, C.POINTER(C.c_int32), C.c_int32]
    L.optik_robot_num_devices.argtypes = [vp]
    L.optik_robot_last_parts.argtypes = [vp]
    L.optik_robot_last_parts.restype = C.c_int32
    L.optik_robot_chain_tables.argtypes = [vp, C.POINTER(C.c_int32), dp, dp, C.POINTER(C.c_int32)]
    L.optik_robot_chain_tables_n.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), dp, dp, C.POINTER(C.c_int32)]
    L.optik_robot_hip_chain.argtypes = [vp]
    L.optik_robot_hip_chain.restype = vp
    L.optik_robot_joint_limits.argtypes = [vp]
    L.optik_robot_joint_limits.restype = dp
    L._robot_bound = True
    return L


BATCH_ROW_MAJOR, BATCH_VALIDATE_POSES = 1, 2  # include/optik.h: OPTIK_BATCH_*


def _err(L):
    m = L.optik_robot_last_error()
    return m.decode() if m else "unknown error"


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pose16(m):
    """Row-major nested 4x4 -> column-major flat (parse_pose, optik-py/src/lib.rs:8-15).

    parse_pose converts with nalgebra's ``try_convert::<Matrix4, Isometry3>`` and panics with
    "invalid target transform specified" unless the matrix is an isometry: bottom row exactly
    (0, 0, 0, 1) and the 3x3 block special-orthogonal -- R^T R equal to the identity within
    100 * f64::EPSILON per entry (``is_special_orthogonal``) and det R > 0.  Host-side input
    validation, the same message."""
    a = np.asarray(m, dtype=np.float64)
    if a.shape != (4, 4):
        raise ValueError("pose must be a 4x4 homogeneous matrix")
    R = a[:3, :3]
    eps = 100.0 * np.finfo(np.float64).eps
    ok = (a[3, 0] == 0.0 and a[3, 1] == 0.0 and a[3, 2] == 0.0 and a[3, 3] == 1.0
          and np.all(np.abs(R.T @ R - np.eye(3)) <= eps) and np.linalg.det(R) > 0.0)
    if not ok:
        raise ValueError("invalid target transform specified")
    return np.ascontiguousarray(a.T).ravel()


class SolverConfig:
    """optik.pyi:9-20; defaults of optik-py/src/lib.rs:24-31 / config.rs:52-65."""

    def __init__(self, solution_mode="speed", max_time=0.1, max_restarts=U64_MAX, tol_f=1e-6,
                 tol_df=-1.0, tol_dx=-1.0, linear_weight=(1.0, 1.0, 1.0),
                 angular_weight=(1.0, 1.0, 1.0)):
        if solution_mode not in ("speed", "quality"):
            raise ValueError("solution_mode must be 'speed' or 'quality'")  # config.rs:10-20
        if max_time == 0.0 and max_restarts == 0:
            # optik-py/src/lib.rs:45-47
            raise ValueError("no time or restart limit applied (solver would run forever)")
        self.solution_mode = solution_mode
        self.max_time = float(max_time)
        self.max_restarts = int(max_restarts)
        self.tol_f, self.tol_df, self.tol_dx = float(tol_f), float(tol_df), float(tol_dx)
        self.linear_weight = [float(v) for v in linear_weight]
        self.angular_weight = [float(v) for v in angular_weight]

    def to_c(self):
        return nat.make_config(self.solution_mode, self.max_time,
                               0 if self.max_restarts >= U64_MAX else self.max_restarts,
                               self.tol_f, self.tol_df, self.tol_dx, self.linear_weight,
                               self.angular_weight)


class Robot:
    """optik.pyi:22-49."""

    def __init__(self, handle):
        self._L = _bind(nat.lib())
        self._h = handle
        self._hip = {}

    @staticmethod
    def from_urdf_str(urdf: str, base_link: str, ee_link: str) -> "Robot":
        L = _bind(nat.lib())
        h = C.c_void_p()
        rc = L.optik_robot_try_from_urdf_str(urdf.encode(), base_link.encode(), ee_link.encode(),
                                             C.byref(h))
        if rc:
            raise RuntimeError(_err(L))
        return Robot(h)

    @staticmethod
    def from_urdf_file(path: str, base_link: str, ee_link: str) -> "Robot":
        try:
            with open(path) as fh:
                text = fh.read()
        except OSError as e:
            raise RuntimeError("error parsing URDF file!") from e  # lib.rs:55
        return Robot.from_urdf_str(text, base_link, ee_link)

    def __del__(self):
        try:
            if self._h:
                self._L.optik_robot_free(self._h)
                self._h = None
        except Exception:
            pass

    def set_parallelism(self, n: int) -> None:
        """lib.rs:66-72.  The GPU needs no thread count; what n keeps from the reference is Speed
        mode's early-exit rule.  n = 1 returns the lowest successful restart -- the reference's
        deterministic 1-thread answer (its own determinism test sets 1, tests/test_ik.rs:45-89).
        Never calling this (the reference's default pool has every core) or n > 1 stops at the
        first success of any restart, like rayon's find_any with several threads (README.md:17,
        96): lower latency, a valid but timing-dependent choice among the solutions."""
        self._L.optik_robot_set_parallelism(self._h, int(n))

    def set_devices(self, device_ids) -> None:
        """GPUs of this node the robot spreads restart ranges (ik) and targets (ik_batch) over
        (extension; include/optik.h: optik_robot_set_devices).  Before the first GPU call."""
        ids = (C.c_int32 * len(device_ids))(*[int(d) for d in device_ids])
        if self._L.optik_robot_set_devices(self._h, ids, len(device_ids)):
            raise RuntimeError(_err(self._L))

    def num_devices(self) -> int:
        return int(self._L.optik_robot_num_devices(self._h))

    def last_parts(self) -> int:
        """Over how many of the robot's devices the last ik / ik_batch call was actually cut."""
        return int(self._L.optik_robot_last_parts(self._h))

    def num_positions(self) -> int:
        return int(self._L.optik_robot_num_positions(self._h))

    def joint_limits(self):
        n = self.num_positions()
        p = self._L.optik_robot_joint_limits(self._h)
        vals = [p[i] for i in range(2 * n)]
        C.CDLL(None).free(p)
        return vals[:n], vals[n:]

    def _check_x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        if x.size != self.num_positions():
            # kinematics.rs:129-133
            raise ValueError("generalized position vector `q` is of incorrect length")
        return x

    def fk(self, x, ee_offset=None):
        x = self._check_x(x)
        ee = _pose16(ee_offset) if ee_offset is not None else None
        out = np.zeros(16)
        if self._L.optik_robot_fk_ex(self._h, _dp(x), _dp(ee) if ee is not None else None, _dp(out)):
            raise RuntimeError(_err(self._L))
        return out.reshape(4, 4).T.tolist()

    def joint_jacobian(self, x, ee_offset=None):
        x = self._check_x(x)
        ee = _pose16(ee_offset) if ee_offset is not None else None
        n = self.num_positions()
        out = np.zeros(6 * n)
        if self._L.optik_robot_joint_jacobian_ex(self._h, _dp(x), _dp(ee) if ee is not None else None,
                                                 _dp(out)):
            raise RuntimeError(_err(self._L))
        return out.reshape(n, 6).T.tolist()

    def ik(self, config: SolverConfig, target, x0, ee_offset=None, return_index=False):
        """Returns (x, c) or None (optik.pyi:36-42).  `return_index=True` appends the winning
        restart index (an extension used by the parity tests)."""
        x0 = self._check_x(x0)
        tgt = _pose16(target)
        ee = _pose16(ee_offset) if ee_offset is not None else None
        cfg = config.to_c()
        n = self.num_positions()
        x, f, idx = np.zeros(n), C.c_double(0.0), C.c_uint64(0)
        rc = self._L.optik_robot_ik_ex(self._h, C.byref(cfg), _dp(tgt), _dp(x0),
                                       _dp(ee) if ee is not None else None, _dp(x), C.byref(f),
                                       C.byref(idx))
        if rc < 0:
            raise RuntimeError(_err(self._L))  # e.g. "seed joint position outside of joint limits"
        if rc == 1:
            return None
        return (x.tolist(), f.value, idx.value) if return_index else (x.tolist(), f.value)

    def ik_batch_arrays(self, config: SolverConfig, targets, x0s, ee_offset=None):
        """Many ik() calls at once (extension), array form: `targets` [T, 4, 4] row-major poses,
        `x0s` [T, n] seeds -> (x [T, n], c [T], found [T] bool); rows with found False are zero.
        Each target gets the semantics of ik() with the same config (poses validated the same way)."""
        tg = np.asarray(targets, dtype=np.float64)
        if tg.ndim != 3 or tg.shape[1:] != (4, 4):
            raise ValueError("targets must be [T, 4, 4]")
        T = tg.shape[0]
        n = self.num_positions()
        x0s = np.ascontiguousarray(x0s, dtype=np.float64).reshape(T, n)
        tg16 = np.ascontiguousarray(tg).reshape(T, 16)  # row-major as given; the host layer transposes
        ee = _pose16(ee_offset) if ee_offset is not None else None
        cfg = config.to_c()
        x = np.zeros((T, n))
        f = np.zeros(T)
        found = np.zeros(T, dtype=np.int32)
        # BATCH_VALIDATE_POSES: parse_pose's isometry test (see _pose16) on every target, in C++
        rc = self._L.optik_robot_ik_batch_poses(self._h, C.byref(cfg), T, _dp(tg16),
                                                BATCH_ROW_MAJOR | BATCH_VALIDATE_POSES, _dp(x0s),
                                                _dp(ee) if ee is not None else None, _dp(x), _dp(f),
                                                found.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc == -3:
            raise ValueError(_err(self._L))
        if rc < 0:
            raise RuntimeError(_err(self._L))
        return x, f, found.astype(bool)

    def ik_batch(self, config: SolverConfig, targets, x0s, ee_offset=None):
        """ik_batch_arrays as a list: (x, c) or None per target, like T calls of ik()."""
        x, f, found = self.ik_batch_arrays(config, targets, x0s, ee_offset)
        # Building ~10 Python objects per target trips the cyclic collector every few hundred targets, and
        # each of its passes walks what has been built so far: 65 536 targets take 105 ms instead of 41.
        # For large batches the collector is paused while the list is built (none of these objects can be
        # garbage).  Its state is process-global: a thread that changes it at the same moment may find it
        # re-enabled -- callers for whom that matters (or who want arrays anyway) use ik_batch_arrays.
        import gc
        pause = len(found) >= 2048 and gc.isenabled()
        if pause:
            gc.disable()
        try:
            xs, fs = x.tolist(), f.tolist()
            return [(xs[t], fs[t]) if ok else None for t, ok in enumerate(found.tolist())]
        finally:
            if pause:
                gc.enable()

    def diff_ik(self, x0, V_WE, v_max, ee_offset=None):
        """Returns (alpha, v) or None (optik.pyi:43-49; lib.rs:123-239): the joint velocities
        realising alpha * V_WE for the largest feasible 0 <= alpha <= 1 under |v_i| <= v_max_i."""
        x0 = self._check_x(x0)
        n = self.num_positions()
        V = np.ascontiguousarray(V_WE, dtype=np.float64).reshape(6)
        vm = np.ascontiguousarray(v_max, dtype=np.float64).reshape(n)
        ee = _pose16(ee_offset) if ee_offset is not None else None
        alpha, v = C.c_double(0.0), np.zeros(n)
        rc = self._L.optik_robot_diff_ik_ex(self._h, _dp(x0), _dp(V), _dp(vm),
                                            _dp(ee) if ee is not None else None, C.byref(alpha), _dp(v))
        if rc < 0:
            raise RuntimeError(_err(self._L))
        if rc == 1:
            return None
        return alpha.value, v.tolist()

    # -- extensions ---------------------------------------------------------------
    def chain_tables(self):
        """Flat chain (types, origins[J,7], axes[J,3], lb, ub) as loaded by the C++ URDF loader."""
        nj = C.c_int32(0)
        # first call: joint count only (NULL buffers), then buffers of exactly that size
        if self._L.optik_robot_chain_tables_n(self._h, 0, C.byref(nj), None, None, None):
            raise RuntimeError(_err(self._L))
        origins, axes = np.zeros(nj.value * 7), np.zeros(nj.value * 3)
        types = np.zeros(nj.value, dtype=np.int32)
        if self._L.optik_robot_chain_tables_n(self._h, nj.value, C.byref(nj), _dp(origins), _dp(axes),
                                              types.ctypes.data_as(C.POINTER(C.c_int32))):
            raise RuntimeError(_err(self._L))
        J = nj.value
        lb, ub = self.joint_limits()
        return dict(types=types[:J].copy(), origins=origins[:J * 7].reshape(J, 7).copy(),
                    axes=axes[:J * 3].reshape(J, 3).copy(), lb=np.array(lb), ub=np.array(ub))

    def hip_chain(self, device="cuda:0"):
        """Device-buffer interface (optik_amd.device.HipChain) for this robot."""
        from .device import HipChain
        key = str(device)
        if key not in self._hip:
            self._hip[key] = HipChain(device=device, **self.chain_tables())
        return self._hip[key]
