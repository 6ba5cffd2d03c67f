"""Device-buffer front end over the kernel-layer C ABI (``include/optik_hip.h``).

PyTorch is used only as plumbing: HBM allocation, streams, and (in ``bench.py``)
``torch.distributed``.  All arithmetic happens in the HIP kernels of
``csrc/ik_capi.hip``; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native as nat


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class HipChain:
    """A flat kinematic chain uploaded to the GPU (optik_hip_chain)."""

    def __init__(self, types, origins, axes, lb, ub, device="cuda:0"):
        if not torch.cuda.is_available():
            raise nat.OptikHipError("no GPU visible to torch; optik_amd has no CPU fallback")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        types = np.ascontiguousarray(types, dtype=np.int32)
        origins = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 7)
        axes = np.ascontiguousarray(axes, dtype=np.float64).reshape(-1, 3)
        lb = np.ascontiguousarray(lb, dtype=np.float64)
        ub = np.ascontiguousarray(ub, dtype=np.float64)
        self.n = int(len(lb))
        self.lb, self.ub = lb, ub
        self._h = C.c_void_p()
        nat.check(nat.lib().optik_hip_chain_create(
            _dp(origins), _dp(axes), types.ctypes.data_as(C.POINTER(C.c_int32)), len(types),
            _dp(lb), _dp(ub), self.n, C.byref(self._h)))

    def close(self):
        if self._h:
            nat.lib().optik_hip_chain_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_range_rule(self, rule):
        """Which rand 0.9.2 code path the restart seeds follow (nat.RANGE_*; optik_hip.h)."""
        nat.check(nat.lib().optik_hip_chain_set_range_rule(self._h, int(rule)))

    # -- batched primitives ----------------------------------------------------
    def eval_batch(self, q, target7, cfg=None, ee_offset7=None, grad=True):
        """q: [n, B] float64 cuda tensor -> (f [B], g [n, B])."""
        cfg = cfg or nat.make_config()
        assert q.is_cuda and q.dtype == torch.float64 and q.shape[0] == self.n and q.is_contiguous()
        B = q.shape[1]
        f = torch.empty(B, dtype=torch.float64, device=q.device)
        g = torch.empty_like(q) if grad else None
        t7 = np.ascontiguousarray(target7, dtype=np.float64)
        ee = np.ascontiguousarray(ee_offset7, dtype=np.float64) if ee_offset7 is not None else None
        nat.check(nat.lib().optik_hip_eval_batch(self._h, C.byref(cfg), _dp(t7),
                                                 _dp(ee) if ee is not None else None, _ptr(q), B,
                                                 _ptr(f), _ptr(g), _stream_ptr()))
        return f, g

    def fk_batch(self, q, ee_offset7=None, jacobian=False):
        """q: [n, B] -> pose [7, B] (and jac [6n, B], column-major 6 x n per column)."""
        assert q.is_cuda and q.dtype == torch.float64 and q.shape[0] == self.n and q.is_contiguous()
        B = q.shape[1]
        pose = torch.empty((7, B), dtype=torch.float64, device=q.device)
        jac = torch.empty((6 * self.n, B), dtype=torch.float64, device=q.device) if jacobian else None
        ee = np.ascontiguousarray(ee_offset7, dtype=np.float64) if ee_offset7 is not None else None
        nat.check(nat.lib().optik_hip_fk_batch(self._h, _dp(ee) if ee is not None else None, _ptr(q),
                                               B, _ptr(pose), _ptr(jac), _stream_ptr()))
        return (pose, jac) if jacobian else pose

    def seed_batch(self, first, count):
        q = torch.empty((self.n, count), dtype=torch.float64, device=self.device)
        nat.check(nat.lib().optik_hip_seed_batch(self._h, int(first), int(count), _ptr(q), _stream_ptr()))
        return q

    # -- the hot path -------------------------------------------------------------
    def alloc_ik_buffers(self, T, R, per_restart=True):
        dev = self.device
        bufs = dict(
            win_x=torch.empty((T, self.n), dtype=torch.float64, device=dev),
            win_f=torch.empty(T, dtype=torch.float64, device=dev),
            win_idx=torch.empty(T, dtype=torch.int64, device=dev),
            win_key=torch.empty(T, dtype=torch.float64, device=dev))
        if per_restart:
            bufs.update(
                x=torch.empty((self.n, T * R), dtype=torch.float64, device=dev),
                f=torch.empty(T * R, dtype=torch.float64, device=dev),
                status=torch.empty(T * R, dtype=torch.int32, device=dev),
                evals=torch.empty(T * R, dtype=torch.int32, device=dev))
        return bufs

    def ik_batch(self, cfg, targets, x0, restart_begin, restart_end, flags=0, deadline_s=0.0,
                 ee_offset7=None, bufs=None, per_restart=True):
        """targets [T, 7], x0 [T, n] float64 cuda tensors.  Stream-ordered; returns the
        buffer dict (win_idx is int64 with -1 = UINT64_MAX = no solution)."""
        assert targets.is_cuda and targets.dtype == torch.float64 and targets.is_contiguous()
        assert x0.is_cuda and x0.dtype == torch.float64 and x0.is_contiguous()
        T = targets.shape[0]
        assert targets.shape == (T, 7) and x0.shape == (T, self.n)
        R = int(restart_end - restart_begin)
        if bufs is None:
            bufs = self.alloc_ik_buffers(T, R, per_restart)
        o = nat.IkOutputs()
        o.d_x, o.d_f = _ptr(bufs.get("x")), _ptr(bufs.get("f"))
        o.d_status, o.d_evals = _ptr(bufs.get("status")), _ptr(bufs.get("evals"))
        o.d_win_x, o.d_win_f = _ptr(bufs["win_x"]), _ptr(bufs["win_f"])
        o.d_win_idx, o.d_win_key = _ptr(bufs["win_idx"]), _ptr(bufs["win_key"])
        ee = np.ascontiguousarray(ee_offset7, dtype=np.float64) if ee_offset7 is not None else None
        nat.check(nat.lib().optik_hip_ik_batch(
            self._h, C.byref(cfg), _ptr(targets), _ptr(x0), T, _dp(ee) if ee is not None else None,
            int(restart_begin), int(restart_end), int(flags), float(deadline_s), C.byref(o),
            _stream_ptr()))
        return bufs

    def ik_host(self, cfg, targets, x0, restart_begin, restart_end, flags=0, deadline_s=0.0, ee_offset7=None):
        """optik_hip_ik_host: host arrays in (targets [T, 7], x0 [T, n]), the winners back as numpy arrays; blocking.
        With IK_EARLY_EXIT | IK_FIND_ANY and one target the call returns when the first restart has succeeded."""
        targets = np.ascontiguousarray(targets, dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        T = targets.shape[0]
        assert targets.shape == (T, 7) and x0.shape == (T, self.n)
        win_x = np.zeros((T, self.n)); win_f = np.zeros(T); win_key = np.zeros(T)
        win_idx = np.zeros(T, dtype=np.uint64)
        ee = np.ascontiguousarray(ee_offset7, dtype=np.float64) if ee_offset7 is not None else None
        nat.check(nat.lib().optik_hip_ik_host(
            self._h, C.byref(cfg), _dp(targets), _dp(x0), T, _dp(ee) if ee is not None else None,
            int(restart_begin), int(restart_end), int(flags), float(deadline_s), _dp(win_x), _dp(win_f),
            win_idx.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(win_key)))
        return dict(win_x=win_x, win_f=win_f, win_idx=win_idx.astype(np.int64), win_key=win_key)

    def set_timing(self, enabled=True):
        nat.lib().optik_hip_set_timing(self._h, 1 if enabled else 0)

    def timing_mean(self):
        """(mean solve-kernel ms, launches) since set_timing(True); HIP events on the launch stream."""
        ms, cnt = C.c_double(0.0), C.c_int32(0)
        nat.check(nat.lib().optik_hip_timing_mean(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def last_launch(self):
        info = nat.LaunchInfo()
        nat.check(nat.lib().optik_hip_last_launch(self._h, C.byref(info)))
        return dict(grid=info.grid, block=info.block, lds_bytes=info.lds_bytes, tiles=info.tiles,
                    kernel_ms=info.kernel_ms)


def probe(op, a, b=None):
    """Elementary device functions (test hook).  numpy in, numpy out."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b if b is not None else a, dtype=np.float64)
    out = np.empty_like(a)
    nat.check(nat.lib().optik_hip_probe(int(op), _dp(a), _dp(b), a.size, _dp(out)))
    return out


_MATH_STRIDE = {0: 3, 1: 9, 2: 6, 3: 36}


def probe_math(op, poses7):
    """math.rs functions on the device (test hook): poses7 [count, 7] = t[3], quat[i,j,k,w].
    op 0 so3::log -> [count, 3]; 1 so3::right_jacobian(so3::log(q)) -> [count, 3, 3]; 2 se3::log ->
    [count, 6]; 3 se3::right_jacobian -> [count, 6, 6] (row-major matrices)."""
    p = np.ascontiguousarray(poses7, dtype=np.float64).reshape(-1, 7)
    out = np.empty((p.shape[0], _MATH_STRIDE[int(op)]), dtype=np.float64)
    nat.check(nat.lib().optik_hip_probe_math(int(op), _dp(p), p.shape[0], _dp(out)))
    if op == 1:
        return out.reshape(-1, 3, 3)
    if op == 3:
        return out.reshape(-1, 6, 6)
    return out
