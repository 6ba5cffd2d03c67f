"""optik_amd -- MI355X-native batched random-restart IK (the hot path of kylc/optik).

``optik_amd.Robot`` / ``optik_amd.SolverConfig`` mirror the reference's Python module
(/root/reference/optik.pyi); ``optik_amd.device.HipChain`` exposes the kernel layer on
device buffers.  Everything computes in hand-written HIP kernels (csrc/); there is no
CPU fallback.
"""
from ._native import OptikHipError  # noqa: F401
from .robot import Robot, SolverConfig  # noqa: F401

__all__ = ["OptikHipError", "Robot", "SolverConfig"]
