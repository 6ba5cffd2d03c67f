"""Import shim: `from optik import Robot, SolverConfig` (what the reference's Python scripts and its
type stub /root/reference/optik.pyi use) resolves to the MI355X implementation when this
repository is on PYTHONPATH ahead of -- or instead of -- the reference's compiled module."""
from optik_amd import Robot, SolverConfig  # noqa: F401

__all__ = ["Robot", "SolverConfig"]
