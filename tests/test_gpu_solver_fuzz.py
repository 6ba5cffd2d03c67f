"""-m gpu: a short run of tools/solver_fuzz.py -- random robots / targets / tolerances / weights / ee offsets / restart
ranges / early-exit rules on the solver a launch gets, the quad solver forced and the lane-per-restart form forced,
every restart against the CPU oracle bit for bit.  (Long runs: python tools/solver_fuzz.py 300 <seed>.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_solver_fuzz(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "solver_fuzz.py"), "25", str(seed)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "fuzz ok" in r.stdout
