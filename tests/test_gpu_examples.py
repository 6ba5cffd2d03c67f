"""-m gpu: the scripts under examples/ run as a user would run them."""
import os
import subprocess
import sys

import pytest

from conftest import ROBOTS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PANDA = [os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link8"]


def _run(script, *args):
    env = dict(os.environ, PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), *PANDA, *args], env=env,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-2000:]
    return res.stdout


def test_single_ik_example():
    out = _run("single_ik.py", "50")
    solved = int(out.split()[0])
    assert solved >= 48, out


def test_many_targets_example():
    out = _run("many_targets.py", "4096")
    assert int(out.split()[0]) == 4096, out


def test_diff_ik_example():
    out = _run("diff_ik.py")
    assert out.startswith("alpha = "), out
    resid = float(out.split("|J v - alpha V| = ")[1].split(";")[0])
    assert resid < 1e-9, out
