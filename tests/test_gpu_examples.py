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


def test_reference_style_script_runs_unchanged(tmp_path):
    """A script written against the reference's Python module (`from optik import Robot,
    SolverConfig`; optik.pyi) runs as it is with this repository on PYTHONPATH."""
    script = tmp_path / "user_script.py"
    script.write_text(
        "import sys\n"
        "import numpy as np\n"
        "from optik import Robot, SolverConfig\n"
        "robot = Robot.from_urdf_file(*sys.argv[1:4])\n"
        "robot.set_parallelism(1)\n"
        "config = SolverConfig()\n"
        "rng = np.random.default_rng(0)\n"
        "ok = 0\n"
        "for _ in range(20):\n"
        "    x0 = rng.uniform(*robot.joint_limits())\n"
        "    q = rng.uniform(*robot.joint_limits())\n"
        "    target = np.array(robot.fk(q))\n"
        "    sol = robot.ik(config, target, x0)\n"
        "    if sol is not None:\n"
        "        q_opt, c = sol\n"
        "        assert c < 1e-6 and np.allclose(np.array(robot.fk(q_opt)), target, atol=2e-3)\n"
        "        ok += 1\n"
        "print(ok)\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, str(script), *PANDA], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-500:] + res.stderr[-2000:]
    assert int(res.stdout.split()[-1]) >= 19


@pytest.mark.parametrize("spec", [PANDA, [os.path.join(ROOT, "tests", "golden", "robots", "arm10.urdf"), "l0", "l11"]])
def test_c_api_example(tmp_path, spec):
    """examples/c_api.c: the reference's C ABI (include/optik.h) from plain C11 -- URDF in, ik() on the GPU, the
    answer checked with fk(), every returned buffer freed with free() -- on Panda and on a 10-joint chain."""
    libdir = os.path.join(ROOT, "optik_amd", "csrc")
    exe = tmp_path / "c_api"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_api.c"), "-L", libdir, "-loptik_amd",
                           "-Wl,-rpath," + libdir, "-lm", "-o", str(exe)])
    res = subprocess.run([str(exe), *spec], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-500:] + res.stderr[-2000:]
    assert res.stdout.startswith("solved "), res.stdout
