import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_GOLDEN = os.path.join(GOLDEN, "reference")
ROBOTS = os.path.join(ROOT, "optik_amd", "robots")
TEST_ROBOTS = os.path.join(GOLDEN, "robots")  # synthetic chains written for the tests


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    return binding


def _load_chain(path, base, ee):
    from oracle import urdf_chain
    with open(path) as fh:
        return urdf_chain.chain_from_urdf(fh.read(), base, ee)


ROBOT_SPECS = {
    "ur3e": (os.path.join(REF_GOLDEN, "ur3e.urdf"), "ur_base_link", "ur_ee_link"),
    "panda": (os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link8"),
    "panda_hand": (os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_hand"),
    "ur10": (os.path.join(ROBOTS, "ur10.urdf"), "base_link", "ee_link"),
    # sub-chains of the Panda: 2..5 revolute joints, no trailing fixed joint
    "panda2": (os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link2"),
    "panda3": (os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link3"),
    "panda4": (os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link4"),
    "panda5": (os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link5"),
    # the ends of the supported range: 1 joint, and 8 joints + trailing fixed joint
    "panda1": (os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link1"),
    "arm8": (os.path.join(TEST_ROBOTS, "arm8.urdf"), "l0", "l9"),
    # beyond the tuned solvers' range: 9 .. 16 joints on the general kernels (ik_wide.hpp); the reference
    # has no limit (kinematics.rs:107-110).  Written by tools/gen_wide_robots.py
    "arm9": (os.path.join(TEST_ROBOTS, "arm9.urdf"), "l0", "l9"),
    "arm10": (os.path.join(TEST_ROBOTS, "arm10.urdf"), "l0", "l11"),
    "arm12": (os.path.join(TEST_ROBOTS, "arm12.urdf"), "l0", "l12"),
    "arm16": (os.path.join(TEST_ROBOTS, "arm16.urdf"), "l0", "l17"),
    # prismatic joints: forward kinematics only (kinematics.rs:185, 243-255)
    "gantry": (os.path.join(TEST_ROBOTS, "gantry.urdf"), "g0", "g5"),
}


@pytest.fixture(scope="session")
def chains(oracle):
    """name -> (flat chain dict from the Python URDF oracle, ok_chain for the C oracle)."""
    out = {}
    for name, (path, base, ee) in ROBOT_SPECS.items():
        d = _load_chain(path, base, ee)
        out[name] = (d, oracle.make_chain(**d))
    return out
