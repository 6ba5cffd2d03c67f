"""CPU-side checks of the product's host layer (no compute calls): the C-ABI library
loads and exports every symbol include/*.h declares, the C++ URDF loader reproduces the
oracle's restatement of kinematics.rs:18-105, and host-side argument validation mirrors
the reference's panics."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import REF_GOLDEN, ROBOT_SPECS, ROOT


@pytest.fixture(scope="module")
def built():
    from optik_amd import build
    build.build()
    from optik_amd import _native
    return _native.lib()


def _declared_symbols():
    syms = set()
    for hdr in ("optik_hip.h", "optik.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms.update(re.findall(r"\b(optik_(?:hip|robot)_[a-z0-9_]+)\s*\(", text))
    return syms


def test_every_declared_symbol_is_exported(built):
    syms = _declared_symbols()
    assert {"optik_robot_ik", "optik_robot_fk", "optik_robot_from_urdf_str", "optik_hip_ik_batch",
            "optik_hip_eval_batch", "optik_robot_diff_ik"} <= syms
    assert len(syms) >= 30
    for s in sorted(syms):
        assert hasattr(built, s), f"{s} is declared in include/ but not exported"


def test_reference_c_abi_symbol_list(built):
    """The 11 extern "C" functions of crates/optik-cpp/src/lib.rs:26-183."""
    for s in ["optik_robot_from_urdf_file", "optik_robot_from_urdf_str", "optik_robot_free",
              "optik_robot_set_parallelism", "optik_robot_num_positions", "optik_robot_joint_limits",
              "optik_robot_joint_jacobian", "optik_robot_fk", "optik_robot_random_configuration",
              "optik_robot_ik", "optik_robot_diff_ik"]:
        assert hasattr(built, s)


def test_solver_config_layout_matches_csolverconfig():
    from optik_amd import _native as nat
    assert C.sizeof(nat.SolverConfigC) == 96  # #[repr(C)] CSolverConfig, optik-cpp/src/lib.rs:10-20
    offs = {f[0]: getattr(nat.SolverConfigC, f[0]).offset for f in nat.SolverConfigC._fields_}
    assert (offs["solution_mode"], offs["max_time"], offs["max_restarts"], offs["tol_f"], offs["tol_df"],
            offs["tol_dx"], offs["linear_weight"], offs["angular_weight"]) == (0, 8, 16, 24, 32, 40, 48, 72)


@pytest.mark.parametrize("name", sorted(ROBOT_SPECS))
def test_cpp_urdf_loader_matches_oracle_loader(built, chains, name):
    from optik_amd import Robot
    path, base, ee = ROBOT_SPECS[name]
    robot = Robot.from_urdf_file(path, base, ee)
    got, (want, _) = robot.chain_tables(), chains[name]
    assert robot.num_positions() == len(want["lb"])
    assert np.array_equal(got["types"], want["types"])
    for k in ("origins", "axes", "lb", "ub"):
        np.testing.assert_allclose(got[k], want[k], rtol=0, atol=1e-15)
    lb, ub = robot.joint_limits()
    assert lb == list(want["lb"]) and ub == list(want["ub"])


def test_ur3e_chain_shape(built):
    """6 revolute joints + the trailing fixed joint (ur_base_link -> ur_ee_link)."""
    from optik_amd import Robot
    r = Robot.from_urdf_file(os.path.join(REF_GOLDEN, "ur3e.urdf"), "ur_base_link", "ur_ee_link")
    t = r.chain_tables()
    assert list(t["types"]) == [1, 1, 1, 1, 1, 1, 0]
    assert np.allclose(t["lb"], -np.pi) and np.allclose(t["ub"], np.pi)


URDF_TMPL = """<robot name="t"><link name="a"/><link name="b"/><link name="c"/><link name="d"/>
<joint name="j1" type="{t1}"><parent link="a"/><child link="b"/><origin xyz="0 0 1" rpy="0.1 0.2 0.3"/>
<axis xyz="0 0 2"/><limit lower="-1" upper="1"/></joint>
<joint name="j2" type="fixed"><parent link="b"/><child link="c"/><origin xyz="0.5 0 0" rpy="0 0 1.0"/></joint>
<joint name="j3" type="{t3}"><parent link="c"/><child link="d"/><origin xyz="0 0.25 0"/>
<axis xyz="0 1 0"/><limit lower="0" upper="0"/></joint></robot>"""


def test_loader_folds_fixed_joints_and_handles_limits(built):
    from optik_amd import Robot
    from oracle import urdf_chain
    text = URDF_TMPL.format(t1="revolute", t3="revolute")
    r = Robot.from_urdf_str(text, "a", "d")
    got, want = r.chain_tables(), urdf_chain.chain_from_urdf(text, "a", "d")
    assert list(got["types"]) == [1, 1]                    # the fixed joint is folded into j3's origin
    np.testing.assert_allclose(got["origins"], want["origins"], atol=1e-15)
    np.testing.assert_allclose(got["axes"][0], [0, 0, 1])  # axis normalised (kinematics.rs:289-294)
    assert got["lb"][1] == -np.inf and got["ub"][1] == np.inf  # upper - lower <= 0 (kinematics.rs:299-303)
    # sub-chain a -> c: one revolute joint and a trailing fixed joint
    assert list(Robot.from_urdf_str(text, "a", "c").chain_tables()["types"]) == [1, 0]


@pytest.mark.parametrize("base,ee,msg", [("zz", "d", "base link 'zz' does not exist"),
                                         ("a", "zz", "EE link 'zz' does not exist"),
                                         ("d", "a", "no path from base to EE link"),
                                         ("b", "c", "kinematic chain is empty")])
def test_loader_errors_carry_the_reference_messages(built, base, ee, msg):
    from optik_amd import Robot
    with pytest.raises(RuntimeError, match=re.escape(msg)):
        Robot.from_urdf_str(URDF_TMPL.format(t1="revolute", t3="revolute"), base, ee)


def test_unsupported_joint_type_and_bad_xml(built):
    from optik_amd import Robot
    with pytest.raises(RuntimeError, match="joint type not supported"):
        Robot.from_urdf_str(URDF_TMPL.format(t1="continuous", t3="revolute"), "a", "d")
    with pytest.raises(RuntimeError, match="error parsing URDF file"):
        Robot.from_urdf_str("<robot><link name='a'></robot>", "a", "a")
    with pytest.raises(RuntimeError, match="error parsing URDF file"):
        Robot.from_urdf_file("/nonexistent/x.urdf", "a", "b")


def test_seed_outside_limits_is_rejected_before_any_gpu_work(built):
    """tests/test_ik.rs:10-22: the message contains "joint limits"."""
    from optik_amd import Robot, SolverConfig
    r = Robot.from_urdf_file(os.path.join(REF_GOLDEN, "ur3e.urdf"), "ur_base_link", "ur_ee_link")
    _, ub = r.joint_limits()
    x0 = [0.0] * 6
    x0[4] = ub[4] + 1.0
    with pytest.raises(RuntimeError, match="joint limits"):
        r.ik(SolverConfig(), np.eye(4), x0)


def test_solver_config_guards():
    from optik_amd import SolverConfig
    with pytest.raises(ValueError):
        SolverConfig(max_time=0.0, max_restarts=0)   # optik-py/src/lib.rs:45-47
    with pytest.raises(ValueError):
        SolverConfig(solution_mode="fast")           # config.rs:10-20
    c = SolverConfig()
    assert (c.solution_mode, c.max_time, c.tol_f, c.tol_df, c.tol_dx) == ("speed", 0.1, 1e-6, -1.0, -1.0)
    assert c.to_c().max_restarts == 0                # u64::MAX and 0 both mean unlimited (lib.rs:273-277)


def test_no_gpu_means_loud_failure(built):
    """Without a device every compute entry point fails; nothing falls back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from optik_amd import Robot, SolverConfig, OptikHipError
    from optik_amd import device
    r = Robot.from_urdf_file(os.path.join(REF_GOLDEN, "ur3e.urdf"), "ur_base_link", "ur_ee_link")
    with pytest.raises(RuntimeError):
        r.fk([0.0] * 6)
    with pytest.raises(RuntimeError):
        r.ik(SolverConfig(max_time=0.0, max_restarts=4), np.eye(4), [0.0] * 6)
    with pytest.raises(OptikHipError):
        device.HipChain(**r.chain_tables())
    with pytest.raises(OptikHipError):
        device.probe(0, np.ones(4), np.ones(4))
    with pytest.raises(RuntimeError):
        r.ik_batch_arrays(SolverConfig(max_time=0.0, max_restarts=4), np.tile(np.eye(4), (3, 1, 1)), np.zeros((3, 6)))
    with pytest.raises(RuntimeError):
        r.diff_ik([0.0] * 6, [0.1, 0, 0, 0, 0, 0], [1.0] * 6)


def test_batch_inputs_are_validated_on_the_host(built):
    """optik_robot_ik_batch_poses checks every target with parse_pose's isometry test
    (optik-py/src/lib.rs:8-15) and every seed against the joint limits (lib.rs:251-254) before any
    GPU work: both errors surface without a device."""
    from optik_amd import Robot, SolverConfig
    r = Robot.from_urdf_file(os.path.join(REF_GOLDEN, "ur3e.urdf"), "ur_base_link", "ur_ee_link")
    lb, ub = (np.array(v) for v in r.joint_limits())
    good = np.tile(np.eye(4), (4, 1, 1))
    seeds = np.tile((lb + ub) / 2, (4, 1))
    for bad_at, edit in ((2, lambda m: m.__setitem__((slice(0, 3), slice(0, 3)), m[:3, :3] * 1.001)),   # not orthonormal
                         (0, lambda m: m.__setitem__((3, 0), 1e-9)),                                       # bottom row
                         (3, lambda m: m.__setitem__((slice(0, 3), 0), -m[:3, 0])),                        # det = -1
                         (1, lambda m: m.__setitem__((0, 3), float("nan")) or m.__setitem__((0, 0), float("nan")))):
        tg = good.copy()
        edit(tg[bad_at])
        with pytest.raises(ValueError, match="invalid target transform"):
            r.ik_batch_arrays(SolverConfig(), tg, seeds)
    out_of_limits = seeds.copy()
    out_of_limits[2, 4] = ub[4] + 1.0
    with pytest.raises(RuntimeError, match="joint limits"):
        r.ik_batch_arrays(SolverConfig(), good, out_of_limits)
    with pytest.raises(ValueError):
        r.ik_batch_arrays(SolverConfig(), good[:, :3], seeds)


def test_headers_compile_as_c_and_link(built, tmp_path):
    """include/optik.h and include/optik_hip.h are plain C (what a cgo / FFI binding consumes): a
    C11 translation unit that uses the reference's 11 symbols compiles with gcc and links against
    the shared library (no GPU call is made: it only loads a URDF and reads the joint limits)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "optik_amd", "csrc")
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include "optik.h"
int main(int argc, char **argv) {
    optik_robot *r = optik_robot_from_urdf_file(argv[1], argv[2], argv[3]);
    unsigned n = optik_robot_num_positions(r);
    const double *lim = optik_robot_joint_limits(r);
    printf("%u %.17g %.17g\n", n, lim[0], lim[n]);
    free((void *)lim);
    optik_robot_set_parallelism(r, 4);
    CSolverConfig cfg;
    (void)cfg; (void)sizeof(optik_hip_ik_outputs);
    /* referenced, not called (they need a GPU) */
    void *fns[] = {(void *)optik_robot_ik, (void *)optik_robot_fk, (void *)optik_robot_joint_jacobian,
                   (void *)optik_robot_diff_ik, (void *)optik_robot_random_configuration,
                   (void *)optik_robot_from_urdf_str, (void *)optik_hip_ik_batch, (void *)optik_hip_ik_host};
    printf("%d\n", (int)(sizeof fns / sizeof fns[0]));
    optik_robot_free(r);
    return 0;
}
''')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src),
                           "-L", libdir, "-loptik_amd", "-Wl,-rpath," + libdir, "-o", str(exe)])
    path, base, ee = ROBOT_SPECS["panda"]
    out = subprocess.check_output([str(exe), path, base, ee], text=True).split()
    assert out[0] == "7" and out[3] == "8"
    assert abs(float(out[1]) + 2.8973) < 1e-12 and abs(float(out[2]) - 2.8973) < 1e-12
