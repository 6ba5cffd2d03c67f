"""-m gpu: the HIP path against the committed fixtures of tests/golden/generated/ (both GPU
paths, both readings of rand's random_range): seeds, per-restart status / evaluation count /
x / f, and the Speed and Quality winners, bit for bit."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from gpu_util import assert_bit_equal

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GEN = os.path.join(GOLDEN, "generated")


def _load(name):
    with open(os.path.join(GEN, f"{name}.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def hip_chains(chains):
    from optik_amd import device
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return {name: device.HipChain(**chains[name][0]) for name in ("ur3e", "panda", "ur10", "arm10")}


@pytest.mark.parametrize("robot", ["ur3e", "panda", "ur10", "arm10"])
@pytest.mark.parametrize("rule", ["single_inclusive", "new_inclusive"])
def test_seeds_match_fixture(hip_chains, robot, rule):
    from optik_amd import _native as nat
    hc = hip_chains[robot]
    hc.set_range_rule({"single_inclusive": nat.RANGE_SINGLE_INCLUSIVE, "new_inclusive": nat.RANGE_NEW_INCLUSIVE}[rule])
    try:
        want = np.array(_load("rng")["robots"][robot][rule])
        got = hc.seed_batch(1, len(want)).cpu().numpy().T
        assert_bit_equal(got, want, "restart seeds vs fixture")
    finally:
        hc.set_range_rule(nat.RANGE_SINGLE_INCLUSIVE)


@pytest.mark.parametrize("robot", ["ur3e", "panda", "ur10", "arm10"])
@pytest.mark.parametrize("rule", ["single_inclusive", "new_inclusive"])
@pytest.mark.parametrize("path", ["kernel", "quad", "lane64"])
def test_restarts_and_winners_match_fixture(hip_chains, robot, rule, path):
    from optik_amd import _native as nat
    doc = _load(robot)
    hc = hip_chains[robot]
    R = doc["n_restarts"]
    tgd = torch.tensor([doc["target_pose7"]], dtype=torch.float64, device="cuda")
    x0d = torch.tensor([doc["x0"]], dtype=torch.float64, device="cuda")
    hc.set_range_rule({"single_inclusive": nat.RANGE_SINGLE_INCLUSIVE, "new_inclusive": nat.RANGE_NEW_INCLUSIVE}[rule])
    try:
        for tol, blk in doc["rules"][rule].items():
            tol_f = float(tol.split("_")[-1])
            for mode in ("speed", "quality"):
                cfg = nat.make_config(solution_mode=mode, tol_f=tol_f)
                if path == "kernel":
                    out = hc.ik_batch(cfg, tgd, x0d, 0, R)
                elif path == "lane64":  # (the lane-per-restart form, forced at the fixture's size; arm10: the general solver)
                    with nat.options(solve_kernel="lane64"):
                        out = hc.ik_batch(cfg, tgd, x0d, 0, R)
                        torch.cuda.synchronize()
                else:  # (the quad solver forced; arm10: the general solver)
                    with nat.options(solve_kernel="quad"):
                        out = hc.ik_batch(cfg, tgd, x0d, 0, R)
                        torch.cuda.synchronize()
                torch.cuda.synchronize()
                rs = blk["restarts"]
                assert out["status"].cpu().tolist() == [r["status"] for r in rs]
                assert out["evals"].cpu().tolist() == [r["evals"] for r in rs]
                assert_bit_equal(out["f"].cpu().numpy(), [r["f"] for r in rs], "f vs fixture")
                assert_bit_equal(out["x"].cpu().numpy(), np.array([r["x"] for r in rs]).T, "x vs fixture")
                w = blk["winners"][mode]
                assert int(out["win_idx"].cpu()[0]) == w["index"]
                assert_bit_equal(out["win_x"].cpu().numpy()[0], w["x"], "winner x vs fixture")
                assert_bit_equal(out["win_f"].cpu().numpy(), [w["f"]], "winner f vs fixture")
    finally:
        hc.set_range_rule(nat.RANGE_SINGLE_INCLUSIVE)


# ---- the HIP math functions against the reference's OWN golden vectors (a6-a9), directly ----------
# /root/reference/crates/optik/tests/test_math.rs:14-61 (ground truth from Pinocchio), abs 1e-6 as
# there; the fixtures are the reference's data files (tests/golden/reference/README.md).

REF_TOL = 1e-6


def _ref(name):
    from conftest import REF_GOLDEN
    with open(os.path.join(REF_GOLDEN, name)) as fh:
        return json.load(fh)


def _math_inputs():
    inp = _ref("test_math_inputs.json")
    return np.array([list(i["translation"]) + list(i["rotation"]) for i in inp], dtype=np.float64)


def test_hip_so3_log_vs_reference_fixture():
    from optik_amd import device
    want = np.array([np.ravel(o) for o in _ref("test_math_outputs_so3_log.json")])
    np.testing.assert_allclose(device.probe_math(0, _math_inputs()), want, atol=REF_TOL, rtol=0)
    # test_math.rs:24-30: the zero rotation
    z = device.probe_math(0, np.array([[0, 0, 0, 0, 0, 0, 1.0]]))
    np.testing.assert_allclose(z, np.zeros((1, 3)), atol=REF_TOL, rtol=0)


def test_hip_so3_right_jacobian_vs_reference_fixture():
    from optik_amd import device
    want = np.array([np.array(o).reshape(3, 3).T for o in _ref("test_math_outputs_so3_right_jacobian.json")])
    np.testing.assert_allclose(device.probe_math(1, _math_inputs()), want, atol=REF_TOL, rtol=0)


def test_hip_se3_log_vs_reference_fixture():
    from optik_amd import device
    want = np.array([np.ravel(o) for o in _ref("test_math_outputs_se3_log.json")])
    np.testing.assert_allclose(device.probe_math(2, _math_inputs()), want, atol=REF_TOL, rtol=0)


def test_hip_se3_right_jacobian_vs_reference_fixture():
    from optik_amd import device
    want = np.array([np.array(o).reshape(6, 6).T for o in _ref("test_math_outputs_se3_right_jacobian.json")])
    np.testing.assert_allclose(device.probe_math(3, _math_inputs()), want, atol=REF_TOL, rtol=0)


def test_hip_math_probes_equal_the_oracle_bit_for_bit(oracle):
    from optik_amd import device
    p = _math_inputs()
    assert_bit_equal(device.probe_math(0, p), np.array([oracle.so3_log(r[3:]) for r in p]), "so3::log")
    assert_bit_equal(device.probe_math(1, p), np.array([oracle.so3_right_jacobian(oracle.so3_log(r[3:])) for r in p]),
                     "so3::right_jacobian")
    assert_bit_equal(device.probe_math(2, p), np.array([oracle.se3_log(r[:3], r[3:]) for r in p]), "se3::log")
    assert_bit_equal(device.probe_math(3, p), np.array([oracle.se3_right_jacobian(r[:3], r[3:]) for r in p]),
                     "se3::right_jacobian")
