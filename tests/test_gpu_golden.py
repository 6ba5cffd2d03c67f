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
    return {name: device.HipChain(**chains[name][0]) for name in ("ur3e", "panda", "ur10")}


@pytest.mark.parametrize("robot", ["ur3e", "panda", "ur10"])
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


@pytest.mark.parametrize("robot", ["ur3e", "panda", "ur10"])
@pytest.mark.parametrize("rule", ["single_inclusive", "new_inclusive"])
@pytest.mark.parametrize("path", ["kernel", "engine"])
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
                else:
                    out = hc.engine_submit(cfg, tgd, x0d, 0, R)
                    hc.engine_run()
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
