"""The generated fixtures under tests/golden/generated/ (tools/gen_golden.py) against the oracle.

The files are outputs of this repository's oracle, committed so that (i) the oracle cannot
drift silently, (ii) the HIP path is checked against data rather than against whatever the
oracle computes today (tests/test_gpu_golden.py), and (iii) a maintainer with cargo can run the
Rust program of INTEGRATION.md section 5 against the same numbers (tools/compare_golden.py).
Parity with the reference itself stays UNPINNED until someone does (iii): see DESIGN.md section 3.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROBOT_SPECS

GEN = os.path.join(GOLDEN, "generated")
ROBOTS = ["ur3e", "panda", "ur10", "arm10"]
RULES = {"single_inclusive": 0, "new_inclusive": 1}


def _load(name):
    with open(os.path.join(GEN, f"{name}.json")) as fh:
        return json.load(fh)


def _bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


def test_seed_from_u64_key(oracle):
    import ctypes as C
    key = (C.c_uint32 * 8)()
    oracle.lib().ok_seed_from_u64(42, key)
    assert bytes(key).hex() == _load("rng")["seed_from_u64_42_key_hex"]


@pytest.mark.parametrize("robot", ROBOTS)
@pytest.mark.parametrize("rule", list(RULES))
def test_restart_seeds(oracle, chains, robot, rule):
    d, ch = chains[robot]
    want = np.array(_load("rng")["robots"][robot][rule])
    with oracle.range_rule(RULES[rule]):
        got = np.array([oracle.restart_seed(ch, i) for i in range(1, len(want) + 1)])
    assert np.array_equal(_bits(got), _bits(want))
    assert np.all(got >= d["lb"]) and np.all(got <= d["ub"])


def test_the_two_rules_differ_only_in_the_last_bits():
    r = _load("rng")["robots"]
    for robot in ROBOTS:
        a, b = np.array(r[robot]["single_inclusive"]), np.array(r[robot]["new_inclusive"])
        assert not np.array_equal(_bits(a), _bits(b))
        assert np.max(np.abs(a - b)) < 1e-14


@pytest.mark.parametrize("robot", ROBOTS)
@pytest.mark.parametrize("rule", list(RULES))
@pytest.mark.parametrize("tol", ["tol_f_1e-6", "tol_f_1e-12"])
def test_oracle_reproduces_fixture(oracle, chains, robot, rule, tol):
    doc = _load(robot)
    assert doc["urdf"].endswith(os.path.basename(ROBOT_SPECS[robot][0]))
    d, ch = chains[robot]
    blk = doc["rules"][rule][tol]
    tol_f = float(tol.split("_")[-1])
    tgt, x0 = np.array(doc["target_pose7"]), np.array(doc["x0"])
    R = doc["n_restarts"]
    with oracle.range_rule(RULES[rule]):
        per = oracle.ik(ch, oracle.make_config("speed", tol_f=tol_f), tgt, x0, 0, R, n_threads=2,
                        early_exit=False, per_restart=True)
        for i, r in enumerate(blk["restarts"]):
            assert r["index"] == i
            assert per["status"][i] == r["status"] and per["evals"][i] == r["evals"]
            assert bool(per["success"][i]) == r["success"]
            assert np.array_equal(_bits(per["xs"][i]), _bits(r["x"])) and _bits(per["fs"][i]) == _bits(r["f"])
            if i > 0:
                assert np.array_equal(_bits(oracle.restart_seed(ch, i)), _bits(r["seed"]))
        for mode in ("speed", "quality"):
            w = oracle.ik(ch, oracle.make_config(mode, tol_f=tol_f), tgt, x0, 0, R, early_exit=(mode == "speed"))
            g = blk["winners"][mode]
            assert w["found"] == g["found"] and w["winner"] == g["index"]
            assert np.array_equal(_bits(w["x"]), _bits(g["x"])) and _bits(w["f"]) == _bits(g["f"])


@pytest.mark.parametrize("robot", ROBOTS)
def test_fixture_is_self_consistent(oracle, chains, robot):
    """Winners follow from the per-restart rows by lib.rs:397-413; successes solve the target."""
    doc = _load(robot)
    d, ch = chains[robot]
    x0, tgt = np.array(doc["x0"]), np.array(doc["target_pose7"])
    for rule, blocks in doc["rules"].items():
        for tol, blk in blocks.items():
            tol_f = float(tol.split("_")[-1])
            ok = [r for r in blk["restarts"] if r["success"]]
            assert ok and len(ok) < len(blk["restarts"])
            for r in blk["restarts"]:
                assert r["success"] == (r["status"] == 2)       # defaults: only StopVal counts (lib.rs:376-379)
                if r["success"]:
                    assert r["f"] < tol_f
                    _, ee = oracle.fk(ch, np.array(r["x"]))
                    assert np.allclose(ee[:3], tgt[:3], atol=2e-3)
                assert np.all(np.array(r["x"]) >= d["lb"]) and np.all(np.array(r["x"]) <= d["ub"])
            assert blk["winners"]["speed"]["index"] == ok[0]["index"]
            dist = [np.linalg.norm(np.array(r["x"]) - x0) for r in ok]
            assert blk["winners"]["quality"]["index"] == ok[int(np.argmin(dist))]["index"]
            assert blk["restarts"][0]["seed"] == doc["x0"]
