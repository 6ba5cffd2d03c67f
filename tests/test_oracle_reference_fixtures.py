"""Pins the C oracle against every golden vector the reference's own tests hold for
the hot path: tests/test_math.rs:14-61, tests/test_fk.rs:13-26 and the
finite-difference property of tests/test_gradient.rs:34-68 (same weights, same eps).
Tolerance 1e-6 absolute, as in the reference tests."""
import json
import os

import numpy as np

from conftest import REF_GOLDEN

TOL = 1e-6  # assert_abs_diff_eq!(..., epsilon = 1e-6) in the reference tests


def _load(name):
    with open(os.path.join(REF_GOLDEN, name)) as fh:
        return json.load(fh)


def test_so3_log(oracle):
    inp, out = _load("test_math_inputs.json"), _load("test_math_outputs_so3_log.json")
    for i, o in zip(inp, out):
        np.testing.assert_allclose(oracle.so3_log(i["rotation"]), np.ravel(o), atol=TOL, rtol=0)


def test_so3_log_singularity(oracle):
    # tests/test_math.rs:24-30: zero rotation about z
    np.testing.assert_allclose(oracle.so3_log([0, 0, 0, 1]), np.zeros(3), atol=TOL, rtol=0)


def test_so3_right_jacobian(oracle):
    inp = _load("test_math_inputs.json")
    out = _load("test_math_outputs_so3_right_jacobian.json")
    for i, o in zip(inp, out):
        w = oracle.so3_log(i["rotation"])
        ref = np.array(o).reshape(3, 3).T  # fixtures are column-major
        np.testing.assert_allclose(oracle.so3_right_jacobian(w), ref, atol=TOL, rtol=0)


def test_se3_log(oracle):
    inp, out = _load("test_math_inputs.json"), _load("test_math_outputs_se3_log.json")
    for i, o in zip(inp, out):
        np.testing.assert_allclose(oracle.se3_log(i["translation"], i["rotation"]), np.ravel(o),
                                   atol=TOL, rtol=0)


def test_se3_right_jacobian(oracle):
    inp = _load("test_math_inputs.json")
    out = _load("test_math_outputs_se3_right_jacobian.json")
    for i, o in zip(inp, out):
        ref = np.array(o).reshape(6, 6).T
        np.testing.assert_allclose(oracle.se3_right_jacobian(i["translation"], i["rotation"]), ref,
                                   atol=TOL, rtol=0)


def test_fk_ur3e(oracle, chains):
    _, ch = chains["ur3e"]
    for q, o in zip(_load("test_fk_inputs.json"), _load("test_fk_outputs.json")):
        _, ee = oracle.fk(ch, q)
        np.testing.assert_allclose(ee[:3], o["translation"], atol=TOL, rtol=0)
        r = np.array(o["rotation"])
        # nalgebra's UnitQuaternion comparison is double-cover aware
        assert min(np.abs(ee[3:] - r).max(), np.abs(ee[3:] + r).max()) < TOL


def test_gradient_analytical_vs_numerical(oracle, chains):
    """tests/test_gradient.rs:34-68 with the reference's non-trivial weights.  The
    reference draws from StdRng(42) (ChaCha12, not reproducible here); any draw works."""
    _, ch = chains["ur3e"]
    wl, wa = [0.0, 5.0, 0.25], [0.005, 1.0, 0.99]
    eps = np.finfo(float).eps ** (1.0 / 3.0)
    rng = np.random.default_rng(42)
    for _ in range(100):
        q = rng.random(6)
        quat = rng.normal(size=4)
        quat /= np.linalg.norm(quat)
        tgt = np.concatenate([rng.random(3), quat])
        _, g = oracle.eval_fg(ch, tgt, q, wl, wa)
        gn = np.zeros(6)
        for i in range(6):
            lo, hi = q.copy(), q.copy()
            lo[i] -= eps
            hi[i] += eps
            gn[i] = (oracle.eval_fg(ch, tgt, hi, wl, wa, grad=False)
                     - oracle.eval_fg(ch, tgt, lo, wl, wa, grad=False)) / (2 * eps)
        np.testing.assert_allclose(g, gn, atol=TOL, rtol=0)


def test_default_weights_take_the_rotate_path(oracle, chains):
    """Quirk Q2: [1,1,1] is not 'identity' for nalgebra's is_identity on a 3-vector,
    [1,0,0] is.  Both must give finite values; [1,1,1] ~ unweighted to roundoff."""
    _, ch = chains["ur3e"]
    rng = np.random.default_rng(1)
    q = rng.random(6)
    quat = rng.normal(size=4)
    quat /= np.linalg.norm(quat)
    tgt = np.concatenate([rng.random(3), quat])
    f1 = oracle.eval_fg(ch, tgt, q, grad=False)
    e = oracle.se3_log(*_x_target_act(oracle, ch, tgt, q))
    assert abs(f1 - float(e @ e)) < 1e-12


def _x_target_act(oracle, ch, tgt, q):
    _, ee = oracle.fk(ch, q)

    def qmul(a, b):
        return np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1],
                         a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
                         a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3],
                         a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])

    def qrot(qq, v):
        t = 2 * np.cross(qq[:3], v)
        return t * qq[3] + np.cross(qq[:3], t) + v

    qc = tgt[3:] * np.array([-1, -1, -1, 1])
    return qrot(qc, ee[:3] - tgt[:3]), qmul(qc, ee[3:])
