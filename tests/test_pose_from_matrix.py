"""Target matrix -> pose conversion of the host API, both readings of the reference's bindings (CPU only).

The reference's C binding (crates/optik-cpp/src/lib.rs:137-144) converts the 3x3 block of the target with nalgebra's
ITERATIVE UnitQuaternion::from_matrix, its Python binding (crates/optik-py/src/lib.rs:8-15) with the closed-form
from_rotation_matrix.  optik_robot_ik / optik_robot_ik_batch_ex feed the kernels the first, the Python front end
the second (include/optik.h: OPTIK_POSE_FROM_MATRIX).  Here: the host's iterative conversion equals the oracle's
twin bit for bit on 1000 random rotations, it reproduces the rotation, and the two readings -- the same rotation --
differ in the last bits of the quaternion for most inputs, which is why the C path had to get its own."""
import ctypes as C

import numpy as np
import pytest

from optik_amd import _native as nat

POSE_FROM_MATRIX = 4
dp = C.POINTER(C.c_double)


def _host(m44, flags):
    L = nat.lib()
    L.optik_pose_from_matrix.argtypes = [dp, C.c_uint32, dp]
    mc = np.ascontiguousarray(np.asarray(m44, dtype=np.float64).T).ravel()  # column-major
    p = np.zeros(7)
    assert L.optik_pose_from_matrix(mc.ctypes.data_as(dp), flags, p.ctypes.data_as(dp)) == 0
    return p


def _rotations(n, seed):
    rng = np.random.default_rng(seed)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    i, j, k, w = q.T
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = w*w+i*i-j*j-k*k; R[:, 0, 1] = 2*(i*j-w*k); R[:, 0, 2] = 2*(w*j+i*k)
    R[:, 1, 0] = 2*(w*k+i*j); R[:, 1, 1] = w*w-i*i+j*j-k*k; R[:, 1, 2] = 2*(j*k-w*i)
    R[:, 2, 0] = 2*(i*k-w*j); R[:, 2, 1] = 2*(w*i+j*k); R[:, 2, 2] = w*w-i*i-j*j+k*k
    return R, rng.normal(size=(n, 3))


def _rot_of(q):
    i, j, k, w = q
    return np.array([[w*w+i*i-j*j-k*k, 2*(i*j-w*k), 2*(w*j+i*k)], [2*(w*k+i*j), w*w-i*i+j*j-k*k, 2*(j*k-w*i)],
                     [2*(i*k-w*j), 2*(w*i+j*k), w*w-i*i-j*j+k*k]])


def test_iterative_conversion_equals_the_oracle_and_differs_from_the_closed_form(oracle):
    Rs, ts = _rotations(1000, 7)
    differ, worst = 0, 0.0
    for R, t in zip(Rs, ts):
        m = np.eye(4)
        m[:3, :3], m[:3, 3] = R, t
        it, cf = _host(m, POSE_FROM_MATRIX), _host(m, 0)
        assert np.array_equal(it[:3], t) and np.array_equal(cf[:3], t)
        ref = oracle.quat_from_matrix(R, True)
        assert np.array_equal(it[3:].view(np.uint64), ref.view(np.uint64)), (it[3:], ref)
        # both are the rotation (up to the sign of the quaternion), to roundoff
        for q in (it[3:], cf[3:]):
            assert abs(np.linalg.norm(q) - 1.0) < 4e-16 * 4
            assert np.abs(_rot_of(q) - R).max() < 5e-15
        s = 1.0 if np.dot(it[3:], cf[3:]) > 0 else -1.0
        d = np.abs(it[3:] - s * cf[3:]).max()
        worst = max(worst, d)
        differ += d != 0.0
    # the two readings agree to a few ulps and are NOT the same numbers
    assert worst < 5e-15
    assert differ > 500, differ


def test_closed_form_is_the_python_bindings_reading(oracle):
    """from_rotation_matrix's four branches (trace > 0, or the largest diagonal entry): the host's closed form is the
    oracle's bit for bit -- nalgebra's function ends in new_unchecked, so neither side normalises (round 5's
    host normalised on this path only, which made two code paths that claim to be the same function disagree)."""
    Rs, _ = _rotations(200, 11)
    # rotations by ~pi about each axis reach the three trace <= 0 branches
    for ax in range(3):
        for ang in (3.1, 3.14159, 2.9):
            u = np.zeros(3); u[ax] = 1.0
            K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
            Rs = np.concatenate([Rs, (np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K)[None]])
    for R in Rs:
        m = np.eye(4)
        m[:3, :3] = R
        cf = _host(m, 0)[3:]
        ref = oracle.quat_from_matrix(R, False)
        assert np.array_equal(cf.view(np.uint64), ref.view(np.uint64)), (cf, ref)
        it = _host(m, POSE_FROM_MATRIX)[3:]
        assert np.array_equal(it.view(np.uint64), oracle.quat_from_matrix(R, True).view(np.uint64))


def test_identity_and_non_rotations_terminate(oracle):
    assert _host(np.eye(4), POSE_FROM_MATRIX)[3:].tolist() == [0.0, 0.0, 0.0, 1.0]
    # a scaled rotation is still mapped to its rotational part (what from_matrix is for); no hang
    Rs, _ = _rotations(5, 3)
    for R in Rs:
        m = np.eye(4)
        m[:3, :3] = 1.01 * R
        q = _host(m, POSE_FROM_MATRIX)[3:]
        assert np.array_equal(q.view(np.uint64), oracle.quat_from_matrix(1.01 * R, True).view(np.uint64))
        assert np.abs(_rot_of(q / np.linalg.norm(q)) - R).max() < 1e-9


def test_degenerate_matrices_terminate(oracle):
    """nalgebra's iteration has no iteration cap (usize::MAX); this one stops after 100 000 steps: a matrix of NaNs, the
    zero matrix and a reflection come back (with whatever quaternion the iteration ends on) instead of hanging the call."""
    for R in (np.full((3, 3), np.nan), np.zeros((3, 3)), np.diag([1.0, 1.0, -1.0])):
        m = np.eye(4)
        m[:3, :3] = R
        q = _host(m, POSE_FROM_MATRIX)[3:]
        ref = oracle.quat_from_matrix(R, True)
        assert np.array_equal(q.view(np.uint64), ref.view(np.uint64))
