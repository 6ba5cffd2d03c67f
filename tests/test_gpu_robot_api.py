"""-m gpu: the reference-shaped front end (optik.pyi surface over the optik_robot_* C ABI)
on the real device, written after the reference's own integration tests
(/root/reference/crates/optik/tests/test_fk.rs, test_ik.rs)."""
import json
import os
import time

import numpy as np
import pytest

from conftest import REF_GOLDEN, ROBOTS
from gpu_util import assert_bit_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ur3e():
    from optik_amd import Robot
    r = Robot.from_urdf_file(os.path.join(REF_GOLDEN, "ur3e.urdf"), "ur_base_link", "ur_ee_link")
    r.set_parallelism(1)  # deterministic Speed answers, as tests/test_ik.rs:45-89 asks of the reference
    return r


@pytest.fixture(scope="module")
def panda():
    from optik_amd import Robot
    r = Robot.from_urdf_file(os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link8")
    r.set_parallelism(1)
    return r


def _quat_to_R(q):
    i, j, k, w = q
    return np.array([[w*w + i*i - j*j - k*k, 2*(i*j - w*k), 2*(w*j + i*k)],
                     [2*(w*k + i*j), w*w - i*i + j*j - k*k, 2*(j*k - w*i)],
                     [2*(i*k - w*j), 2*(w*i + j*k), w*w - i*i - j*j + k*k]])


def test_fk_golden(ur3e):
    """tests/test_fk.rs:13-26 on the GPU path: 50 golden UR3e poses, abs 1e-6."""
    inp = json.load(open(os.path.join(REF_GOLDEN, "test_fk_inputs.json")))
    out = json.load(open(os.path.join(REF_GOLDEN, "test_fk_outputs.json")))
    for q, o in zip(inp, out):
        m = np.array(ur3e.fk(q))
        np.testing.assert_allclose(m[:3, 3], o["translation"], atol=1e-6, rtol=0)
        np.testing.assert_allclose(m[:3, :3], _quat_to_R(o["rotation"]), atol=1e-6, rtol=0)
        assert np.allclose(m[3], [0, 0, 0, 1])


def test_jacobian_matches_oracle_and_finite_differences(ur3e, oracle, chains):
    _, ch = chains["ur3e"]
    rng = np.random.default_rng(0)
    for _ in range(5):
        q = rng.uniform(-1, 1, 6)
        J = np.array(ur3e.joint_jacobian(q))
        assert J.shape == (6, 6)
        assert_bit_equal(J, oracle.joint_jacobian(ch, q), "jacobian")
    with pytest.raises(ValueError):
        ur3e.fk([0.0] * 5)


def test_invalid_seed(ur3e):
    """tests/test_ik.rs:10-22."""
    from optik_amd import SolverConfig
    _, ub = ur3e.joint_limits()
    x0 = [0.0] * 6
    x0[4] = ub[4] + 1.0
    with pytest.raises(RuntimeError, match="joint limits"):
        ur3e.ik(SolverConfig(), np.eye(4), x0)


def test_stopping_maxtime(ur3e):
    """tests/test_ik.rs:24-43: impossible goal, max_time = 0.05 s, returns within 0.05 +- 0.1 s."""
    from optik_amd import SolverConfig
    tgt = np.eye(4)
    tgt[:3, 3] = 100.0
    ur3e.fk([0.0] * 6)  # warm the device up before timing
    t0 = time.perf_counter()
    sol = ur3e.ik(SolverConfig(max_time=0.05), tgt, [0.0] * 6)
    dt = time.perf_counter() - t0
    assert sol is None
    assert abs(dt - 0.05) < 0.1, dt


def test_determinism_and_oracle_agreement(ur3e, oracle, chains):
    """tests/test_ik.rs:45-89 (11 identical calls), plus: the solution IS the oracle's."""
    from optik_amd import SolverConfig
    _, ch = chains["ur3e"]
    rng = np.random.default_rng(42)
    tgt = np.array(ur3e.fk(rng.random(6)))
    cfg = SolverConfig(max_time=0.0, max_restarts=25)
    x_sol, c, idx = ur3e.ik(cfg, tgt, [0.0] * 6, return_index=True)
    for _ in range(10):
        x_i, _ = ur3e.ik(cfg, tgt, [0.0] * 6)
        assert x_i == x_sol
    _, ee = oracle.fk(ch, rng.random(6) * 0)  # noqa: F841 (keeps the oracle chain warm)
    t7 = _mat_to_pose7(tgt)
    ref = oracle.ik(ch, oracle.make_config(solution_mode="speed", max_restarts=25), t7, np.zeros(6), 0, 25)
    assert ref["found"] and ref["winner"] == idx
    np.testing.assert_allclose(x_sol, ref["x"], atol=1e-6, rtol=0)  # north-star tolerance
    assert abs(c - ref["f"]) < 1e-12


def _mat_to_pose7(m):
    m = np.asarray(m)
    tr = np.trace(m[:3, :3])
    # Shepperd, trace > 0 branch is enough for the FK-generated targets used here
    if tr > 0:
        d = np.sqrt(tr + 1.0) * 2.0
        w = 0.25 * d
        i, j, k = (m[2, 1] - m[1, 2]) / d, (m[0, 2] - m[2, 0]) / d, (m[1, 0] - m[0, 1]) / d
    else:
        idx = int(np.argmax(np.diag(m[:3, :3])))
        a, b, c = idx, (idx + 1) % 3, (idx + 2) % 3
        d = np.sqrt(1.0 + m[a, a] - m[b, b] - m[c, c]) * 2.0
        v = [0.0, 0.0, 0.0]
        v[a] = 0.25 * d
        v[b] = (m[a, b] + m[b, a]) / d
        v[c] = (m[a, c] + m[c, a]) / d
        w = (m[c, b] - m[b, c]) / d
        i, j, k = v
    q = np.array([i, j, k, w])
    q /= np.linalg.norm(q)
    return np.concatenate([m[:3, 3], q])


def test_solution_forward_backward(ur3e):
    """tests/test_ik.rs:91-130: tol_f = 1e-12, 25 restarts, FK(ik(T)) == T to 1e-6."""
    from optik_amd import SolverConfig
    rng = np.random.default_rng(42)
    cfg = SolverConfig(solution_mode="speed", tol_f=1e-12, max_time=0.0, max_restarts=25)
    for _ in range(10):
        tgt = np.array(ur3e.fk(rng.random(6)))
        x, c = ur3e.ik(cfg, tgt, [0.0] * 6)
        np.testing.assert_allclose(np.array(ur3e.fk(x)), tgt, atol=1e-6, rtol=0)
        assert c < 1e-12


def test_solution_quality(ur3e):
    """tests/test_ik.rs:132-182: Quality is never farther from the seed than Speed."""
    from optik_amd import SolverConfig
    rng = np.random.default_rng(42)
    speed = SolverConfig(solution_mode="speed", max_time=0.0, max_restarts=15)
    quality = SolverConfig(solution_mode="quality", max_time=0.0, max_restarts=15)
    for _ in range(20):
        tgt = np.array(ur3e.fk(rng.random(6)))
        xs, _ = ur3e.ik(speed, tgt, [0.0] * 6)
        xq, _ = ur3e.ik(quality, tgt, [0.0] * 6)
        assert np.linalg.norm(xq) <= np.linalg.norm(xs)


def test_panda_default_config_and_ee_offset(panda):
    """Default SolverConfig (0.1 s budget, unlimited restarts, Speed) on the headline robot,
    with a non-trivial ee_offset (optik.pyi:36-42)."""
    from optik_amd import SolverConfig
    rng = np.random.default_rng(3)
    lb, ub = (np.array(v) for v in panda.joint_limits())
    off = np.eye(4)
    off[:3, 3] = [0.0, 0.05, 0.1]
    off[:3, :3] = _quat_to_R(np.array([0.0, 0.0, np.sin(0.3), np.cos(0.3)]))
    solved = 0
    for _ in range(10):
        tgt = np.array(panda.fk(rng.uniform(lb, ub), off))
        sol = panda.ik(SolverConfig(), tgt, rng.uniform(lb, ub).tolist(), off)
        if sol is not None:
            x, c = sol
            assert c < 1e-6 and np.all(np.array(x) >= lb) and np.all(np.array(x) <= ub)
            # f < 1e-6 is a squared log-error (quirk Q8): pose error below ~1e-3
            np.testing.assert_allclose(np.array(panda.fk(x, off)), tgt, atol=2e-3, rtol=0)
            solved += 1
    assert solved >= 9


def test_c_abi_ik_returns_malloced_buffer(ur3e):
    """optik_robot_ik / optik_robot_fk as the reference's C++ wrapper uses them
    (crates/optik-cpp/src/lib.cpp:105-120): col-major target, free() the result."""
    import ctypes as C
    from optik_amd import _native as nat
    L = nat.lib()
    dp = C.POINTER(C.c_double)
    L.optik_robot_fk.restype = dp
    L.optik_robot_fk.argtypes = [C.c_void_p, dp]
    L.optik_robot_ik.restype = dp
    L.optik_robot_ik.argtypes = [C.c_void_p, C.POINTER(nat.SolverConfigC), dp, dp]
    q = (C.c_double * 6)(0.1, 0.2, 0.0, 0.3, -0.2, -1.1)
    m = L.optik_robot_fk(ur3e._h, q)
    tgt16 = (C.c_double * 16)(*[m[i] for i in range(16)])
    assert abs(tgt16[15] - 1.0) < 1e-15 and tgt16[3] == 0.0   # column-major homogeneous
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(m)
    cfg = nat.make_config("speed", 0.0, 64)
    x0 = (C.c_double * 6)(*([0.0] * 6))
    sol = L.optik_robot_ik(ur3e._h, C.byref(cfg), tgt16, x0)
    assert bool(sol)
    x = [sol[i] for i in range(6)]
    libc.free(sol)
    np.testing.assert_allclose(np.array(ur3e.fk(x)), np.array(list(tgt16)).reshape(4, 4).T, atol=2e-3)
    far = (C.c_double * 16)(*np.eye(4).T.ravel())
    far[12] = far[13] = far[14] = 100.0
    assert not bool(L.optik_robot_ik(ur3e._h, C.byref(cfg), far, x0))   # NULL = no solution


def test_c_path_feeds_the_kernels_the_iteratively_converted_target(oracle, chains):
    """optik_robot_ik (the reference's C symbol) converts the 4x4 target with nalgebra's ITERATIVE from_matrix
    (optik-cpp/src/lib.rs:141-142), optik_robot_ik_ex / the Python front end with the closed form
    (optik-py/src/lib.rs:8-15): each returns, bit for bit, what the oracle returns for THAT pose -- and over 40 targets
    the two readings do not always return the same joint angles (last bits of the target decide long restarts)."""
    import ctypes as C
    from optik_amd import Robot
    from optik_amd import _native as nat
    L = nat.lib()
    dp = C.POINTER(C.c_double)
    L.optik_pose_from_matrix.argtypes = [dp, C.c_uint32, dp]
    L.optik_robot_ik_pose.argtypes = [C.c_void_p, C.POINTER(nat.SolverConfigC), dp, C.c_uint32, dp, dp, dp, dp,
                                      C.POINTER(C.c_uint64)]
    robot = Robot.from_urdf_file(os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link8")
    robot.set_parallelism(1)  # (the deterministic rule: the lowest successful index)
    d, ch = chains["panda"]
    rng = np.random.default_rng(123)
    cfg = nat.make_config("speed", 0.0, 96)
    ocfg = oracle.make_config(solution_mode="speed", max_restarts=96)
    differ = 0
    for trial in range(40):
        m = np.array(robot.fk(rng.uniform(d["lb"], d["ub"])))
        mc = np.ascontiguousarray(m.T).ravel()
        x0 = rng.uniform(d["lb"], d["ub"])
        got = {}
        for flags in (0, 4):  # 4 = OPTIK_POSE_FROM_MATRIX
            p7 = np.zeros(7)
            assert L.optik_pose_from_matrix(mc.ctypes.data_as(dp), flags, p7.ctypes.data_as(dp)) == 0
            x, f, idx = np.zeros(7), C.c_double(0.0), C.c_uint64(0)
            rc = L.optik_robot_ik_pose(robot._h, C.byref(cfg), mc.ctypes.data_as(dp), flags, x0.ctypes.data_as(dp), None,
                                       x.ctypes.data_as(dp), C.byref(f), C.byref(idx))
            ref = oracle.ik(ch, ocfg, p7, x0, 0, 96, n_threads=4, early_exit=True)
            assert (rc == 0) == bool(ref["found"])
            if rc == 0:
                assert idx.value == ref["winner"]
                assert_bit_equal(x, ref["x"], f"trial {trial} flags {flags}")
            got[flags] = x.copy()
        differ += not np.array_equal(got[0], got[4])
    assert differ > 0


def test_ik_batch_of_41000_targets_equals_individual_calls(panda):
    """A Speed batch of tens of thousands of targets (one launch per round, robot_host.cpp:ik_batch_on_device;
    rounds 1-4 ran batches of this size on a streaming engine): the same answers as ik() target by target."""
    from optik_amd import SolverConfig
    rng = np.random.default_rng(18)
    lb, ub = (np.array(v) for v in panda.joint_limits())
    T = 41000
    base = [np.array(panda.fk(rng.uniform(lb, ub))) for _ in range(40)]
    targets = np.stack([base[t % 40] for t in range(T)])
    x0s = rng.uniform(lb, ub, size=(T, 7))
    cfg = SolverConfig(solution_mode="speed", max_time=0.0, max_restarts=64)
    x, f, found = panda.ik_batch_arrays(cfg, targets, x0s)
    assert found.mean() > 0.9
    for t in (0, 13, 4097, 20011, 40999):
        single = panda.ik(cfg, targets[t], x0s[t].tolist())
        assert (single is not None) == bool(found[t])
        if single is not None:
            assert x[t].tolist() == single[0] and f[t] == single[1]


def test_ik_batch_equals_individual_calls(panda, oracle, chains):
    """Robot.ik_batch (restart-major queue, early exit, on the quad solver) returns for every target exactly
    what ik() returns for it alone -- and what the oracle's restart loop returns."""
    from optik_amd import SolverConfig
    _, ch = chains["panda"]
    rng = np.random.default_rng(8)
    lb, ub = (np.array(v) for v in panda.joint_limits())
    T = 48
    targets = [np.array(panda.fk(rng.uniform(lb, ub))) for _ in range(T)]
    x0s = rng.uniform(lb, ub, size=(T, 7))
    for mode in ("speed", "quality"):
        cfg = SolverConfig(solution_mode=mode, max_time=0.0, max_restarts=40)
        batch = panda.ik_batch(cfg, targets, x0s)
        assert len(batch) == T
        for t in (0, 7, 19, 33, 47):
            single = panda.ik(cfg, targets[t], x0s[t].tolist())
            assert (batch[t] is None) == (single is None)
            if single is not None:
                assert batch[t][0] == single[0] and batch[t][1] == single[1]
        # against the oracle (targets enter through the 4x4 -> pose conversion of the host layer)
        for t in (3, 11):
            ref = oracle.ik(ch, oracle.make_config(solution_mode=mode, max_restarts=40), _mat_to_pose7(targets[t]),
                            x0s[t], 0, 40)
            assert ref["found"] == (batch[t] is not None)
            if ref["found"]:
                np.testing.assert_allclose(batch[t][0], ref["x"], atol=1e-6, rtol=0)
    # Quality with thousands of restarts per target: one round (one launch) covers them all
    cfg = SolverConfig(solution_mode="quality", max_time=0.0, max_restarts=3000)
    batch = panda.ik_batch(cfg, targets[:20], x0s[:20])
    for t in (0, 9, 19):
        assert batch[t] == panda.ik(cfg, targets[t], x0s[t].tolist())
    # a target nobody can reach stays None; the others are unaffected
    far = np.eye(4)
    far[:3, 3] = 50.0
    res = panda.ik_batch(SolverConfig(max_time=0.0, max_restarts=16), [targets[0], far], x0s[:2])
    assert res[1] is None and res[0] is not None
    # the array form carries the same numbers
    xa, fa, ok = panda.ik_batch_arrays(SolverConfig(max_time=0.0, max_restarts=16), [targets[0], far], x0s[:2])
    assert ok.tolist() == [True, False] and xa[0].tolist() == res[0][0] and fa[0] == res[0][1]
    # parse_pose's isometry test applies to every target of a batch (optik-py/src/lib.rs:8-15)
    bad = np.array(targets[1])
    bad[:3, :3] *= 1.001
    with pytest.raises(ValueError, match="invalid target transform"):
        panda.ik_batch(SolverConfig(max_time=0.0, max_restarts=16), [targets[0], bad], x0s[:2])


def test_big_speed_batch_equals_individual_calls(panda):
    """A Speed batch of 65 536 targets runs in rounds capped at ~4 M work items (64 restart indices per target in
    the first, the unsolved rest goes through later rounds): the answers are still those of ik() alone -- the
    lowest successful restart index with set_parallelism(1)."""
    from optik_amd import SolverConfig
    rng = np.random.default_rng(15)
    lb, ub = (np.array(v) for v in panda.joint_limits())
    D, copies = 256, 256
    distinct = np.array([panda.fk(rng.uniform(lb, ub)) for _ in range(D)])
    seeds = rng.uniform(lb, ub, size=(D, 7))
    far = np.eye(4)
    far[:3, 3] = 50.0
    distinct[17] = far  # one target nobody can reach: runs every round, stays unsolved
    targets = np.tile(distinct, (copies, 1, 1))
    x0s = np.tile(seeds, (copies, 1))
    cfg = SolverConfig(solution_mode="speed", max_time=0.0, max_restarts=300)
    x, f, ok = panda.ik_batch_arrays(cfg, targets, x0s)
    assert x.shape == (D * copies, 7)
    xr, fr, okr = x.reshape(copies, D, 7), f.reshape(copies, D), ok.reshape(copies, D)
    assert np.all(okr == okr[0]) and np.all(xr == xr[0]) and np.all(fr == fr[0])
    assert not okr[0, 17] and okr[0].sum() == D - 1
    for t in (0, 5, 16, 18, 100, 255):
        single = panda.ik(cfg, distinct[t], seeds[t].tolist())
        assert single is not None and single[0] == xr[0, t].tolist() and single[1] == fr[0, t]


def test_speed_batch_with_hard_targets_runs_growing_rounds(panda):
    """A Speed batch's first round is 256 restart indices per target; what it leaves unsolved goes
    through rounds four times as long each: reachable targets keep the answer of ik() alone, unreachable ones come back
    None after all max_restarts -- in a time that shows the rounds grew (390 rounds of 256 took 0.8 s)."""
    from optik_amd import SolverConfig
    rng = np.random.default_rng(61)
    lb, ub = (np.array(v) for v in panda.joint_limits())
    T = 12
    targets = np.array([panda.fk(rng.uniform(lb, ub)) for _ in range(T)])
    far = np.eye(4)
    far[:3, 3] = 50.0
    targets[[2, 7]] = far
    x0s = rng.uniform(lb, ub, size=(T, 7))
    cfg = SolverConfig(solution_mode="speed", max_time=0.0, max_restarts=60_000)
    panda.ik_batch_arrays(cfg, targets, x0s)  # (the first call allocates the launch workspace)
    t0 = time.perf_counter()
    x, f, ok = panda.ik_batch_arrays(cfg, targets, x0s)
    dt = time.perf_counter() - t0
    assert ok.tolist() == [t not in (2, 7) for t in range(T)]
    for t in (0, 3, 11):
        single = panda.ik(cfg, targets[t], x0s[t].tolist())
        assert single is not None and single[0] == x[t].tolist() and single[1] == f[t]
    assert dt < 0.25, dt


def test_eight_dof_batch_equals_individual_calls():
    """n = 8 runs on the quad solve kernel (nine-row NNLS columns in LDS; the lane-per-restart form is built for
    n <= 7): a Speed batch is handed out
    restart-major, same answers as ik() alone."""
    from conftest import TEST_ROBOTS
    from optik_amd import Robot, SolverConfig
    r = Robot.from_urdf_file(os.path.join(TEST_ROBOTS, "arm8.urdf"), "l0", "l9")
    r.set_parallelism(1)
    assert r.num_positions() == 8
    rng = np.random.default_rng(5)
    lb, ub = (np.array(v) for v in r.joint_limits())
    T = 40
    targets = np.array([r.fk(rng.uniform(lb, ub)) for _ in range(T)])
    x0s = rng.uniform(lb, ub, size=(T, 8))
    for mode in ("speed", "quality"):
        cfg = SolverConfig(solution_mode=mode, max_time=0.0, max_restarts=48)
        batch = r.ik_batch(cfg, targets, x0s)
        for t in (0, 13, 39):
            assert batch[t] == r.ik(cfg, targets[t], x0s[t].tolist())


def _world_jacobian(robot, x):
    fk = np.array(robot.fk(x))
    J = np.array(robot.joint_jacobian(x))
    R = fk[:3, :3]
    return np.vstack([R @ J[:3], R @ J[3:]])


@pytest.mark.parametrize("name", ["ur3e", "panda"])
def test_diff_ik(name, ur3e, panda):
    """tests/test_ik.rs:185-209 (alpha in [0, 1], |v| <= v_max), plus what its TODO leaves out:
    the Cartesian velocity is tracked (J_W v = alpha V) and alpha is the LP optimum (checked
    against scipy's HiGHS on the same LP)."""
    from scipy.optimize import linprog
    robot = {"ur3e": ur3e, "panda": panda}[name]
    n = robot.num_positions()
    rng = np.random.default_rng(42)
    lb, ub = (np.array(v) for v in robot.joint_limits())
    eps = 1e-6
    for trial in range(20):
        x0 = rng.uniform(lb, ub)
        v_max = np.ones(n) if trial % 2 == 0 else rng.uniform(0.2, 2.0, size=n)
        V = rng.random(6) * (1.0 if trial % 3 else 0.05)  # small twists reach alpha = 1
        out = robot.diff_ik(x0.tolist(), V.tolist(), v_max.tolist())
        assert out is not None
        alpha, v = out[0], np.array(out[1])
        assert -eps <= alpha <= 1.0 + eps
        assert np.all(v >= -v_max - eps) and np.all(v <= v_max + eps)
        JW = _world_jacobian(robot, x0.tolist())
        assert np.allclose(JW @ v, alpha * V, atol=1e-8)
        # the same LP on the CPU: max alpha s.t. JW v - alpha V = 0, |v| <= v_max, 0 <= alpha <= 1
        c = np.zeros(n + 1); c[n] = -1.0
        res = linprog(c, A_eq=np.hstack([JW, -V[:, None]]), b_eq=np.zeros(6),
                      bounds=[(-m, m) for m in v_max] + [(0.0, 1.0)], method="highs")
        assert res.status == 0
        assert abs(alpha - res.x[n]) < 1e-7, (alpha, res.x[n])
        if name == "panda" and alpha > 1.0 - 1e-9:
            # redundant arm, full twist reachable: the minimum-norm optimal v is returned
            assert np.dot(v, v) <= np.dot(res.x[:n], res.x[:n]) + 1e-9


def test_diff_ik_with_ee_offset(ur3e):
    rng = np.random.default_rng(5)
    lb, ub = (np.array(v) for v in ur3e.joint_limits())
    x0 = rng.uniform(lb, ub)
    off = np.eye(4); off[:3, 3] = [0.05, -0.02, 0.1]
    V = rng.random(6)
    alpha, v = ur3e.diff_ik(x0.tolist(), V.tolist(), [1.0] * 6, off.tolist())
    fk = np.array(ur3e.fk(x0.tolist(), off.tolist()))
    J = np.array(ur3e.joint_jacobian(x0.tolist(), off.tolist()))
    JW = np.vstack([fk[:3, :3] @ J[:3], fk[:3, :3] @ J[3:]])
    assert np.allclose(JW @ np.array(v), alpha * V, atol=1e-8)
    assert 0.0 <= alpha <= 1.0


def test_multi_device_sharding_gives_the_single_device_answers():
    """optik_robot_set_devices: restart ranges (ik) and targets (ik_batch) spread over several
    device contexts -- here the one GPU listed twice, which runs the same sharding, threads and
    host-side min as two GPUs would -- return exactly what one device returns."""
    from optik_amd import Robot, SolverConfig
    path = os.path.join(ROBOTS, "panda.urdf")
    one = Robot.from_urdf_file(path, "panda_link0", "panda_link8")
    two = Robot.from_urdf_file(path, "panda_link0", "panda_link8")
    two.set_devices([0, 0])
    one.set_parallelism(1)
    two.set_parallelism(1)
    assert one.num_devices() == 1 and two.num_devices() == 2
    rng = np.random.default_rng(21)
    lb, ub = (np.array(v) for v in one.joint_limits())
    targets = [np.array(one.fk(rng.uniform(lb, ub))) for _ in range(9)]
    x0s = rng.uniform(lb, ub, size=(9, 7))
    # Quality over 100 000 restarts: first launch on one context, then rounds cut in two parts
    cfg = SolverConfig(solution_mode="quality", max_time=0.0, max_restarts=100_000)
    a = one.ik(cfg, targets[0], x0s[0].tolist(), return_index=True)
    b = two.ik(cfg, targets[0], x0s[0].tolist(), return_index=True)
    assert a is not None and a == b
    # Speed on an unreachable-then-reachable pair: the winner index is the lowest success
    cfg = SolverConfig(solution_mode="speed", max_time=0.0, max_restarts=5000)
    for t in range(3):
        assert one.ik(cfg, targets[t], x0s[t].tolist(), return_index=True) == \
            two.ik(cfg, targets[t], x0s[t].tolist(), return_index=True)
    # ik_batch: 9 targets -> parts of 4 and 5
    for mode in ("speed", "quality"):
        cfg = SolverConfig(solution_mode=mode, max_time=0.0, max_restarts=300)
        assert one.ik_batch(cfg, targets, x0s) == two.ik_batch(cfg, targets, x0s)
    # a long Quality call: after 512 + 65 536 (x 2) restarts on the solve kernel the rounds of 1 M
    # restarts per context are one launch on each context; same winner, same numbers
    cfg = SolverConfig(solution_mode="quality", max_time=0.0, max_restarts=2_400_000)
    a = one.ik(cfg, targets[1], x0s[1].tolist(), return_index=True)
    b = two.ik(cfg, targets[1], x0s[1].tolist(), return_index=True)
    assert a is not None and a == b
    with pytest.raises(RuntimeError):
        two.set_devices([0])  # only before the first GPU call


def test_ik_batch_honours_max_time_inside_a_launch(panda):
    """lib.rs:308: the time-out is checked at every evaluation, so a batch whose round would
    run for tens of ms returns close to max_time (the launch's waves abandon what is in flight)."""
    from optik_amd import SolverConfig
    far = np.eye(4)
    far[:3, 3] = 50.0  # unreachable: every restart runs until it stalls
    T = 4096
    rng = np.random.default_rng(4)
    lb, ub = (np.array(v) for v in panda.joint_limits())
    x0s = rng.uniform(lb, ub, size=(T, 7))
    cfg_free = SolverConfig(max_time=0.0, max_restarts=256)
    panda.ik_batch(cfg_free, [far] * 64, x0s[:64])  # warm-up: workspace allocation, module load
    t0 = time.perf_counter()
    res = panda.ik_batch(cfg_free, [far] * T, x0s)
    full = time.perf_counter() - t0
    assert all(r is None for r in res)
    budget = max(full / 8.0, 0.004)
    t0 = time.perf_counter()
    res = panda.ik_batch(SolverConfig(max_time=budget, max_restarts=256), [far] * T, x0s)
    took = time.perf_counter() - t0
    assert all(r is None for r in res)
    assert took < budget + max(0.25 * full, 0.01), (took, budget, full)


def test_pose_validation_matches_parse_pose(ur3e):
    """optik-py parse_pose: a matrix that is not an isometry is rejected with the reference's
    message; its own fk output round-trips."""
    from optik_amd import SolverConfig
    x = [0.1, -0.4, 0.3, 0.2, -0.1, 0.5]
    good = np.array(ur3e.fk(x))
    cfg = SolverConfig(max_time=0.0, max_restarts=50)
    assert ur3e.ik(cfg, good.tolist(), x) is not None
    for bad in (good * np.array([[1.0 + 1e-9] * 4] * 3 + [[1.0] * 4]),      # scaled rotation block
                np.vstack([good[:3], [0.0, 0.0, 1e-3, 1.0]]),              # bottom row
                np.diag([1.0, 1.0, -1.0, 1.0])):                           # reflection
        with pytest.raises(ValueError, match="invalid target transform specified"):
            ur3e.ik(cfg, bad.tolist(), x)
    with pytest.raises(ValueError, match="invalid target transform specified"):
        ur3e.fk(x, ee_offset=(good * 2.0).tolist())


def test_diff_ik_six_dof_solution_is_the_unique_ray(ur3e):
    """Where the reference's diff_ik can run at all (n = 6: lib.rs:196-197 sizes the equality
    cone with n for a 6-row block) a non-singular Jacobian leaves one ray v = alpha J^-1 V, so the
    LP has a unique solution -- what any solver, Clarabel included, must return:
    alpha = min(1, min_i v_max_i / |w_i|), v = alpha w, w = J_W^-1 V."""
    rng = np.random.default_rng(77)
    lb, ub = (np.array(v) for v in ur3e.joint_limits())
    for trial in range(30):
        x0 = rng.uniform(lb, ub)
        V = rng.normal(size=6) * (0.05 if trial % 3 == 0 else 1.0)
        v_max = rng.uniform(0.2, 2.0, size=6)
        JW = _world_jacobian(ur3e, x0.tolist())
        if np.linalg.cond(JW) > 1e6:
            continue
        w = np.linalg.solve(JW, V)
        alpha_ref = min(1.0, float(np.min(v_max / np.abs(w))))
        alpha, v = ur3e.diff_ik(x0.tolist(), V.tolist(), v_max.tolist())
        assert abs(alpha - alpha_ref) < 1e-9
        np.testing.assert_allclose(v, alpha_ref * w, atol=1e-8, rtol=0)


def test_long_calls_move_to_big_rounds_with_the_same_answer(panda):
    """Throughput-bound calls run in rounds of 1 M restarts (single launches of the lane-per-restart form) -- Quality with a restart budget of ~260 000 or more from index 0, any call still running
    after its first two launches: Quality over 300 000 restarts returns the restart a solve-kernel launch around it
    selects; an unreachable Speed target comes back None after all of them; max_time ends a
    round early."""
    import torch
    from optik_amd import SolverConfig
    rng = np.random.default_rng(77)
    lb, ub = (np.array(v) for v in panda.joint_limits())
    R = 300_000
    cfg = SolverConfig(solution_mode="quality", max_time=0.0, max_restarts=R)
    for _ in range(8):  # (a target whose winner lies beyond the first two solve-kernel launches of a timed call)
        tgt = np.array(panda.fk(rng.uniform(lb, ub)))
        x0 = rng.uniform(lb, ub)
        x, c, idx = panda.ik(cfg, tgt, x0.tolist(), return_index=True)
        if idx >= 512 + 65536:
            break
    assert idx >= 512 + 65536
    hc = panda.hip_chain()
    t7 = torch.tensor(_mat_to_pose7(tgt)[None], dtype=torch.float64, device="cuda")
    ref = hc.ik_batch(cfg.to_c(), t7, torch.tensor(x0[None], device="cuda"), idx - 2000, idx + 2000,
                      per_restart=False)
    torch.cuda.synchronize()
    # the same restart wins its neighbourhood, with the same numbers up to the 4x4 -> pose conversion
    assert int(ref["win_idx"][0]) == idx
    np.testing.assert_allclose(ref["win_x"][0].cpu().numpy(), x, atol=1e-6, rtol=0)
    far = np.eye(4)
    far[:3, 3] = 50.0
    assert panda.ik(SolverConfig(max_time=0.0, max_restarts=200_000), far, x0.tolist()) is None
    t0 = time.perf_counter()
    assert panda.ik(SolverConfig(max_time=0.03, max_restarts=50_000_000), far, x0.tolist()) is None
    assert time.perf_counter() - t0 < 0.13


def test_robot_is_reentrant(panda):
    """Robot::ik takes &self and is called from many host threads at once (lib.rs:241; SURVEY 8b
    "Threading"): concurrent ik / ik_batch / fk calls on ONE robot -- short ones on the solve
    kernel, a long Quality call of 1 M-restart rounds, a batch -- return what they return alone."""
    import threading
    from optik_amd import SolverConfig
    rng = np.random.default_rng(31)
    lb, ub = (np.array(v) for v in panda.joint_limits())
    qs = rng.uniform(lb, ub, size=(12, 7))
    targets = [np.array(panda.fk(q)) for q in qs]
    x0s = rng.uniform(lb, ub, size=(12, 7))
    speed = SolverConfig(max_time=0.0, max_restarts=2000)
    long_q = SolverConfig(solution_mode="quality", max_time=0.0, max_restarts=250_000)
    batch_cfg = SolverConfig(max_time=0.0, max_restarts=64)
    want_speed = [panda.ik(speed, targets[t], x0s[t].tolist(), return_index=True) for t in range(12)]
    want_long = panda.ik(long_q, targets[0], x0s[0].tolist(), return_index=True)
    want_batch = panda.ik_batch(batch_cfg, targets, x0s)
    want_fk = [panda.fk(q) for q in qs]
    errors = []

    def check(name, fn, want, reps):
        try:
            for _ in range(reps):
                got = fn()
                if got != want:
                    errors.append(f"{name}: differs")
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(f"{name}: {e!r}")

    threads = [threading.Thread(target=check, args=(f"speed{t}", (lambda t=t: panda.ik(
        speed, targets[t], x0s[t].tolist(), return_index=True)), want_speed[t], 15)) for t in range(6)]
    threads.append(threading.Thread(target=check, args=("long", lambda: panda.ik(
        long_q, targets[0], x0s[0].tolist(), return_index=True), want_long, 3)))
    threads.append(threading.Thread(target=check, args=("batch", lambda: panda.ik_batch(batch_cfg, targets, x0s),
                                                        want_batch, 10)))
    threads.append(threading.Thread(target=check, args=("fk", lambda: [panda.fk(q) for q in qs], want_fk, 10)))
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads), "deadlock"
    assert not errors, errors


def test_robots_on_one_device_from_many_threads():
    """Several robots -- different chains, used from different threads at once -- share one device: each returns what
    it returns alone, and the launch workspaces stay small (rounds 1-4 kept a ~1 GB slot pool per device here)."""
    import threading
    import torch
    from optik_amd import Robot, SolverConfig
    specs = [("panda.urdf", "panda_link0", "panda_link8"), ("ur10.urdf", "base_link", "ee_link"),
             ("panda.urdf", "panda_link0", "panda_link5"), ("panda.urdf", "panda_link0", "panda_hand")]
    robots = [Robot.from_urdf_file(os.path.join(ROBOTS, f), b, e) for f, b, e in specs]
    rng = np.random.default_rng(9)
    cfg = SolverConfig(solution_mode="quality", max_time=0.0, max_restarts=150_000)
    cases = []
    for r in robots:
        lb, ub = (np.array(v) for v in r.joint_limits())
        cases.append((np.array(r.fk(rng.uniform(lb, ub))), rng.uniform(lb, ub)))
    free0, _ = torch.cuda.mem_get_info()
    want = [r.ik(cfg, t, x0, return_index=True) for r, (t, x0) in zip(robots, cases)]
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 0.5 * 2**30, (free0 - free1) / 2**30  # per-restart keys / points of 150 000 restarts x 4
    got = [None] * len(robots)

    def run(k):
        for _ in range(3):
            got[k] = robots[k].ik(cfg, cases[k][0], cases[k][1], return_index=True)

    threads = [threading.Thread(target=run, args=(k,)) for k in range(len(robots))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads), "deadlock"
    assert got == want


def test_set_parallelism_selects_find_any(oracle, chains):
    """set_parallelism(n > 1): SolutionMode::Speed stops at the first success of ANY restart
    (rayon's find_any with several threads, lib.rs:409-412) -- the returned restart is then not
    fixed, but it is a restart that succeeds, and its x / f are exactly what that restart
    computes on its own (checked against the oracle by index)."""
    from optik_amd import Robot, SolverConfig
    r = Robot.from_urdf_file(os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link8")
    # (parallelism never set = the reference's default pool of every core: the same rule; set
    # explicitly for the second half of the loop below)
    _, ch = chains["panda"]
    rng = np.random.default_rng(52)
    lb, ub = (np.array(v) for v in r.joint_limits())
    cfg = SolverConfig(max_time=0.0, max_restarts=4000)
    for trial in range(6):
        if trial == 3:
            r.set_parallelism(8)
        tgt = np.array(r.fk(rng.uniform(lb, ub)))
        x0 = rng.uniform(lb, ub)
        x, f, idx = r.ik(cfg, tgt, x0.tolist(), return_index=True)
        ref = oracle.solve_restart(ch, oracle.make_config("speed"), _mat_to_pose7(tgt), x0, int(idx))
        # (the target reaches the kernels through the host layer's 4x4 -> pose conversion, the
        # oracle through the test's: agreement to roundoff, as in test_determinism_and_oracle_agreement)
        assert ref.success and abs(ref.f - f) < 1e-9
        np.testing.assert_allclose(x, np.array(ref.x[:7]), atol=1e-6, rtol=0)
    # ik_batch with the same rule
    targets = [np.array(r.fk(rng.uniform(lb, ub))) for _ in range(16)]
    x0s = rng.uniform(lb, ub, size=(16, 7))
    out = r.ik_batch(SolverConfig(max_time=0.0, max_restarts=256), targets, x0s)
    for t, o in enumerate(out):
        assert o is not None and o[1] < 1e-6
        np.testing.assert_allclose(np.array(r.fk(o[0]))[:3, 3], targets[t][:3, 3], atol=2e-3)


def test_first_success_calls_return_early_and_leave_a_consistent_chain(oracle, chains):
    """Under the first-success rule a single ik() returns when the first restart has succeeded and its launch ends
    behind the call (ik_capi.hip: optik_hip_ik_host, claim block).  Back-to-back calls, a call without any
    solution (the launch then ends the ordinary way) and a deterministic call right behind an early return all give
    answers the oracle reproduces by restart index."""
    from optik_amd import Robot, SolverConfig
    r = Robot.from_urdf_file(os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link8")
    _, ch = chains["panda"]
    rng = np.random.default_rng(77)
    lb, ub = (np.array(v) for v in r.joint_limits())
    cfg = SolverConfig(max_time=0.0, max_restarts=2000)
    cases = [(np.array(r.fk(rng.uniform(lb, ub))), rng.uniform(lb, ub)) for _ in range(24)]
    got = [r.ik(cfg, t, x0.tolist(), return_index=True) for t, x0 in cases]  # back to back, nothing in between
    for (t, x0), g in zip(cases, got):
        assert g is not None
        x, f, idx = g
        ref = oracle.solve_restart(ch, oracle.make_config("speed"), _mat_to_pose7(t), x0, int(idx))
        assert ref.success and abs(ref.f - f) < 1e-9
        np.testing.assert_allclose(x, np.array(ref.x[:7]), atol=1e-6, rtol=0)
    # out of reach: no restart succeeds, the call ends with its launches and says so
    far = np.eye(4)
    far[:3, 3] = [3.0, 0.0, 0.5]
    assert r.ik(SolverConfig(max_time=0.0, max_restarts=600), far, cases[0][1].tolist()) is None
    # ... and the next call is served as usual
    assert r.ik(cfg, cases[0][0], cases[0][1].tolist()) is not None
    # a deterministic call right behind an early return (its launch queues behind the one left running)
    t, x0 = cases[1]
    r.ik(cfg, t, x0.tolist())
    r.set_parallelism(1)
    x, f, idx = r.ik(cfg, t, x0.tolist(), return_index=True)
    ref = oracle.ik(ch, oracle.make_config("speed"), _mat_to_pose7(t), x0, 0, 2000, n_threads=4, early_exit=False,
                    per_restart=True)
    assert ref["found"] and int(idx) == int(ref["winner"])


def test_early_return_then_a_launch_on_another_stream(oracle, chains):
    """A launch on a stream of its own shares the chain's workspace with the launch an early-returned call left
    running on the null stream: it waits for that one (ik_capi.hip: claim_pending) and gives the oracle's bits."""
    import torch
    from optik_amd import _native as nat
    from optik_amd import device
    d, ch = chains["panda"]
    hc = device.HipChain(**d)
    rng = np.random.default_rng(78)
    cfg = nat.make_config(solution_mode="speed")
    side = torch.cuda.Stream()
    for trial in range(4):
        _, tgt = oracle.fk(ch, rng.uniform(d["lb"], d["ub"]))
        x0 = rng.uniform(d["lb"], d["ub"])
        # a host call under the first-success rule: returns on the first success
        early = hc.ik_host(cfg, tgt[None], x0[None], 0, 512, flags=nat.IK_EARLY_EXIT | nat.IK_FIND_ANY)
        assert int(early["win_idx"][0]) >= 0
        with torch.cuda.stream(side):
            out = hc.ik_batch(cfg, torch.tensor(tgt[None], device="cuda:0"), torch.tensor(x0[None], device="cuda:0"), 0, 300)
        side.synchronize()
        ref = oracle.ik(ch, oracle.make_config(solution_mode="speed"), tgt, x0, 0, 300, n_threads=2, early_exit=False,
                        per_restart=True)
        assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
        assert_bit_equal(out["x"].cpu().numpy(), ref["xs"].T.copy())


def test_early_return_then_a_staged_many_target_host_call(oracle, chains):
    """optik_hip_ik_host stages more than 16 targets through pinned block 0; a first-success call that returned early may
    have left a launch running whose selection kernel still writes its winner into that block -- inside the region the
    targets are staged in (poses 2-3 of the next call were overwritten, silently).  The staged call now waits
    for that launch: every target's winner equals the winner of the same call made on a fresh chain."""
    from optik_amd import _native as nat
    from optik_amd import device
    d, ch = chains["panda"]
    rng = np.random.default_rng(91)
    cfg = nat.make_config(solution_mode="speed")
    T = 24
    tg = np.array([oracle.fk(ch, rng.uniform(d["lb"], d["ub"]))[1] for _ in range(T)])
    x0 = rng.uniform(d["lb"], d["ub"], size=(T, 7))
    fresh = device.HipChain(**d)
    want = fresh.ik_host(cfg, tg, x0, 0, 64, flags=nat.IK_EARLY_EXIT)
    hc = device.HipChain(**d)
    for trial in range(6):
        # (both pinned blocks get used by early-returning calls, so the next staged call meets one still in flight)
        for k in range(2):
            early = hc.ik_host(cfg, tg[k:k + 1], x0[k:k + 1], 0, 1024, flags=nat.IK_EARLY_EXIT | nat.IK_FIND_ANY)
            assert int(early["win_idx"][0]) >= 0
        got = hc.ik_host(cfg, tg, x0, 0, 64, flags=nat.IK_EARLY_EXIT)
        assert np.array_equal(got["win_idx"], want["win_idx"]), trial
        assert_bit_equal(got["win_x"], want["win_x"], "staged call after early returns")


def test_null_config_through_the_c_abi_fails_cleanly(chains):
    """optik_hip_ik_host reads cfg->solution_mode before the launch's own argument check: a NULL config used to be a
    segfault; it is EINVAL with a message like every other bad argument."""
    import ctypes as C
    from optik_amd import _native as nat
    from optik_amd import device
    d, _ = chains["panda"]
    hc = device.HipChain(**d)
    dp = C.POINTER(C.c_double)
    tg = np.zeros((1, 7)); tg[0, 6] = 1.0
    x0 = np.zeros((1, 7))
    out = np.zeros(16)
    idx = np.zeros(1, dtype=np.uint64)
    rc = nat.lib().optik_hip_ik_host(hc._h, None, tg.ctypes.data_as(dp), x0.ctypes.data_as(dp), 1, None, 0, 64, 0, 0.0,
                                     out.ctypes.data_as(dp), out[8:].ctypes.data_as(dp),
                                     idx.ctypes.data_as(C.POINTER(C.c_uint64)), out[9:].ctypes.data_as(dp))
    assert rc != 0


def test_first_success_calls_from_many_threads_and_robots(oracle, chains):
    """Early-return calls (first-success rule) from several host threads on several robots of one device: every
    launch one of them leaves behind queues in front of the others' launches on the null stream.  Every answer is
    a restart that succeeds on its own (the oracle's, by index); nothing hangs."""
    import threading
    from optik_amd import Robot, SolverConfig
    _, ch = chains["panda"]
    robots = [Robot.from_urdf_file(os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link8") for _ in range(3)]
    lb, ub = (np.array(v) for v in robots[0].joint_limits())
    cfg = SolverConfig(max_time=0.0, max_restarts=2000)
    errors, results = [], []
    lock = threading.Lock()

    def work(k):
        try:
            rng = np.random.default_rng(500 + k)
            r = robots[k % len(robots)]
            for _ in range(25):
                tgt = np.array(r.fk(rng.uniform(lb, ub)))
                x0 = rng.uniform(lb, ub)
                got = r.ik(cfg, tgt, x0.tolist(), return_index=True)
                with lock:
                    results.append((tgt, x0, got))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=180)
    assert not any(th.is_alive() for th in threads), "deadlock"
    assert not errors, errors
    assert len(results) == 150
    for tgt, x0, got in results[::5]:
        assert got is not None
        x, f, idx = got
        ref = oracle.solve_restart(ch, oracle.make_config("speed"), _mat_to_pose7(tgt), x0, int(idx))
        assert ref.success and abs(ref.f - f) < 1e-9
        np.testing.assert_allclose(x, np.array(ref.x[:7]), atol=1e-6, rtol=0)
