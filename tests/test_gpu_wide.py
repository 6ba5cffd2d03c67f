"""-m gpu: chains with 9 .. 16 joint positions (optik_amd/csrc/ik_wide.hpp -- the general kernels, joint
count at run time) against the CPU oracle, bit for bit, through the C ABI.  The reference accepts any
chain length (/root/reference/crates/optik/src/kinematics.rs:107-110); the tuned solvers stop at 8."""
import os

import numpy as np
import pytest

from gpu_util import assert_bit_equal, make_targets

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

WIDE = ["arm9", "arm10", "arm12", "arm16"]


@pytest.fixture(scope="module")
def dev():
    from optik_amd import device
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return device


@pytest.fixture(scope="module")
def hip_chains(dev, chains):
    return {name: dev.HipChain(**chains[name][0]) for name in WIDE}


def _oracle_all(oracle, ch, cfg_kw, tgt, x0, begin, end):
    return oracle.ik(ch, oracle.make_config(**cfg_kw), tgt, x0, begin, end, n_threads=4, early_exit=False,
                     per_restart=True)


@pytest.mark.parametrize("robot", WIDE)
def test_restart_seeds_bit_exact(dev, oracle, chains, hip_chains, robot):
    """More than eight joints draw from the second ChaCha8 block of the restart's stream
    (lib.rs:86-91, 358-370: one next_u64 per joint)."""
    _, ch = chains[robot]
    for first, count in ((1, 2000), (4194304 - 5, 8), (2**32 - 3, 8), (2**40 + 7, 8)):
        got = hip_chains[robot].seed_batch(first, count).cpu().numpy()
        want = np.array([oracle.restart_seed(ch, i) for i in range(first, first + count)]).T
        assert_bit_equal(got, want, f"restart seeds from {first}")


@pytest.mark.parametrize("robot", WIDE)
@pytest.mark.parametrize("weights", ["default", "reference_test", "identity_quirk"])
def test_objective_and_gradient_bit_exact(dev, oracle, chains, hip_chains, robot, weights):
    from optik_amd import _native as nat
    d, ch = chains[robot]
    n = len(d["lb"])
    wl, wa = {"default": ((1, 1, 1), (1, 1, 1)),
              "reference_test": ((0.0, 5.0, 0.25), (0.005, 1.0, 0.99)),  # tests/test_gradient.rs:37-38
              "identity_quirk": ((1, 0, 0), (1, 1, 1))}[weights]
    rng = np.random.default_rng(7)
    B = 1500
    q = rng.uniform(d["lb"], d["ub"], size=(B, n))
    quat = rng.normal(size=4)
    quat /= np.linalg.norm(quat)
    tgt = np.concatenate([rng.uniform(-0.5, 0.5, 3), quat])
    ee_off = None
    if weights == "reference_test":
        eq = rng.normal(size=4)
        eq /= np.linalg.norm(eq)
        ee_off = np.concatenate([rng.uniform(-0.1, 0.1, 3), eq])
    cfg = nat.make_config(linear_weight=wl, angular_weight=wa)
    f, g = hip_chains[robot].eval_batch(torch.tensor(q.T.copy(), device="cuda"), tgt, cfg, ee_off)
    f, g = f.cpu().numpy(), g.cpu().numpy()
    ee_pose = oracle.Pose.make(ee_off[:3], ee_off[3:]) if ee_off is not None else None
    fr, gr = np.empty(B), np.empty((n, B))
    for i in range(B):
        fr[i], gr[:, i] = oracle.eval_fg(ch, tgt, q[i], wl, wa, ee_offset=ee_pose)
    assert_bit_equal(f, fr, "objective")
    assert_bit_equal(g, gr, "gradient")


@pytest.mark.parametrize("robot", ["arm10", "arm16"])
def test_fk_and_jacobian_bit_exact(dev, oracle, chains, hip_chains, robot):
    d, ch = chains[robot]
    n = len(d["lb"])
    rng = np.random.default_rng(3)
    B = 500
    q = rng.uniform(d["lb"], d["ub"], size=(B, n))
    pose, jac = hip_chains[robot].fk_batch(torch.tensor(q.T.copy(), device="cuda"), jacobian=True)
    pose, jac = pose.cpu().numpy(), jac.cpu().numpy()
    pr, jr = np.empty((7, B)), np.empty((6 * n, B))
    for i in range(B):
        _, ee = oracle.fk(ch, q[i])
        pr[:, i] = ee
        jr[:, i] = oracle.joint_jacobian(ch, q[i]).T.ravel()  # column-major 6 x n
    assert_bit_equal(pose, pr, "fk pose")
    assert_bit_equal(jac, jr, "jacobian")


@pytest.mark.parametrize("robot,tol_f", [("arm9", 1e-6), ("arm10", 1e-8), ("arm12", 1e-10), ("arm16", 1e-8)])
@pytest.mark.parametrize("mode", ["speed", "quality"])
@pytest.mark.parametrize("form", ["lds", "lds_one_lane", "hbm"])
def test_every_restart_bit_exact(dev, oracle, chains, hip_chains, robot, tol_f, mode, form, monkeypatch):
    """Two targets, restarts 0..R-1 each: status, evaluation count, returned x and f of EVERY restart
    equal the oracle's, and so does the selected winner (Speed: lowest index; Quality: nearest the seed).
    Both forms of the general solver: one restart per wave with its arrays in LDS (launches that leave
    waves to spare: a single ik() call's rounds) and several per wave with the HBM workspace."""
    from optik_amd import _native as nat
    d, ch = chains[robot]
    rng = np.random.default_rng(11)
    T, R = 2, {"lds": 500, "lds_one_lane": 300, "hbm": 2600}[form]
    # (the scheduler picks the cooperative LDS form: the wave's 64 lanes on its one restart; "lds_one_lane" is the
    # same layout with one lane doing all the work, "hbm" a restart per lane with the HBM workspace)
    tg, x0 = make_targets(oracle, d, ch, rng, T)
    kw = dict(solution_mode=mode, tol_f=tol_f)
    with nat.options(wide_form={"lds": 0, "hbm": 1, "lds_one_lane": 2}[form]):
        out = hip_chains[robot].ik_batch(nat.make_config(**kw), torch.tensor(tg, device="cuda"),
                                         torch.tensor(x0, device="cuda"), 0, R)
        torch.cuda.synchronize()
    assert (hip_chains[robot].last_launch()["lds_bytes"] > 8192) == (form != "hbm")
    st = out["status"].cpu().numpy().reshape(T, R)
    ev = out["evals"].cpu().numpy().reshape(T, R)
    fs = out["f"].cpu().numpy().reshape(T, R)
    xs = out["x"].cpu().numpy()
    seen = set()
    for t in range(T):
        ref = _oracle_all(oracle, ch, kw, tg[t], x0[t], 0, R)
        assert np.array_equal(st[t], ref["status"]), np.argwhere(st[t] != ref["status"])[:10]
        assert np.array_equal(ev[t], ref["evals"])
        assert_bit_equal(fs[t], ref["fs"], f"per-restart f, target {t}")
        assert_bit_equal(xs[:, t * R:(t + 1) * R], ref["xs"].T, f"per-restart x, target {t}")
        assert int(out["win_idx"].cpu()[t]) == (ref["winner"] if ref["found"] else -1)
        if ref["found"]:
            assert_bit_equal(out["win_x"].cpu().numpy()[t], ref["x"], "winner x")
        seen.update(np.unique(ref["status"]).tolist())
    assert len(seen) >= 2  # more than one way of ending is exercised


@pytest.mark.parametrize("tols", [dict(tol_df=1e-18, tol_dx=1e-9), dict(tol_df=-1.0, tol_dx=-1.0)])
def test_tight_tolerances_and_other_endings(dev, oracle, chains, hip_chains, tols):
    """tol_f = 0: no restart can end on stopval; they end on the ftol / xtol branches of NLopt's driver
    (lib.rs:376-379) or, with both disabled, on NLopt's zero-step rule -- the far side of the SLSQP
    driver from the usual exit, including line searches that run to their last trial."""
    from optik_amd import _native as nat
    d, ch = chains["arm10"]
    rng = np.random.default_rng(5)
    tg, x0 = make_targets(oracle, d, ch, rng, 1)
    kw = dict(solution_mode="quality", tol_f=0.0, **tols)
    R = 400
    out = hip_chains["arm10"].ik_batch(nat.make_config(**kw), torch.tensor(tg, device="cuda"),
                                       torch.tensor(x0, device="cuda"), 0, R)
    torch.cuda.synchronize()
    ref = _oracle_all(oracle, ch, kw, tg[0], x0[0], 0, R)
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    assert np.array_equal(out["evals"].cpu().numpy(), ref["evals"])
    assert_bit_equal(out["f"].cpu().numpy(), ref["fs"], "per-restart f")
    assert_bit_equal(out["x"].cpu().numpy(), ref["xs"].T, "per-restart x")
    assert 2 not in ref["status"]


def test_restart_ranges_compose(dev, oracle, chains, hip_chains):
    """The multi-GPU partition (contiguous index ranges) on a wide chain and a ragged range: the same bits."""
    from optik_amd import _native as nat
    d, ch = chains["arm12"]
    hc = hip_chains["arm12"]
    rng = np.random.default_rng(9)
    tg, x0 = make_targets(oracle, d, ch, rng, 1)
    cfg = nat.make_config(solution_mode="quality", tol_f=1e-8)
    tgd, x0d = torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda")
    full = hc.ik_batch(cfg, tgd, x0d, 0, 500)
    a = hc.ik_batch(cfg, tgd, x0d, 0, 133)
    b = hc.ik_batch(cfg, tgd, x0d, 133, 500)
    torch.cuda.synchronize()
    for k in ("f", "status", "evals"):
        assert torch.equal(torch.cat([a[k], b[k]]), full[k])
    assert torch.equal(torch.cat([a["x"], b["x"]], dim=1), full["x"])


def test_early_exit_speed_winner(dev, oracle, chains, hip_chains):
    """Speed with the reference's should_exit (lib.rs:269, 308, 382-384) in its deterministic reading: the
    winner is the lowest successful index whatever was abandoned on the way."""
    from optik_amd import _native as nat
    d, ch = chains["arm10"]
    rng = np.random.default_rng(13)
    T, R = 6, 256
    tg, x0 = make_targets(oracle, d, ch, rng, T)
    kw = dict(solution_mode="speed", tol_f=1e-8)
    out = hip_chains["arm10"].ik_batch(nat.make_config(**kw), torch.tensor(tg, device="cuda"),
                                       torch.tensor(x0, device="cuda"), 0, R, flags=nat.IK_EARLY_EXIT)
    torch.cuda.synchronize()
    win = out["win_idx"].cpu().numpy()
    wx = out["win_x"].cpu().numpy()
    for t in range(T):
        ref = _oracle_all(oracle, ch, kw, tg[t], x0[t], 0, R)
        assert win[t] == (ref["winner"] if ref["found"] else -1)
        if ref["found"]:
            assert_bit_equal(wx[t], ref["x"], f"winner x target {t}")


def test_early_exit_many_concurrent_writers(dev, oracle, chains, hip_chains):
    """Stress of the early-exit word under the cooperative form (a restart per wave, 64 lanes sharing its
    state in LDS): hundreds of targets, each with thousands of restarts in flight whose successes land on
    first_success concurrently.  The stop decision must be ONE per wave (lane 0's): every published record is
    then a complete one -- the winner is a solved restart with in-limit joints whose FK meets the target -- and
    the winners are the same from run to run and equal the no-early-exit run's lowest solved index."""
    from optik_amd import _native as nat
    d, ch = chains["arm10"]
    hc = hip_chains["arm10"]
    rng = np.random.default_rng(29)
    T, R = 192, 512
    tg, x0 = make_targets(oracle, d, ch, rng, T)
    tgd, x0d = torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda")
    cfg = nat.make_config(solution_mode="speed", tol_f=1e-6)
    full = hc.ik_batch(cfg, tgd, x0d, 0, R)
    torch.cuda.synchronize()
    ok = (full["status"] == nat.RES_STOPVAL).view(T, R)
    want = torch.where(ok.any(1), ok.double().argmax(1), torch.full((T,), -1, device="cuda"))
    lb, ub = torch.tensor(d["lb"], device="cuda"), torch.tensor(d["ub"], device="cuda")
    for flags in (nat.IK_EARLY_EXIT, nat.IK_EARLY_EXIT | nat.IK_RESTART_MAJOR):
        runs = []
        for _ in range(3):
            out = hc.ik_batch(cfg, tgd, x0d, 0, R, flags=flags, per_restart=False)
            torch.cuda.synchronize()
            runs.append(out)
            assert torch.equal(out["win_idx"], want)
            solved = out["win_idx"] >= 0
            wx = out["win_x"][solved]
            assert (wx >= lb).all() and (wx <= ub).all()
            pose = hc.fk_batch(wx.T.contiguous())
            t = tgd[solved].T
            dq = torch.minimum((pose[3:] - t[3:]).abs().amax(0), (pose[3:] + t[3:]).abs().amax(0))
            assert torch.maximum((pose[:3] - t[:3]).abs().amax(0), dq).max().item() < 2e-3
            assert (out["win_f"][solved] < 1e-6).all()
        assert torch.equal(runs[0]["win_x"], runs[1]["win_x"]) and torch.equal(runs[1]["win_x"], runs[2]["win_x"])


def test_robot_api_ten_joints(dev, oracle, chains):
    """The reference's own ik property (tests/test_ik.rs:91-130: FK(ik(T)) == T within 1e-6) through the
    host API on a 10-joint URDF; the winning restart is the oracle's and the joint angles agree to 1e-6
    (the target enters through the host layer's 4x4 -> pose conversion, hence not bit for bit)."""
    from conftest import ROBOT_SPECS
    from optik_amd import Robot, SolverConfig
    from test_gpu_robot_api import _mat_to_pose7
    path, base, ee = ROBOT_SPECS["arm10"]
    robot = Robot.from_urdf_file(path, base, ee)
    robot.set_parallelism(1)
    assert robot.num_positions() == 10
    d, ch = chains["arm10"]
    rng = np.random.default_rng(21)
    cfg = SolverConfig(solution_mode="speed", max_time=0.0, max_restarts=64, tol_f=1e-14)
    for _ in range(4):
        q = rng.uniform(d["lb"], d["ub"])
        tgt = np.array(robot.fk(list(q)))
        x0 = rng.uniform(d["lb"], d["ub"])
        sol = robot.ik(cfg, tgt, list(x0), return_index=True)
        assert sol is not None
        x, f, idx = sol
        got = np.array(robot.fk(list(x)))
        assert np.abs(got - tgt).max() < 1e-6
        ref = oracle.ik(ch, oracle.make_config(solution_mode="speed", tol_f=1e-14), _mat_to_pose7(tgt), x0, 0, 64)
        assert ref["found"] and ref["winner"] == idx
        np.testing.assert_allclose(x, ref["x"], atol=1e-6, rtol=0)
    # the Jacobian of a wide chain through the host API; diff_ik is refused (the reference's only runs for n = 6)
    J = np.array(robot.joint_jacobian(list(q)))
    assert_bit_equal(J, oracle.joint_jacobian(ch, q), "jacobian")
    with pytest.raises(Exception):
        robot.diff_ik(list(q), [0.1, 0, 0, 0, 0, 0], [1.0] * 10)
    # many targets at once (the host scheduler's rounds on the general kernel) == one call per target
    T = 24
    targets = [np.array(robot.fk(list(rng.uniform(d["lb"], d["ub"])))) for _ in range(T)]
    x0s = rng.uniform(d["lb"], d["ub"], size=(T, 10))
    batch = robot.ik_batch(cfg, targets, x0s)
    for t in range(T):
        assert batch[t] == robot.ik(cfg, targets[t], x0s[t].tolist())
    assert sum(b is not None for b in batch) >= T // 2


@pytest.mark.parametrize("seed", range(6))
def test_random_wide_chains_bit_exact(dev, oracle, seed):
    """Random 9 .. 16-joint arms (tools/gen_wide_robots.py's generator, other seeds; with and without a
    trailing fixed joint), weighted objective, an ee_offset: every restart against the oracle."""
    import sys
    from conftest import ROOT
    from optik_amd import _native as nat
    from oracle import urdf_chain
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_wide_robots
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(9, 17))
    tip = bool(rng.integers(0, 2))
    d = urdf_chain.chain_from_urdf(gen_wide_robots.arm(n, tip, 5000 + seed), "l0", f"l{n + (1 if tip else 0)}")
    ch = oracle.make_chain(**d)
    hc = dev.HipChain(**d)
    tg, x0 = make_targets(oracle, d, ch, rng, 1)
    eq = rng.normal(size=4)
    eq /= np.linalg.norm(eq)
    ee_off = np.concatenate([rng.uniform(-0.1, 0.1, 3), eq])
    kw = dict(solution_mode=["speed", "quality"][seed % 2], tol_f=10.0 ** -int(rng.integers(5, 11)),
              linear_weight=(1.0, 2.0, 0.5), angular_weight=(0.3, 1.0, 1.0))
    R = 192
    out = hc.ik_batch(nat.make_config(**kw), torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda"), 0, R,
                      ee_offset7=ee_off)
    torch.cuda.synchronize()
    ref = oracle.ik(ch, oracle.make_config(**kw), tg[0], x0[0], 0, R, n_threads=4, early_exit=False, per_restart=True,
                    ee_offset=oracle.Pose.make(ee_off[:3], ee_off[3:]))
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    assert np.array_equal(out["evals"].cpu().numpy(), ref["evals"])
    assert_bit_equal(out["f"].cpu().numpy(), ref["fs"], "per-restart f")
    assert_bit_equal(out["x"].cpu().numpy(), ref["xs"].T, "per-restart x")
    assert int(out["win_idx"].cpu()[0]) == (ref["winner"] if ref["found"] else -1)


def test_multi_device_sharding_on_a_wide_chain():
    """optik_robot_set_devices with a 10-joint chain: restart ranges (ik) and targets (ik_batch) spread over
    two device contexts (the one GPU listed twice: the same sharding, threads and host-side min as two GPUs)
    return exactly what one context returns."""
    from conftest import ROBOT_SPECS
    from optik_amd import Robot, SolverConfig
    path, base, ee = ROBOT_SPECS["arm10"]
    one = Robot.from_urdf_file(path, base, ee)
    two = Robot.from_urdf_file(path, base, ee)
    two.set_devices([0, 0])
    one.set_parallelism(1)
    two.set_parallelism(1)
    assert one.num_devices() == 1 and two.num_devices() == 2
    rng = np.random.default_rng(23)
    lb, ub = (np.array(v) for v in one.joint_limits())
    targets = [np.array(one.fk(rng.uniform(lb, ub))) for _ in range(7)]
    x0s = rng.uniform(lb, ub, size=(7, 10))
    cfg = SolverConfig(solution_mode="quality", max_time=0.0, max_restarts=20_000)
    a = one.ik(cfg, targets[0], x0s[0].tolist(), return_index=True)
    b = two.ik(cfg, targets[0], x0s[0].tolist(), return_index=True)
    assert a is not None and a == b
    cfg = SolverConfig(solution_mode="speed", max_time=0.0, max_restarts=3000)
    for t in range(3):
        assert one.ik(cfg, targets[t], x0s[t].tolist(), return_index=True) == \
            two.ik(cfg, targets[t], x0s[t].tolist(), return_index=True)
    for mode in ("speed", "quality"):
        cfg = SolverConfig(solution_mode=mode, max_time=0.0, max_restarts=200)
        assert one.ik_batch(cfg, targets, x0s) == two.ik_batch(cfg, targets, x0s)


def test_max_time_and_invalid_seed_on_a_wide_chain():
    """tests/test_ik.rs:24-43 on a 16-joint chain: an impossible goal with max_time = 0.05 s returns None
    within 0.05 +- 0.1 s (the general kernel's waves watch the device clock at every evaluation, lib.rs:308);
    a seed outside the joint limits is refused as the reference's assert does (lib.rs:248-256)."""
    import time
    from conftest import ROBOT_SPECS
    from optik_amd import Robot, SolverConfig
    path, base, ee = ROBOT_SPECS["arm16"]
    robot = Robot.from_urdf_file(path, base, ee)
    lb, ub = (np.array(v) for v in robot.joint_limits())
    mid = (0.5 * (lb + ub)).tolist()
    far = np.eye(4)
    far[:3, 3] = 100.0
    robot.fk(mid)  # warm the device up before timing
    robot.ik(SolverConfig(max_time=0.0, max_restarts=8), far, mid)
    t0 = time.perf_counter()
    sol = robot.ik(SolverConfig(max_time=0.05), far, mid)
    dt = time.perf_counter() - t0
    assert sol is None
    assert abs(dt - 0.05) < 0.1, dt
    bad = list(mid)
    bad[3] = ub[3] + 1.0
    with pytest.raises(Exception):
        robot.ik(SolverConfig(max_time=0.0, max_restarts=8), np.array(robot.fk(mid)), bad)


@pytest.mark.parametrize("robot", ["panda", "ur10", "panda_hand", "panda3", "arm8"])
@pytest.mark.parametrize("R", [400, 3000])
def test_general_solver_equals_the_tuned_solvers(dev, oracle, chains, robot, R, monkeypatch):
    """Option solve_kernel = general runs a chain of at most 8 joints on the run-time-n solver too: a third,
    independently written device solver (textbook loop nests over a workspace; both of its forms) gives the
    bits of the quad solver -- and the oracle's -- for every restart."""
    from optik_amd import _native as nat
    d, ch = chains[robot]
    hc = dev.HipChain(**d)
    rng = np.random.default_rng(19)
    tg, x0 = make_targets(oracle, d, ch, rng, 1)
    kw = dict(solution_mode="quality", tol_f=1e-8)
    cfg = nat.make_config(**kw)
    tgd, x0d = torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda")
    tuned = hc.ik_batch(cfg, tgd, x0d, 0, R)
    torch.cuda.synchronize()
    with nat.options(solve_kernel="general", wide_form="lds" if R <= 2048 else "hbm"):
        general = hc.ik_batch(cfg, tgd, x0d, 0, R)
        torch.cuda.synchronize()
    assert (hc.last_launch()["lds_bytes"] > 8192) == (R <= 2048)  # the LDS form / the HBM workspace
    for k in ("status", "evals", "win_idx"):
        assert torch.equal(tuned[k], general[k]), k
    for k in ("x", "f", "win_x", "win_f", "win_key"):
        assert torch.equal(tuned[k].view(torch.int64), general[k].view(torch.int64)), k
    ref = _oracle_all(oracle, ch, kw, tg[0], x0[0], 0, R)
    assert np.array_equal(general["status"].cpu().numpy(), ref["status"])
    assert_bit_equal(general["x"].cpu().numpy(), ref["xs"].T, "general solver vs oracle")
