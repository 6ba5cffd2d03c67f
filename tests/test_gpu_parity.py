"""-m gpu: the HIP path (through the C ABI of include/optik_hip.h) against the CPU
oracle on the same seeded inputs.  The kernels execute the oracle's operation
sequence (no FMA contraction, shared elementary functions), so floating-point
outputs are compared BIT-FOR-BIT, not within a tolerance; the north-star tolerance
(1e-6 on joint angles, exact winner index) is therefore met with margin."""
import numpy as np
import pytest

from gpu_util import assert_bit_equal, make_targets

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    from optik_amd import device
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return device


@pytest.fixture(scope="module")
def hip_chains(dev, chains):
    return {name: dev.HipChain(**d) for name, (d, _) in chains.items()}


def test_elementary_functions_bit_exact(dev, oracle):
    """IEEE division / sqrt on gfx950 and the shared sincos / atan2 sequences."""
    import ctypes as C
    L = oracle.lib()
    L.ok_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ok_atan2_q1.argtypes = [C.c_double, C.c_double]
    L.ok_atan2_q1.restype = C.c_double
    rng = np.random.default_rng(0)
    n = 200000
    a = np.concatenate([rng.uniform(-10, 10, n // 2), rng.standard_normal(n // 2) * 10.0 ** rng.integers(-30, 30, n // 2)])
    b = np.concatenate([rng.uniform(-10, 10, n // 2), rng.standard_normal(n // 2) * 10.0 ** rng.integers(-30, 30, n // 2)])
    assert_bit_equal(dev.probe(0, a, b), a / b, "f64 division")
    assert_bit_equal(dev.probe(1, np.abs(a)), np.sqrt(np.abs(a)), "f64 sqrt")
    ang = np.concatenate([rng.uniform(-7, 7, 20000), rng.uniform(-1e-4, 1e-4, 2000), rng.uniform(-500, 500, 2000)])
    s_ref, c_ref = np.empty_like(ang), np.empty_like(ang)
    for i, x in enumerate(ang):
        s, c = C.c_double(), C.c_double()
        L.ok_sincos(float(x), C.byref(s), C.byref(c))
        s_ref[i], c_ref[i] = s.value, c.value
    assert_bit_equal(dev.probe(2, ang), s_ref, "sin")
    assert_bit_equal(dev.probe(3, ang), c_ref, "cos")
    y = rng.uniform(1e-3, 1, 20000)
    x = rng.uniform(0, 1, 20000)
    x[:500] = 0.0
    at_ref = np.array([L.ok_atan2_q1(float(yy), float(xx)) for yy, xx in zip(y, x)])
    assert_bit_equal(dev.probe(4, y, x), at_ref, "atan2")


@pytest.mark.parametrize("robot", ["ur3e", "panda", "panda_hand", "ur10"])
def test_restart_seeds_bit_exact(dev, oracle, chains, hip_chains, robot):
    """ChaCha8 stream-i seeds (lib.rs:358-370): integer path + f64 mapping."""
    _, ch = chains[robot]
    first, count = 1, 3000
    got = hip_chains[robot].seed_batch(first, count).cpu().numpy()
    want = np.array([oracle.restart_seed(ch, i) for i in range(first, first + count)]).T
    assert_bit_equal(got, want, "restart seeds")
    # far-away stream ids (4M restarts sharded over 8 GPUs, and > 2^32)
    for first in (4194304 - 5, 2**32 - 3, 2**40 + 7):
        got = hip_chains[robot].seed_batch(first, 8).cpu().numpy()
        want = np.array([oracle.restart_seed(ch, i) for i in range(first, first + 8)]).T
        assert_bit_equal(got, want, "restart seeds (large index)")


@pytest.mark.parametrize("robot", ["ur3e", "panda", "panda_hand", "ur10"])
@pytest.mark.parametrize("weights", ["default", "reference_test", "identity_quirk"])
def test_objective_and_gradient_bit_exact(dev, oracle, chains, hip_chains, robot, weights):
    from optik_amd import _native as nat
    d, ch = chains[robot]
    wl, wa = {"default": ((1, 1, 1), (1, 1, 1)),
              "reference_test": ((0.0, 5.0, 0.25), (0.005, 1.0, 0.99)),  # tests/test_gradient.rs:37-38
              "identity_quirk": ((1, 0, 0), (1, 1, 1))}[weights]
    rng = np.random.default_rng(7)
    B = 3000
    q = rng.uniform(d["lb"], d["ub"], size=(B, len(d["lb"])))
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
    fr, gr = np.empty(B), np.empty((len(d["lb"]), B))
    for i in range(B):
        fr[i], gr[:, i] = oracle.eval_fg(ch, tgt, q[i], wl, wa, ee_offset=ee_pose)
    assert_bit_equal(f, fr, "objective")
    assert_bit_equal(g, gr, "gradient")


@pytest.mark.parametrize("robot", ["ur3e", "panda_hand"])
def test_fk_and_jacobian_bit_exact(dev, oracle, chains, hip_chains, robot):
    d, ch = chains[robot]
    rng = np.random.default_rng(3)
    B = 1000
    q = rng.uniform(d["lb"], d["ub"], size=(B, len(d["lb"])))
    pose, jac = hip_chains[robot].fk_batch(torch.tensor(q.T.copy(), device="cuda"), jacobian=True)
    pose, jac = pose.cpu().numpy(), jac.cpu().numpy()
    n = len(d["lb"])
    pr, jr = np.empty((7, B)), np.empty((6 * n, B))
    for i in range(B):
        _, ee = oracle.fk(ch, q[i])
        pr[:, i] = ee
        jr[:, i] = oracle.joint_jacobian(ch, q[i]).T.ravel()  # column-major 6 x n
    assert_bit_equal(pose, pr, "fk pose")
    assert_bit_equal(jac, jr, "jacobian")


def _oracle_all(oracle, ch, cfg_kw, tgt, x0, begin, end):
    cfg = oracle.make_config(**cfg_kw)
    return oracle.ik(ch, cfg, tgt, x0, begin, end, n_threads=4, early_exit=False, per_restart=True)


def _run(hc, path, cfg, tgd, x0d, begin, end, flags=0):
    """The GPU paths: the single-launch solvers -- `kernel`: what a launch of this size gets (the quad solver below
    one full load of the chip), `lane64`: the lane-per-restart form (ik_lane64.hpp), the default from there on,
    forced here at the test's size -- and `quad`: the quad solver forced whatever the size."""
    from optik_amd import _native as nat
    if path == "lane64":
        with nat.options(solve_kernel="lane64"):
            out = hc.ik_batch(cfg, tgd, x0d, begin, end, flags=flags)
            torch.cuda.synchronize()
    elif path == "kernel":
        out = hc.ik_batch(cfg, tgd, x0d, begin, end, flags=flags)
    else:
        with nat.options(solve_kernel="quad"):
            out = hc.ik_batch(cfg, tgd, x0d, begin, end, flags=flags)
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("robot,tol_f,R", [("panda", 1e-6, 4096), ("ur10", 1e-12, 2048),
                                           ("ur3e", 1e-6, 2048), ("panda_hand", 1e-8, 2048)])
@pytest.mark.parametrize("mode", ["speed", "quality"])
@pytest.mark.parametrize("path", ["kernel", "lane64"])
def test_every_restart_bit_exact(dev, oracle, chains, hip_chains, robot, tol_f, R, mode, path):
    """One target, restarts 0..R-1: status, evaluation count, returned x and f of EVERY
    restart equal the oracle's, and so does the selected winner."""
    from optik_amd import _native as nat
    d, ch = chains[robot]
    rng = np.random.default_rng(11)
    tg, x0 = make_targets(oracle, d, ch, rng, 1)
    kw = dict(solution_mode=mode, tol_f=tol_f)
    cfg = nat.make_config(**kw)
    out = _run(hip_chains[robot], path, cfg, torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda"), 0, R)
    ref = _oracle_all(oracle, ch, kw, tg[0], x0[0], 0, R)
    status = out["status"].cpu().numpy()
    assert np.array_equal(status, ref["status"]), np.argwhere(status != ref["status"])[:10]
    assert np.array_equal(out["evals"].cpu().numpy(), ref["evals"])
    assert_bit_equal(out["f"].cpu().numpy(), ref["fs"], "per-restart f")
    assert_bit_equal(out["x"].cpu().numpy(), ref["xs"].T, "per-restart x")
    assert ref["found"], "test target should be solvable"
    assert int(out["win_idx"].cpu()[0]) == ref["winner"]
    assert_bit_equal(out["win_x"].cpu().numpy()[0], ref["x"], "winner x")
    assert_bit_equal(out["win_f"].cpu().numpy(), [ref["f"]], "winner f")
    assert 0.02 < ref["success"].mean() < 0.98  # both outcomes are exercised


@pytest.mark.parametrize("robot", ["panda1", "panda2", "panda3", "panda4", "panda5", "arm8"])
@pytest.mark.parametrize("path", ["kernel", "lane64"])
def test_other_joint_counts_bit_exact(dev, oracle, chains, hip_chains, robot, path):
    """Kernels are instantiated for 1 <= n <= 8: sub-chains of the Panda (no trailing fixed
    joint) and a synthetic 8-joint arm through both paths, every restart against the oracle
    (an 8-DoF chain always runs on the quad solver: the lane-per-restart form is built for n <= 7)."""
    from optik_amd import _native as nat
    d, ch = chains[robot]
    rng = np.random.default_rng(17)
    tg, x0 = make_targets(oracle, d, ch, rng, 2)
    kw = dict(solution_mode="quality", tol_f=1e-8)
    out = _run(hip_chains[robot], path, nat.make_config(**kw), torch.tensor(tg, device="cuda"),
               torch.tensor(x0, device="cuda"), 0, 384)
    st = out["status"].cpu().numpy().reshape(2, -1)
    xs = out["x"].cpu().numpy()
    for t in range(2):
        ref = _oracle_all(oracle, ch, kw, tg[t], x0[t], 0, 384)
        assert np.array_equal(st[t], ref["status"])
        assert_bit_equal(xs[:, t * 384:(t + 1) * 384], ref["xs"].T, f"x target {t}")
        assert int(out["win_idx"].cpu()[t]) == (ref["winner"] if ref["found"] else -1)


def test_restart_ranges_compose(dev, oracle, chains, hip_chains):
    """Sharding [0,R) into ranges (the multi-GPU partition) gives the same per-restart
    results; a ragged range (not a multiple of 64) is handled."""
    from optik_amd import _native as nat
    d, ch = chains["panda"]
    rng = np.random.default_rng(5)
    tg, x0 = make_targets(oracle, d, ch, rng, 1)
    cfg = nat.make_config(solution_mode="quality")
    tgd, x0d = torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda")
    full = hip_chains["panda"].ik_batch(cfg, tgd, x0d, 0, 1000)
    a = hip_chains["panda"].ik_batch(cfg, tgd, x0d, 0, 333)
    b = hip_chains["panda"].ik_batch(cfg, tgd, x0d, 333, 1000)
    torch.cuda.synchronize()
    for k in ("f", "status", "evals"):
        assert torch.equal(torch.cat([a[k], b[k]]), full[k])
    assert torch.equal(torch.cat([a["x"], b["x"]], dim=1), full["x"])
    keys = torch.stack([a["win_key"][0], b["win_key"][0]])
    idxs = torch.stack([a["win_idx"][0], b["win_idx"][0]])
    valid = idxs >= 0
    best = torch.argmin(torch.where(valid, keys, torch.full_like(keys, float("inf"))))
    assert int(idxs[best]) == int(full["win_idx"][0])


@pytest.mark.parametrize("path", ["kernel", "lane64"])
def test_many_targets_batch(dev, oracle, chains, hip_chains, path):
    """Config-5 shape: T targets x R restarts each; per-target winners match the oracle
    run target by target (Speed: lowest successful index)."""
    from optik_amd import _native as nat
    d, ch = chains["panda"]
    rng = np.random.default_rng(21)
    T, R = 24, 96
    tg, x0 = make_targets(oracle, d, ch, rng, T)
    kw = dict(solution_mode="speed", tol_f=1e-6)
    out = _run(hip_chains["panda"], path, nat.make_config(**kw), torch.tensor(tg, device="cuda"),
               torch.tensor(x0, device="cuda"), 0, R)
    win = out["win_idx"].cpu().numpy()
    wx = out["win_x"].cpu().numpy()
    st = out["status"].cpu().numpy().reshape(T, R)
    for t in range(T):
        ref = _oracle_all(oracle, ch, kw, tg[t], x0[t], 0, R)
        assert np.array_equal(st[t], ref["status"])
        if ref["found"]:
            assert win[t] == ref["winner"]
            assert_bit_equal(wx[t], ref["x"], f"winner x target {t}")
        else:
            assert win[t] == -1


def test_early_exit_keeps_the_winner(dev, oracle, chains, hip_chains):
    """Speed + early exit (lib.rs:382-384): restarts above a known success are
    abandoned, the winner (lowest successful index) does not change."""
    from optik_amd import _native as nat
    d, ch = chains["panda"]
    rng = np.random.default_rng(2)
    tg, x0 = make_targets(oracle, d, ch, rng, 4)
    cfg = nat.make_config(solution_mode="speed")
    tgd, x0d = torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda")
    full = hip_chains["panda"].ik_batch(cfg, tgd, x0d, 0, 2048)
    fast = hip_chains["panda"].ik_batch(cfg, tgd, x0d, 0, 2048, flags=nat.IK_EARLY_EXIT)
    torch.cuda.synchronize()
    assert torch.equal(full["win_idx"], fast["win_idx"])
    assert torch.equal(full["win_x"], fast["win_x"])
    assert (fast["status"] == nat.RES_FORCED_STOP).any()
    st_full, st_fast = full["status"].view(4, -1), fast["status"].view(4, -1)
    for t in range(4):
        w = int(full["win_idx"][t])
        assert w >= 0
        assert torch.equal(st_full[t, : w + 1], st_fast[t, : w + 1])  # nothing below the winner is abandoned


@pytest.mark.parametrize("flags", ["early", "early+restart_major", "early+find_any"])
def test_early_exit_on_the_lane_per_restart_form(dev, oracle, chains, hip_chains, flags):
    """The same rule on ik_lane64.hpp (forced at this size): every lane checks first_success before its evaluation;
    the deterministic readings return the full run's winners, find_any a solved restart of every solvable target."""
    from optik_amd import _native as nat
    d, ch = chains["panda"]
    rng = np.random.default_rng(4)
    T, R = 12, 1024
    tg, x0 = make_targets(oracle, d, ch, rng, T)
    cfg = nat.make_config(solution_mode="speed")
    tgd, x0d = torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda")
    full = _run(hip_chains["panda"], "lane64", cfg, tgd, x0d, 0, R)
    fl = nat.IK_EARLY_EXIT | (nat.IK_RESTART_MAJOR if "restart_major" in flags else 0) | (nat.IK_FIND_ANY if "find_any" in flags else 0)
    fast = _run(hip_chains["panda"], "lane64", cfg, tgd, x0d, 0, R, flags=fl)
    assert (fast["status"] == nat.RES_FORCED_STOP).any()
    if "find_any" in flags:
        ok = (full["status"] == nat.RES_STOPVAL).view(T, R)
        w = fast["win_idx"]
        assert torch.equal(w >= 0, ok.any(1))
        for t in range(T):
            if int(w[t]) >= 0:
                assert bool(ok[t, int(w[t])])
                assert torch.equal(fast["win_x"][t], full["x"][:, t * R + int(w[t])])
    else:
        assert torch.equal(full["win_idx"], fast["win_idx"])
        assert torch.equal(full["win_x"], fast["win_x"])


def test_launch_words_are_left_clean_for_the_next_launch(dev, oracle, chains, hip_chains):
    """The selection kernel of a launch puts the work-item counter and the first-success words back, and the next launch
    skips its fill commands when it finds them so (ik_kernels.hip: queue_clean / fs_clean).  A sequence of launches with
    and without early exit, with growing and shrinking target counts, small (one selection kernel) and large (two),
    must give the winners of the same launches made on a fresh chain each."""
    from optik_amd import _native as nat
    from optik_amd import device
    d, ch = chains["panda"]
    rng = np.random.default_rng(12)
    cfg = nat.make_config(solution_mode="speed")
    hc = device.HipChain(**d)
    seq = [(3, 96, nat.IK_EARLY_EXIT), (9, 64, nat.IK_EARLY_EXIT), (2, 5000, nat.IK_EARLY_EXIT), (16, 40, 0),
           (12, 128, nat.IK_EARLY_EXIT | nat.IK_RESTART_MAJOR), (1, 300, nat.IK_EARLY_EXIT | nat.IK_FIND_ANY),
           (5, 8192, 0), (7, 100, nat.IK_EARLY_EXIT)]
    for T, R, fl in seq:
        tg, x0 = make_targets(oracle, d, ch, rng, T)
        tgd, x0d = torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda")
        out = hc.ik_batch(cfg, tgd, x0d, 0, R, flags=fl)
        fresh = device.HipChain(**d).ik_batch(cfg, tgd, x0d, 0, R)  # no early exit: every restart runs
        torch.cuda.synchronize()
        if fl & nat.IK_FIND_ANY:
            ok = (fresh["status"] == nat.RES_STOPVAL).view(T, R)
            for t in range(T):
                w = int(out["win_idx"][t])
                assert (w >= 0) == bool(ok[t].any()) and (w < 0 or bool(ok[t, w]))
        else:
            assert torch.equal(out["win_idx"], fresh["win_idx"]), (T, R, fl)
            assert torch.equal(out["win_x"], fresh["win_x"])
        # every restart was handed out exactly once: none is left unwritten (status 0 is not a result code)
        assert int((out["status"] == 0).sum()) == 0


def test_ftol_and_xtol_count_as_success_when_enabled(dev, oracle, chains, hip_chains):
    """tol_df >= 0 / tol_dx >= 0 (lib.rs:376-379): FTOL / XTOL exits become successes and
    return NLopt's best-so-far point."""
    from optik_amd import _native as nat
    d, ch = chains["ur3e"]
    rng = np.random.default_rng(9)
    tg, x0 = make_targets(oracle, d, ch, rng, 1)
    kw = dict(solution_mode="quality", tol_f=1e-14, tol_df=1e-12, tol_dx=1e-4)
    out = hip_chains["ur3e"].ik_batch(nat.make_config(**kw), torch.tensor(tg, device="cuda"),
                                      torch.tensor(x0, device="cuda"), 0, 1024)
    torch.cuda.synchronize()
    ref = _oracle_all(oracle, ch, kw, tg[0], x0[0], 0, 1024)
    st = out["status"].cpu().numpy()
    assert np.array_equal(st, ref["status"])
    assert (st == nat.RES_FTOL).any() and (st == nat.RES_XTOL).any()
    assert_bit_equal(out["x"].cpu().numpy(), ref["xs"].T, "x")
    assert int(out["win_idx"].cpu()[0]) == ref["winner"]


@pytest.mark.parametrize("path", ["kernel", "lane64"])
@pytest.mark.parametrize("tol_dx", [-1.0, 0.0])
def test_zero_step_counts_as_xtol(dev, oracle, chains, hip_chains, path, tol_dx):
    """ftol_abs = 0 (tol_f = tol_df = 0): neither stopval nor ftol can fire, and restarts end
    through nlopt_stop_x's zero-step rule (NLopt >= 2.6.2: ||x - oldx|| <= xtol_rel ||x|| with
    xtol_rel = 0) -- XTOL_REACHED, a success only when tol_dx >= 0 (lib.rs:376-379)."""
    from optik_amd import _native as nat
    d, ch = chains["panda"]
    rng = np.random.default_rng(3)
    tg, x0 = make_targets(oracle, d, ch, rng, 1)
    kw = dict(solution_mode="speed", tol_f=0.0, tol_df=0.0, tol_dx=tol_dx)
    R = 256
    out = _run(hip_chains["panda"], path, nat.make_config(**kw), torch.tensor(tg, device="cuda"),
               torch.tensor(x0, device="cuda"), 0, R)
    ref = _oracle_all(oracle, ch, kw, tg[0], x0[0], 0, R)
    status = out["status"].cpu().numpy()
    assert np.array_equal(status, ref["status"])
    assert (status == nat.RES_XTOL).mean() > 0.9
    assert np.array_equal(out["evals"].cpu().numpy(), ref["evals"])
    assert_bit_equal(out["x"].cpu().numpy(), ref["xs"].T, "per-restart x")
    assert int(out["win_idx"].cpu()[0]) == (ref["winner"] if ref["found"] else -1)
    assert ref["found"] == (tol_dx >= 0.0)


@pytest.mark.parametrize("path", ["kernel", "lane64"])
def test_single_launch_deadline_abandons_and_keeps_what_was_found(dev, oracle, chains, hip_chains, path):
    """optik_hip_ik_batch(deadline_s): every lane / quad checks the device clock before its evaluation -- restarts in
    flight at the deadline end FORCED_STOP with their best point so far, the rest of the queue is abandoned unstarted;
    what finished before is the oracle's result.  Both single-launch solvers (2^20 restarts: the lane-per-restart form
    by default; the quad solver forced)."""
    import time
    from optik_amd import _native as nat
    d, ch = chains["panda"]
    rng = np.random.default_rng(31)
    tg, x0 = make_targets(oracle, d, ch, rng, 1)
    cfg = nat.make_config(solution_mode="speed")
    hc = hip_chains["panda"]
    R = 1 << 20
    tgd, x0d = torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda")
    with nat.options(solve_kernel="quad" if path == "kernel" else "lane64"):
        bufs = hc.alloc_ik_buffers(1, R)
        hc.ik_batch(cfg, tgd, x0d, 0, 1 << 16, bufs=hc.alloc_ik_buffers(1, 1 << 16))   # (first-launch costs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = hc.ik_batch(cfg, tgd, x0d, 0, R, deadline_s=0.010, bufs=bufs)
        torch.cuda.synchronize()
        took = time.perf_counter() - t0
    status = out["status"].cpu().numpy()
    evals = out["evals"].cpu().numpy()
    forced = status == nat.RES_FORCED_STOP
    assert forced.any() and (~forced).any(), "the deadline should fall inside the launch"
    assert took < 0.025, took                        # (the whole launch takes 60 - 70 ms)
    assert (forced & (evals == 0)).sum() > R // 4    # never started
    xs = out["x"].cpu().numpy()
    for i in np.flatnonzero(~forced)[:200:10]:
        r = oracle.solve_restart(ch, oracle.make_config("speed"), tg[0], x0[0], int(i))
        assert r.result == status[i] and r.n_evals == evals[i]
        assert_bit_equal(xs[:, i], np.array(r.x[:7]), f"restart {i}")
    ok = np.flatnonzero(status == nat.RES_STOPVAL)
    assert int(out["win_idx"].cpu()[0]) == (ok.min() if len(ok) else -1)


def test_prismatic_chain_has_forward_kinematics_only(dev, oracle, chains, hip_chains):
    """kinematics.rs:243-255 handles prismatic joints in FK; the Jacobian is todo!() (:185), so
    ik() on such a chain panics in the reference -- here FK matches the oracle bit for bit and
    the Jacobian / objective / ik entry points refuse the chain."""
    from optik_amd import _native as nat
    d, ch = chains["gantry"]
    assert list(d["types"]) == [2, 1, 2, 1, 0]
    hc = hip_chains["gantry"]
    rng = np.random.default_rng(31)
    B = 500
    q = rng.uniform(d["lb"], d["ub"], size=(B, 4))
    ee_off = np.array([0.01, -0.02, 0.03, 0.0, 0.0, np.sin(0.2), np.cos(0.2)])
    qd = torch.tensor(q.T.copy(), device="cuda")
    for off, pose_off in ((None, None), (ee_off, oracle.Pose.make(ee_off[:3], ee_off[3:]))):
        got = hc.fk_batch(qd, ee_offset7=off).cpu().numpy()
        want = np.array([oracle.fk(ch, q[i], ee_offset=pose_off)[1] for i in range(B)]).T
        assert_bit_equal(got, want, "prismatic fk")
    with pytest.raises(nat.OptikHipError, match="prismatic"):
        hc.fk_batch(qd, jacobian=True)
    with pytest.raises(nat.OptikHipError, match="prismatic"):
        hc.eval_batch(qd, np.array([0, 0, 0, 0, 0, 0, 1.0]))
    tg = torch.zeros((1, 7), dtype=torch.float64, device="cuda")
    tg[0, 6] = 1.0
    with pytest.raises(nat.OptikHipError, match="prismatic"):
        hc.ik_batch(nat.make_config(), tg, torch.zeros((1, 4), dtype=torch.float64, device="cuda"), 0, 4)
