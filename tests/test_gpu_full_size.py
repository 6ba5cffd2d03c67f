"""-m gpu: BASELINE.json's configurations at their full sizes.

Every restart of config 2 (65 536 Panda restarts, both solvers), of config 3 (2^20 UR10 restarts at
tol_f = 1e-12, one launch of the default solver), of config 4 (2^22 Panda restarts as 8 shards of
524 288, Quality, each shard one launch of the default solver) and every target of one GPU's share of
config 5 (512 targets x 256 restarts) is compared with the CPU oracle bit for bit -- status, evaluation
count, x, f, winner (the oracle runs 20-30 k restarts/s per host thread: a second or two for config 2,
~4 s for config 3 and ~15 s for config 4 on the box's 16 cores).  The properties below hold at any size:

  * round trip: every restart reported as solved satisfies FK(x) == target to the pose
    tolerance implied by tol_f, and respects the joint limits;
  * the two single-launch solvers (lane-per-restart form / quad solver) agree bit for bit --
    a checksum over every per-restart output;
  * selection: the reported winner is the lowest solved index (Speed) / the solved restart
    closest to the seed (Quality), recomputed from the per-restart outputs;
  * sharding a restart range (the multi-GPU partition) does not change any result.
"""
import os

import numpy as np
import pytest

from conftest import ROBOTS
from gpu_util import assert_bit_equal

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def robots():
    from optik_amd import Robot
    return {"panda": Robot.from_urdf_file(os.path.join(ROBOTS, "panda.urdf"), "panda_link0", "panda_link8"),
            "ur10": Robot.from_urdf_file(os.path.join(ROBOTS, "ur10.urdf"), "base_link", "ee_link")}


def _targets(robot, hc, T, seed):
    rng = np.random.default_rng(seed)
    lb, ub = (np.array(v) for v in robot.joint_limits())
    q = rng.uniform(lb, ub, size=(T, len(lb)))
    tg = hc.fk_batch(torch.tensor(q.T.copy(), device="cuda")).T.contiguous()
    x0 = torch.tensor(rng.uniform(lb, ub, size=(T, len(lb))), device="cuda")
    return tg, x0, lb, ub


def _checksum(out):
    """Order-sensitive 64-bit mix of every per-restart output (bit patterns)."""
    parts = [out["x"].view(torch.int64).flatten(), out["f"].view(torch.int64),
             out["status"].to(torch.int64), out["evals"].to(torch.int64)]
    h = torch.zeros((), dtype=torch.int64, device=out["f"].device)
    for p in parts:
        idx = torch.arange(p.numel(), device=p.device, dtype=torch.int64)
        h = h * 1000003 + ((p ^ (idx * 0x9E3779B97F4A7C15 % (2 ** 62))).sum())
    return int(h.item())


def _pose_error(hc, x, target7):
    """max |FK(x) - target| over translation and (sign-aligned) quaternion, per column."""
    pose = hc.fk_batch(x.contiguous())
    t = target7.view(7, 1)
    dt = (pose[:3] - t[:3]).abs().amax(0)
    dq = torch.minimum((pose[3:] - t[3:]).abs().amax(0), (pose[3:] + t[3:]).abs().amax(0))
    return torch.maximum(dt, dq)


def _threads():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def _assert_every_restart_equals_oracle(out, ref, what):
    status = out["status"].cpu().numpy()
    assert np.array_equal(status, ref["status"]), (what, np.argwhere(status != ref["status"])[:10])
    assert np.array_equal(out["evals"].cpu().numpy(), ref["evals"]), what
    assert_bit_equal(out["f"].cpu().numpy(), ref["fs"], what + ": per-restart f")
    assert_bit_equal(out["x"].cpu().numpy(), ref["xs"].T, what + ": per-restart x")


@pytest.mark.parametrize("path", ["kernel", "quad"])
def test_config2_every_restart_equals_oracle(robots, oracle, chains, path):
    """Config 2 at its own size: all 65 536 Panda restarts of the headline workload -- status,
    evaluation count, x, f -- and the Speed winner against the oracle, on both solvers: the single launch
    a batch of this size gets (the lane-per-restart form) and the quad solver forced onto it
    (lib.rs:297-413; the reference's own check of this shape is tests/test_ik.rs:91-130)."""
    from optik_amd import _native as nat
    robot = robots["panda"]
    hc = robot.hip_chain("cuda:0")
    tg, x0, lb, ub = _targets(robot, hc, 1, 0)
    R = 65536
    cfg = nat.make_config("speed", tol_f=1e-6)
    if path == "kernel":
        out = hc.ik_batch(cfg, tg, x0, 0, R)
        assert hc.last_launch()["lds_bytes"] > 30000, "a launch of this size runs on the lane-per-restart form"
    else:
        with nat.options(solve_kernel="quad"):
            out = hc.ik_batch(cfg, tg, x0, 0, R)
        assert hc.last_launch()["lds_bytes"] < 30000
    torch.cuda.synchronize()
    _, ch = chains["panda"]
    ref = oracle.ik(ch, oracle.make_config(solution_mode="speed", tol_f=1e-6), tg[0].cpu().numpy(), x0[0].cpu().numpy(),
                    0, R, n_threads=_threads(), early_exit=False, per_restart=True)
    _assert_every_restart_equals_oracle(out, ref, "config 2, " + path)
    assert ref["found"] and int(out["win_idx"][0]) == ref["winner"]
    assert_bit_equal(out["win_x"].cpu().numpy()[0], ref["x"], "winner x")
    assert_bit_equal(out["win_f"].cpu().numpy(), [ref["f"]], "winner f")


def test_config3_every_restart_equals_oracle(robots, oracle, chains):
    """Config 3 at its own size on the DEFAULT solver: 2^20 UR10 restarts at tol_f = 1e-12, Quality, as ONE launch
    of optik_hip_ik_batch (the lane-per-restart form at this size) -- every restart and the winner (the solved
    restart closest to the seed) against the oracle.  ~4 s of oracle time on 16 threads."""
    from optik_amd import _native as nat
    robot = robots["ur10"]
    hc = robot.hip_chain("cuda:0")
    tg, x0, lb, ub = _targets(robot, hc, 1, 3)
    R = 1 << 20
    out = hc.ik_batch(nat.make_config("quality", tol_f=1e-12), tg, x0, 0, R)
    torch.cuda.synchronize()
    assert hc.last_launch()["lds_bytes"] > 30000, "a launch of this size runs on the lane-per-restart form"
    _, ch = chains["ur10"]
    ref = oracle.ik(ch, oracle.make_config(solution_mode="quality", tol_f=1e-12), tg[0].cpu().numpy(),
                    x0[0].cpu().numpy(), 0, R, n_threads=_threads(), early_exit=False, per_restart=True)
    _assert_every_restart_equals_oracle(out, ref, "config 3")
    assert ref["found"] and int(out["win_idx"][0]) == ref["winner"]
    assert_bit_equal(out["win_x"].cpu().numpy()[0], ref["x"], "winner x")
    # ... and an eighth of it (one GPU's contiguous index range, section 8e) on the quad solver
    g = 5
    with nat.options(solve_kernel="quad"):
        part = hc.ik_batch(nat.make_config("quality", tol_f=1e-12), tg, x0, g * (R // 8), (g + 1) * (R // 8))
        torch.cuda.synchronize()
    sl = slice(g * (R // 8), (g + 1) * (R // 8))
    sub = {k: ref[k][sl] for k in ("status", "evals", "fs", "xs")}
    _assert_every_restart_equals_oracle(part, sub, "config 3, shard 5 on the quad solver")


def _quality_keys(xs, x0):
    """||x - x0||_2 as the kernels and the oracle form it (lib.rs:402-407): squares summed in joint order."""
    acc = np.zeros(xs.shape[0])
    for i in range(xs.shape[1]):
        d = xs[:, i] - x0[i]
        acc = acc + d * d
    return np.sqrt(acc)


def test_config4_every_restart_equals_oracle(robots, oracle, chains):
    """Config 4 at its own size on the DEFAULT solver: Panda, 2^22 restarts as 8 contiguous shards of 524 288 (one
    per GPU), SolutionMode::Quality.  Every shard is one launch of optik_hip_ik_batch (the lane-per-restart form);
    every restart of every shard -- status, evaluation count, x, f -- and every shard's (key, index) record equal
    the oracle's on that index range; the winner the two min-all-reduces of optik_amd.parallel pick from the eight
    records is the oracle's winner over the 4 M restarts (lib.rs:397-413; the reference's own check of the Quality
    rule is a property, tests/test_ik.rs:132-182).  ~15 s of oracle time on 16 threads."""
    from optik_amd import _native as nat
    from optik_amd.parallel import I64_MAX, local_key
    robot = robots["panda"]
    hc = robot.hip_chain("cuda:0")
    tg, x0, lb, ub = _targets(robot, hc, 1, 4)
    cfg = nat.make_config("quality", tol_f=1e-6)
    ocfg = oracle.make_config(solution_mode="quality", tol_f=1e-6)
    _, ch = chains["panda"]
    tgn, x0n = tg[0].cpu().numpy(), x0[0].cpu().numpy()
    R, G = 1 << 22, 8
    S = R // G
    keys, idxs, recs = [], [], []
    best = (np.inf, -1, None)  # the oracle's winner over all shards: (key, index, x)
    bufs = hc.alloc_ik_buffers(1, S)
    for g in range(G):
        sh = hc.ik_batch(cfg, tg, x0, g * S, (g + 1) * S, bufs=dict(bufs))
        torch.cuda.synchronize()
        assert hc.last_launch()["lds_bytes"] > 30000, "a shard of this size runs on the lane-per-restart form"
        ref = oracle.ik(ch, ocfg, tgn, x0n, g * S, (g + 1) * S, n_threads=_threads(), early_exit=False, per_restart=True)
        _assert_every_restart_equals_oracle(sh, ref, f"config 4, shard {g}")
        assert ref["found"] and int(sh["win_idx"][0]) == ref["winner"], g
        assert_bit_equal(sh["win_x"].cpu().numpy()[0], ref["x"], f"shard {g} winner x")
        rk = _quality_keys(ref["xs"], x0n)
        ok = ref["success"] != 0
        kmin = rk[ok].min()
        assert ref["winner"] == g * S + int(np.nonzero(ok & (rk == kmin))[0][0])  # (ties to the lower index)
        assert_bit_equal(sh["win_key"].cpu().numpy(), [kmin], f"shard {g} key")
        if kmin < best[0]:
            best = (kmin, ref["winner"], ref["x"].copy())
        k, i = local_key(sh, "quality")
        keys.append(k.clone()); idxs.append(i.clone())
        recs.append({kk: sh[kk].clone() for kk in ("win_x", "win_key", "win_idx")})
    key = torch.stack(keys).min(0).values                      # all-reduce MIN of the keys
    cand = [torch.where(k == key, i, torch.full_like(i, I64_MAX)) for k, i in zip(keys, idxs)]
    win = torch.stack(cand).min(0).values                      # all-reduce MIN of the indices
    assert int(win[0]) == best[1] >= 0
    owner = int(win[0]) // S
    assert_bit_equal(recs[owner]["win_x"].cpu().numpy()[0], best[2], "config 4 winner x")
    # the winner is a solution: FK(x) == target to the tolerance tol_f implies, inside the limits
    x = recs[owner]["win_x"][0]
    assert _pose_error(hc, x.view(-1, 1), tg[0]).max().item() < 2e-3
    assert (x >= torch.tensor(lb, device="cuda")).all() and (x <= torch.tensor(ub, device="cuda")).all()


@pytest.mark.parametrize("path", ["kernel", "rounds"])
def test_config5_share_winners_equal_oracle(robots, oracle, chains, path):
    """Config 5, one GPU's share (targets 512 .. 1023 of the 4096, 256 restarts each, Speed): the winner
    index, x and f of EVERY target against the oracle's 1-thread run of that target (lowest solved
    index: lib.rs:397-413 in the reference's deterministic reading)."""
    from concurrent.futures import ThreadPoolExecutor
    from optik_amd import _native as nat
    robot = robots["panda"]
    hc = robot.hip_chain("cuda:0")
    T, R = 4096, 256
    tg, x0, lb, ub = _targets(robot, hc, T, 5)
    tg, x0 = tg[512:1024].contiguous(), x0[512:1024].contiguous()
    cfg = nat.make_config("speed", tol_f=1e-6)
    if path == "kernel":
        out = hc.ik_batch(cfg, tg, x0, 0, R, flags=nat.IK_EARLY_EXIT | nat.IK_RESTART_MAJOR, per_restart=False)
    else:
        # the product's round scheduling (robot_host.cpp:ik_batch_on_device): 128 indices restart-major, then the
        # unsolved targets' next 128 -- the deterministic rule (no FIND_ANY)
        out = hc.ik_batch(cfg, tg, x0, 0, 128, flags=nat.IK_EARLY_EXIT | nat.IK_RESTART_MAJOR, per_restart=False)
        torch.cuda.synchronize()
        out = {k: v.clone() for k, v in out.items()}
        left = torch.nonzero(out["win_idx"] < 0).flatten()
        if left.numel():
            more = hc.ik_batch(cfg, tg[left].contiguous(), x0[left].contiguous(), 128, R,
                               flags=nat.IK_EARLY_EXIT | nat.IK_RESTART_MAJOR, per_restart=False)
            torch.cuda.synchronize()
            for k in ("win_idx", "win_x", "win_f"):
                out[k][left] = more[k]
    torch.cuda.synchronize()
    _, ch = chains["panda"]
    ocfg = oracle.make_config(solution_mode="speed", tol_f=1e-6)
    tgn, x0n = tg.cpu().numpy(), x0.cpu().numpy()

    def one(t):
        return oracle.ik(ch, ocfg, tgn[t], x0n[t], 0, R, n_threads=1, early_exit=True)

    with ThreadPoolExecutor(_threads()) as ex:
        refs = list(ex.map(one, range(len(tgn))))
    win = out["win_idx"].cpu().numpy()
    wx, wf = out["win_x"].cpu().numpy(), out["win_f"].cpu().numpy()
    want = np.array([r["winner"] if r["found"] else -1 for r in refs])
    assert np.array_equal(win, want), np.argwhere(win != want)[:10]
    found = want >= 0
    assert found.mean() > 0.99
    assert_bit_equal(wx[found], np.array([r["x"] for r in refs])[found], "config 5 winners' x")
    assert_bit_equal(wf[found], np.array([r["f"] for r in refs])[found], "config 5 winners' f")


def test_config2_panda_65536_speed(robots):
    """Panda 7-DoF, 65 536 restarts, SolutionMode::Speed (the bench configuration)."""
    from optik_amd import _native as nat
    robot = robots["panda"]
    hc = robot.hip_chain("cuda:0")
    tg, x0, lb, ub = _targets(robot, hc, 1, 0)
    cfg = nat.make_config("speed", tol_f=1e-6)
    R = 65536
    a = hc.ik_batch(cfg, tg, x0, 0, R)
    with nat.options(solve_kernel="quad"):
        b = hc.ik_batch(cfg, tg, x0, 0, R)
        torch.cuda.synchronize()
    assert _checksum(a) == _checksum(b)
    assert torch.equal(a["win_idx"], b["win_idx"]) and torch.equal(a["win_x"], b["win_x"])
    ok = a["status"] == nat.RES_STOPVAL
    assert 0.05 < ok.double().mean().item() < 0.95
    assert set(a["status"].unique().tolist()) <= {nat.RES_STOPVAL, nat.RES_FTOL, nat.RES_ROUNDOFF, nat.RES_FAILURE}
    # round trip on every solved restart: f < 1e-6 is a squared log error -> pose error < ~1e-3
    err = _pose_error(hc, a["x"][:, ok], tg[0])
    assert err.max().item() < 2e-3
    assert (a["f"][ok] < 1e-6).all() and (a["f"][~ok] >= 1e-6).all()
    x = a["x"]
    assert (x >= torch.tensor(lb, device="cuda")[:, None]).all() and (x <= torch.tensor(ub, device="cuda")[:, None]).all()
    # Speed: the winner is the lowest solved index (the reference's 1-thread order)
    assert int(a["win_idx"][0]) == int(torch.nonzero(ok)[0])


def test_config3_ur10_one_million_tight_tolerance(robots):
    """UR10 6-DoF, 2^20 restarts, tol_f = 1e-12 (tests/test_ik.rs:99): isolated solutions."""
    from optik_amd import _native as nat
    robot = robots["ur10"]
    hc = robot.hip_chain("cuda:0")
    tg, x0, lb, ub = _targets(robot, hc, 1, 3)
    cfg = nat.make_config("quality", tol_f=1e-12)
    R = 1 << 20
    out = hc.ik_batch(cfg, tg, x0, 0, R)
    torch.cuda.synchronize()
    ok = out["status"] == nat.RES_STOPVAL
    assert ok.sum().item() > 1000
    err = _pose_error(hc, out["x"][:, ok], tg[0])
    assert err.max().item() < 1e-6          # FK(ik(T)) == T to 1e-6, as the reference test asserts
    # a non-redundant arm has finitely many solutions (8 branches x 2*pi wraps inside the +-2*pi
    # limits): the solved restarts cluster on a few hundred points
    xs = out["x"][:, ok].T
    uniq = torch.unique(torch.round(xs * 1e4), dim=0)
    assert uniq.shape[0] * 100 < xs.shape[0]
    # Quality: winner = solved restart closest to the seed, ties to the lower index
    d = (xs - x0[0]).norm(dim=1)
    idx = torch.nonzero(ok).flatten()
    best = d.min()
    assert int(out["win_idx"][0]) == int(idx[d == best].min())
    # sharding (8 ranks x 131072) gives the same per-restart results
    h = _checksum(out)
    parts = [hc.ik_batch(cfg, tg, x0, g * (R // 8), (g + 1) * (R // 8)) for g in (0, 5)]
    torch.cuda.synchronize()
    for g, p in zip((0, 5), parts):
        sl = slice(g * (R // 8), (g + 1) * (R // 8))
        assert torch.equal(p["status"], out["status"][sl]) and torch.equal(p["x"], out["x"][:, sl])
    assert h == _checksum(out)


def test_config4_panda_four_million_sharded_quality(robots):
    """Panda, 2^22 restarts as 8 shards of 524 288 (one per GPU), SolutionMode::Quality: the
    winner the two min-all-reduces of optik_amd.parallel would pick from the shards' winner
    records is the winner of the unsharded run, and every shard's records are the matching
    slice of it (here the eight ranges run one after the other on the one GPU of the box)."""
    from optik_amd import _native as nat
    from optik_amd.parallel import I64_MAX, local_key
    robot = robots["panda"]
    hc = robot.hip_chain("cuda:0")
    tg, x0, lb, ub = _targets(robot, hc, 1, 4)
    cfg = nat.make_config("quality", tol_f=1e-6)
    R, G = 1 << 22, 8
    whole = hc.ik_batch(cfg, tg, x0, 0, R, per_restart=False)
    torch.cuda.synchronize()
    whole = {k: v.clone() for k, v in whole.items()}
    shards = []
    for g in range(G):
        sh = hc.ik_batch(cfg, tg, x0, g * (R // G), (g + 1) * (R // G), per_restart=False)
        torch.cuda.synchronize()
        shards.append({k: v.clone() for k, v in sh.items()})
    keys, idxs = zip(*(local_key(sh, "quality") for sh in shards))
    key = torch.stack(keys).min(0).values                      # all-reduce MIN of the keys
    cand = [torch.where(k == key, i, torch.full_like(i, I64_MAX)) for k, i in zip(keys, idxs)]
    win = torch.stack(cand).min(0).values                      # all-reduce MIN of the indices
    assert int(win[0]) == int(whole["win_idx"][0]) >= 0
    owner = int(win[0]) // (R // G)
    assert torch.equal(shards[owner]["win_x"], whole["win_x"]) and torch.equal(shards[owner]["win_key"], whole["win_key"])
    # the winner is a solution: FK(x) == target to the tolerance tol_f implies, inside the limits
    x = whole["win_x"][0]
    assert _pose_error(hc, x.view(-1, 1), tg[0]).max().item() < 2e-3
    assert (x >= torch.tensor(lb, device="cuda")).all() and (x <= torch.tensor(ub, device="cuda")).all()
    assert abs(float(whole["win_key"][0]) - float((x - x0[0]).norm())) < 1e-9


def test_config5_motion_planning_batch(robots):
    """4096 independent targets x 256 restarts each (512 targets is one GPU's share), Speed."""
    from optik_amd import _native as nat
    robot = robots["panda"]
    hc = robot.hip_chain("cuda:0")
    T, R = 4096, 256
    tg, x0, lb, ub = _targets(robot, hc, T, 5)
    cfg = nat.make_config("speed", tol_f=1e-6)
    out = hc.ik_batch(cfg, tg, x0, 0, R)
    torch.cuda.synchronize()
    st = out["status"].view(T, R)
    ok = st == nat.RES_STOPVAL
    solved = ok.any(dim=1)
    assert solved.double().mean().item() > 0.99        # 256 restarts solve (nearly) every reachable target
    first = torch.where(solved, ok.double().argmax(dim=1), torch.full((T,), -1, device="cuda"))
    assert torch.equal(first, out["win_idx"])
    # round trip of every winner against its own target
    wx = out["win_x"][solved].T.contiguous()
    pose = hc.fk_batch(wx)
    t = tg[solved].T
    dt = (pose[:3] - t[:3]).abs().amax(0)
    dq = torch.minimum((pose[3:] - t[3:]).abs().amax(0), (pose[3:] + t[3:]).abs().amax(0))
    assert torch.maximum(dt, dq).max().item() < 2e-3
    # one GPU's share (targets 512..1023) on the quad solver: identical winners
    with nat.options(solve_kernel="quad"):
        sub = hc.ik_batch(cfg, tg[512:1024].contiguous(), x0[512:1024].contiguous(), 0, R)
        torch.cuda.synchronize()
    assert torch.equal(sub["win_idx"], out["win_idx"][512:1024])
    assert torch.equal(sub["win_x"], out["win_x"][512:1024])
    # the random seeds are the same for every target (lib.rs:360): restart i of any two targets
    # starts from the same configuration, only restart 0 (the caller's seed) differs
    assert torch.equal(hc.seed_batch(1, 255), hc.seed_batch(1, 255))
