"""The quad-distributed restart solver (optik_amd/csrc/ik_quad.hpp + ik_nnls_quad.hpp) on the CPU.

The device headers are compiled for the host with the wave emulated by one thread per lane
(tests/emu/: every cross-lane move and LDS hand-over is an exchange between barriers), and every
restart -- status, evaluation count, x, f -- must equal the C oracle's bit for bit: the same contract
the -m gpu parity tests check on the hardware, available without one.  What this covers that a GPU
run cannot localise: the ORDER of every distributed sum, the lane that owns each row / column, and
that no cross-lane primitive sits under lane-divergent control flow (the emulation aborts there).
The GPU-specific parts (DPP moves, occupancy, LDS banking) are the -m gpu tests' and the profiles'."""
import numpy as np
import pytest

from conftest import ROBOT_SPECS  # noqa: F401


@pytest.fixture(scope="module")
def emu():
    from emu import binding
    binding.build()
    return binding


def test_lane_moves_follow_the_dpp_controls(emu):
    """ik_lane.hpp's quad moves run in the emulation through the SAME __builtin_amdgcn_update_dpp controls the device
    executes, decoded as the hardware decodes a quad_perm (lane i reads lane (ctrl >> 2 i) & 3 of its quad): broadcast
    from lane k, the two butterflies, and the rotations the ChaCha diagonal rounds use."""
    import ctypes as C
    out = (C.c_int * 80)()
    emu.lib().quad_emu_lane_moves(out)
    got = np.array(list(out)).reshape(10, 8)
    lanes = np.arange(8)
    base = lanes & ~3
    for k in range(4):
        assert (got[k] == base + k).all()                       # quad_get(v, k): the value lane k of the quad holds
    assert (got[4] == (lanes ^ 1)).all() and (got[5] == (lanes ^ 2)).all()   # quad_xor
    for r in (1, 2, 3):
        assert (got[5 + r] == base + ((lanes + r) & 3)).all()   # quad_rot(v, r): the value of lane (own + r) & 3
    assert got[9].all()


def _case(oracle, chains, robot, seed):
    d, ch = chains[robot]
    rng = np.random.default_rng(seed)
    _, tgt = oracle.fk(ch, rng.uniform(d["lb"], d["ub"]))
    x0 = rng.uniform(d["lb"], d["ub"])
    return d, ch, tgt, x0


def _assert_same(got, ref, n):
    assert np.array_equal(got["status"], ref["status"]), np.argwhere(got["status"] != ref["status"])[:8]
    assert np.array_equal(got["evals"], ref["evals"])
    assert np.array_equal(got["f"].view(np.uint64), ref["fs"].view(np.uint64))
    assert np.array_equal(got["x"].view(np.uint64), np.ascontiguousarray(ref["xs"][:, :n]).view(np.uint64))


@pytest.mark.parametrize("robot,R,quads", [("panda", 24, 1), ("panda", 12, 2), ("ur10", 24, 2), ("ur3e", 16, 1),
                                           ("panda_hand", 12, 1), ("panda5", 16, 2), ("panda4", 16, 1),
                                           ("panda3", 16, 2), ("panda2", 16, 1), ("panda1", 8, 2), ("arm8", 10, 1)])
def test_every_restart_bit_equal_to_the_oracle(emu, oracle, chains, robot, R, quads):
    from optik_amd import _native as nat
    d, ch, tgt, x0 = _case(oracle, chains, robot, 5)
    n = len(d["lb"])
    got = emu.solve(d, nat.make_config(solution_mode="speed"), tgt, x0, 0, R, quads=quads)
    ref = oracle.ik(ch, oracle.make_config(solution_mode="speed"), tgt, x0, 0, R, n_threads=4, early_exit=False,
                    per_restart=True)
    _assert_same(got, ref, n)
    assert 0 < ref["success"].sum() < R or n < 3  # both outcomes are exercised on the real arms


@pytest.mark.parametrize("robot,R,quads", [("panda", 48, 4), ("ur10", 40, 4), ("panda_hand", 24, 2), ("ur3e", 24, 3),
                                           ("panda3", 24, 2), ("panda1", 8, 1)])
def test_lane_per_restart_form_bit_equal_to_the_oracle(emu, oracle, chains, robot, R, quads):
    """ik_lane64.hpp (one restart per lane, the wave's bounded sub-problems ranked by predicted pass count and solved
    in rounds by its quads) on a partial wave of 4 * quads lanes: more problems than quads, so several rounds per trip."""
    from optik_amd import _native as nat
    d, ch, tgt, x0 = _case(oracle, chains, robot, 7)
    n = len(d["lb"])
    got = emu.solve(d, nat.make_config(solution_mode="speed"), tgt, x0, 0, R, quads=quads, lane64=True)
    ref = oracle.ik(ch, oracle.make_config(solution_mode="speed"), tgt, x0, 0, R, n_threads=4, early_exit=False,
                    per_restart=True)
    _assert_same(got, ref, n)


def test_quality_key_weights_and_tight_tolerance(emu, oracle, chains):
    """SolutionMode::Quality's key ||x - x0|| (an ordered sum over the quad), the reference test's
    non-trivial weights (tests/test_gradient.rs:37-38) and tol_f = 1e-12 (tests/test_ik.rs:99)."""
    from optik_amd import _native as nat
    d, ch, tgt, x0 = _case(oracle, chains, "ur10", 9)
    kw = dict(solution_mode="quality", tol_f=1e-12, linear_weight=(0.7, 5.0, 0.25), angular_weight=(0.5, 1.0, 0.99))
    got = emu.solve(d, nat.make_config(**kw), tgt, x0, 0, 16, quads=2)
    ref = oracle.ik(ch, oracle.make_config(**kw), tgt, x0, 0, 16, n_threads=4, early_exit=False, per_restart=True)
    _assert_same(got, ref, len(d["lb"]))
    ok = ref["success"] != 0
    want = np.where(ok, np.sqrt(((ref["xs"][:, :6] - x0) ** 2).sum(axis=1)), np.inf)
    # (the key's bits are checked on the GPU against the selection; here: same winner, same distance to roundoff)
    assert np.array_equal(np.isinf(got["key"]), ~ok)
    np.testing.assert_allclose(got["key"][ok], want[ok], rtol=1e-14)


def test_ftol_and_xtol_stops(emu, oracle, chains):
    """tol_df / tol_dx >= 0 make FTOL / XTOL count (lib.rs:376-379); the x test is a per-joint test
    reduced over the quad."""
    from optik_amd import _native as nat
    d, ch, tgt, x0 = _case(oracle, chains, "panda", 13)
    kw = dict(solution_mode="speed", tol_f=1e-14, tol_df=1e-10, tol_dx=1e-7)
    got = emu.solve(d, nat.make_config(**kw), tgt, x0, 0, 12, quads=1)
    ref = oracle.ik(ch, oracle.make_config(**kw), tgt, x0, 0, 12, n_threads=4, early_exit=False, per_restart=True)
    _assert_same(got, ref, 7)
    assert set(np.unique(ref["status"])) & {3, 4}
