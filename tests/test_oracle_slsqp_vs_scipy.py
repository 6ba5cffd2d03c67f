"""Cross-checks the oracle's restatement of NLopt's SLSQP (un-vendored dependency:
nlopt 0.8.1 via kylc/rust-nlopt@8e731e3; call sites lib.rs:302-356, 372) against
scipy's Fortran build of the same Kraft code, driven in reverse-communication mode
with acc = 0, on the oracle's own objective.  Every evaluated point must agree.

The two codes share Kraft's SLSQPB/LSQ/LSEI/LSI/LDP/NNLS/H12/LDL; NLopt's deviations
(first trial evaluated with gradient, x clipped into the bounds, its own stopping
rules) do not change the iterates, so this pins the restatement to roundoff.  Long
Panda runs with several joints on their limits amplify a 1e-17 difference (BLAS
summation order) geometrically, hence the per-evaluation envelopes below."""
import numpy as np
import pytest

slsqp = pytest.importorskip("scipy.optimize._slsqp").slsqp


def _scipy_points(oracle, ch, tgt, x0, lb, ub, max_evals):
    n = len(x0)
    m = meq = 0
    la, n1 = 1, n + 1
    mineq = m - meq + n1 + n1
    len_w = ((3 * n1 + m) * (n1 + 1) + (n1 - meq + 1) * (mineq + 2) + 2 * mineq
             + (n1 + mineq) * (n1 - meq) + 2 * meq + n1 + ((n + 1) * n) // 2 + 2 * m + 3 * n
             + 3 * n1 + 1)
    w = np.zeros(len_w)
    jw = np.zeros(mineq, dtype=np.int32)
    x = np.array(x0, dtype=float)
    mode, acc, majiter = np.array(0, int), np.array(0.0), np.array(1000000, int)
    fl = [np.array(0.0) for _ in range(10)]
    il = [np.array(0, int) for _ in range(8)]
    f, g = oracle.eval_fg(ch, tgt, x)
    g = np.append(g, 0.0)
    c, a = np.zeros(la), np.zeros((la, n + 1))
    pts = [x.copy()]
    while len(pts) < max_evals:
        slsqp(m, meq, x, lb.copy(), ub.copy(), f, c, g, a, acc, majiter, mode, w, jw, *fl, *il)
        if mode == 1:
            f = oracle.eval_fg(ch, tgt, x, grad=False)
            pts.append(x.copy())
        elif mode == -1:
            _, gg = oracle.eval_fg(ch, tgt, x)
            g = np.append(gg, 0.0)
        else:
            break
    return pts


def _oracle_points(oracle, ch, tgt, x0, **cfg):
    res, tr = oracle.solve_restart(ch, oracle.make_config(**cfg), tgt, x0, 0, trace_cap=2000)
    rows = [tr[0, :-1]]
    for r in tr[1:]:
        if not np.array_equal(r[:-1], rows[-1]):  # NLopt re-evaluates the accepted point
            rows.append(r[:-1])
    return res, rows


@pytest.mark.parametrize("robot,tol_f,ncase,seed", [
    ("ur3e", 1e-6, 25, 1), ("ur10", 1e-12, 25, 3), ("panda", 1e-6, 40, 2),
    ("panda_hand", 1e-10, 25, 5)])
def test_every_evaluated_point_matches_scipy_kraft(oracle, chains, robot, tol_f, ncase, seed):
    d, ch = chains[robot]
    rng = np.random.default_rng(seed)
    n_exact = 0
    for _ in range(ncase):
        qt = rng.uniform(d["lb"], d["ub"])
        _, tgt = oracle.fk(ch, qt)
        x0 = rng.uniform(d["lb"], d["ub"])
        res, rows = _oracle_points(oracle, ch, tgt, x0, tol_f=tol_f)
        pts = _scipy_points(oracle, ch, tgt, x0, d["lb"], d["ub"], len(rows))
        assert len(pts) == len(rows)  # same number of trial points: same line-search decisions
        diffs = np.array([np.abs(p - r).max() for p, r in zip(pts, rows)])
        assert diffs[:12].max() < 1e-11      # early evaluations: roundoff only
        assert diffs[:30].max() < 1e-7       # envelope of geometric roundoff growth
        if diffs.max() < 1e-9:
            n_exact += 1
        assert res.result in (oracle.RES_STOPVAL, oracle.RES_FTOL)
    assert n_exact >= int(0.85 * ncase)
