"""CPU: the oracle's baseline-only modes (bench.py's cpu_baseline leg; never used by a parity test):
find_any (the reference's multi-thread rule, lib.rs:409-412), the persistent worker pool (rayon's pool), ok_ik_many
(BASELINE config 5 on the CPU).  They must return what the plain oracle returns for the restart they pick."""
import os

import numpy as np

from conftest import ROBOTS


def _panda(oracle):
    from oracle import urdf_chain
    d = urdf_chain.chain_from_urdf(open(os.path.join(ROBOTS, "panda.urdf")).read(), "panda_link0", "panda_link8")
    return d, oracle.make_chain(**d)


def test_find_any_returns_a_restart_the_plain_oracle_also_solves(oracle):
    d, ch = _panda(oracle)
    rng = np.random.default_rng(3)
    lb, ub = np.array(d["lb"]), np.array(d["ub"])
    cfg = oracle.make_config(solution_mode="speed")
    oracle.pool_start(4)
    try:
        for _ in range(20):
            tgt = oracle.fk(ch, rng.uniform(lb, ub))[1]
            x0 = rng.uniform(lb, ub)
            ref = oracle.ik(ch, cfg, tgt, x0, 0, 64, n_threads=1, early_exit=False, per_restart=True)
            for nt in (1, 4):   # 4: through the pool
                got = oracle.ik(ch, cfg, tgt, x0, 0, 64, n_threads=nt, early_exit="find_any")
                assert got["found"] == bool(ref["success"].any())
                if got["found"]:
                    w = got["winner"]
                    assert ref["success"][w]
                    assert np.array_equal(got["x"].view(np.int64), ref["xs"][w].view(np.int64))
            det = oracle.ik(ch, cfg, tgt, x0, 0, 64, n_threads=4, early_exit=True)    # the pool, deterministic rule
            assert det["found"] == ref["found"] and (not det["found"] or det["winner"] == ref["winner"])
    finally:
        oracle.pool_stop()


def test_ik_many_equals_individual_calls(oracle):
    d, ch = _panda(oracle)
    rng = np.random.default_rng(4)
    lb, ub = np.array(d["lb"]), np.array(d["ub"])
    cfg = oracle.make_config(solution_mode="speed", max_restarts=32)
    T = 40
    tg = np.array([oracle.fk(ch, rng.uniform(lb, ub))[1] for _ in range(T)])
    x0 = rng.uniform(lb, ub, size=(T, 7))
    found, xs, _ = oracle.ik_many(ch, cfg, tg, x0, 32, 3)
    for t in range(T):
        one = oracle.ik(ch, cfg, tg[t], x0[t], 0, 32, n_threads=1, early_exit=True)
        assert bool(found[t]) == one["found"]
        if one["found"]:
            assert np.array_equal(xs[t].view(np.int64), one["x"].view(np.int64))
