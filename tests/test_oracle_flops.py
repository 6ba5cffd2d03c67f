"""The oracle's operation counter (oracle/optik_oracle_flops.cpp; SURVEY.md section 8d: "measure with the oracle's
counters"): the counting build returns the plain build's bits, and a restart costs a few hundred thousand f64
operations -- the algorithmic numerator of bench.py's roofline.secondary.frac_algorithmic."""
import os

import numpy as np

from conftest import ROBOTS


def test_counting_build_is_bit_identical_and_counts(oracle):
    from oracle import urdf_chain
    with open(os.path.join(ROBOTS, "panda.urdf")) as fh:
        d = urdf_chain.chain_from_urdf(fh.read(), "panda_link0", "panda_link8")
    ch = oracle.make_chain(**d)
    rng = np.random.default_rng(0)
    _, tgt = oracle.fk(ch, rng.uniform(d["lb"], d["ub"]))
    x0 = rng.uniform(d["lb"], d["ub"])
    cfg = oracle.make_config(solution_mode="speed")
    R = 300
    ref = oracle.ik(ch, cfg, tgt, x0, 0, R, n_threads=2, early_exit=False, per_restart=True)
    try:
        oracle.use_flops_build()
        oracle.flop_reset()
        out = oracle.ik(ch, cfg, tgt, x0, 0, R, n_threads=1, early_exit=False, per_restart=True)
        c = oracle.flop_counts()
        # an evaluation alone
        oracle.flop_reset()
        oracle.eval_fg(ch, tgt, x0)
        e = oracle.flop_counts()
    finally:
        oracle.use_portable_build()
    for k in ("status", "evals"):
        assert np.array_equal(out[k], ref[k])
    assert np.array_equal(out["xs"].view(np.uint64), ref["xs"].view(np.uint64))
    assert np.array_equal(out["fs"].view(np.uint64), ref["fs"].view(np.uint64))
    per = c["flops"] / R
    evals = ref["evals"].mean()
    assert 5e4 < per < 2e6, per
    assert c["flops"] == c["add_sub"] + c["mul"] + c["div"] + c["sqrt"]
    # objective + gradient of a 7-joint chain: a few thousand operations (SURVEY 8d estimated ~2.3 k)
    assert 1500 < e["flops"] < 8000, e
    # ... and a restart is its evaluations plus the SQP work per iteration: several times the evaluations alone
    assert per > evals * e["flops"]
