"""The -march=native build of the oracle that bench.py's cpu_baseline leg times gives the bits
of the portable build (same operation sequence: -ffp-contract=off, no fast-math)."""
import numpy as np


def test_native_build_is_bit_identical(oracle, chains):
    d, ch = chains["panda"]
    rng = np.random.default_rng(2)
    _, tgt = oracle.fk(ch, rng.uniform(d["lb"], d["ub"]))
    x0 = rng.uniform(d["lb"], d["ub"])
    cfg = oracle.make_config("quality")
    a = oracle.ik(ch, cfg, tgt, x0, 0, 96, n_threads=2, early_exit=False, per_restart=True)
    flags = oracle.use_native_build()
    try:
        b = oracle.ik(ch, cfg, tgt, x0, 0, 96, n_threads=2, early_exit=False, per_restart=True)
    finally:
        oracle.use_portable_build()
    if flags.startswith("portable"):
        return  # no compiler: nothing to compare
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["evals"], b["evals"])
    assert np.array_equal(a["xs"].view(np.uint64), b["xs"].view(np.uint64))
    assert np.array_equal(a["fs"].view(np.uint64), b["fs"].view(np.uint64))
    assert a["winner"] == b["winner"]
