"""CPU: the libm-sensitivity study runs and records its numbers (tools/libm_sensitivity.py).

The reference calls the platform's sin_cos / atan2 (/root/reference/crates/optik/src/math.rs:54,76,113,144); the oracle
and the GPU kernels share their own fdlibm sequence so that they can agree bit for bit.  The study runs the same
restarts through the oracle with either and counts how often a restart's outcome differs.  This test asserts only that
the study runs at 1/64 of its size, that its two builds are what they claim to be, and that the committed full-size
record (profiles/r6_libm_sensitivity.json) has the shape DESIGN.md quotes -- the numbers themselves are measurements,
not requirements.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

KEYS = ("frac_status_differs", "frac_x_differs_gt_1e-6", "frac_x_differs_gt_1e-6_among_both_solved",
        "speed_winner_index_changes", "quality_winner_index_changes", "frac_x_not_bit_equal")


def test_the_glibc_build_differs_from_the_oracle_in_the_last_bit_only():
    """-DOK_PLATFORM_LIBM routes ok_sincos / ok_atan2_q1 to glibc: same values to an ulp, not the same bits."""
    import math
    from oracle import binding as ob
    qs = np.random.default_rng(1).uniform(-3.0, 3.0, size=2000)
    try:
        ob.use_portable_build()
        a = np.array([ob.so3_log([math.sin(q / 2) * 0.6, math.sin(q / 2) * 0.8, 0.0, math.cos(q / 2)]) for q in qs])
        ob.use_libm_build()
        b = np.array([ob.so3_log([math.sin(q / 2) * 0.6, math.sin(q / 2) * 0.8, 0.0, math.cos(q / 2)]) for q in qs])
    finally:
        ob.use_portable_build()
    assert np.allclose(a, b, rtol=0, atol=2e-15)
    assert (a.view(np.int64) != b.view(np.int64)).any(), "the two atan2 implementations never differ: is the switch on?"


def test_study_runs_and_records(tmp_path):
    import libm_sensitivity
    out = libm_sensitivity.study(scale=1.0 / 64, verbose=False)
    assert set(out["cases"]) == {"config2_panda_65536", "config3_ur10_1M_tol1e-12", "config5_share_512x256"}
    for rec in out["cases"].values():
        for k in KEYS:
            assert rec[k] is None or 0.0 <= rec[k] <= 1.0, (k, rec[k])
        assert rec["restarts"] >= 256
    (tmp_path / "libm.json").write_text(json.dumps(out))


def test_committed_record_has_the_quoted_shape():
    with open(os.path.join(ROOT, "profiles", "r6_libm_sensitivity.json")) as fh:
        rec = json.load(fh)
    assert rec["scale"] == 1.0
    c = rec["cases"]
    assert c["config2_panda_65536"]["restarts"] == 65536
    assert c["config3_ur10_1M_tol1e-12"]["restarts"] == 1 << 20
    assert c["config5_share_512x256"]["targets"] == 512
    for r in c.values():
        for k in KEYS:
            assert k in r
