"""Helpers shared by the -m gpu parity tests (HIP path vs CPU oracle)."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(got, want, what=""):
    got = np.ascontiguousarray(got, dtype=np.float64)
    want = np.ascontiguousarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    # +0.0 == -0.0 and NaN == NaN count as equal: only the sign of an exact zero may
    # differ (the kernels skip multiplications by structural zeros, see ik_slsqp.hpp)
    same = (bits(got) == bits(want)) | ((got == 0.0) & (want == 0.0)) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        bad = np.argwhere(~same)
        i = tuple(bad[0])
        raise AssertionError(
            f"{what}: {len(bad)} of {got.size} values differ; first at {i}: "
            f"gpu={got[i]!r} oracle={want[i]!r} (diff {got[i] - want[i]:.3e})")


def make_targets(oracle, chain_dict, ch, rng, T):
    """Reachable targets FK(q*) and in-limit seeds, as examples/example.rs:24-26."""
    tg, x0 = [], []
    for _ in range(T):
        qt = rng.uniform(chain_dict["lb"], chain_dict["ub"])
        _, ee = oracle.fk(ch, qt)
        tg.append(ee)
        x0.append(rng.uniform(chain_dict["lb"], chain_dict["ub"]))
    return np.array(tg), np.array(x0)
