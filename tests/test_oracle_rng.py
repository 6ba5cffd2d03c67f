"""Known-answer tests for the oracle's restatement of the un-vendored RNG crates
(rand_core 0.9.3 seed_from_u64, rand_chacha 0.9.0 ChaCha8, rand 0.9.2 uniform f64;
call sites lib.rs:358-370, 86-91).  SURVEY appendix C."""
import ctypes as C
import os

import numpy as np
import pytest

# ChaCha8, zero key, zero counter, zero stream: published keystream block 0.
CHACHA8_ZERO_KAT = (
    "3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e"
    "984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42")


def _block(oracle, key_bytes, counter, stream, rounds):
    key = (C.c_uint32 * 8).from_buffer_copy(key_bytes)
    out = (C.c_uint32 * 16)()
    oracle.lib().ok_chacha_block(key, counter, stream, rounds, out)
    return bytes(out)


def test_chacha8_zero_key_kat(oracle):
    assert _block(oracle, bytes(32), 0, 0, 8).hex() == CHACHA8_ZERO_KAT


def test_chacha20_matches_libsodium(oracle):
    path = "/opt/conda/lib/libsodium.so.23"
    if not os.path.exists(path):
        pytest.skip("libsodium not present")
    s = C.CDLL(path)
    key = bytes(range(32))
    nonce = 0x0123456789ABCDEF
    buf = C.create_string_buffer(192)
    s.crypto_stream_chacha20(buf, C.c_ulonglong(192), nonce.to_bytes(8, "little"), key)
    ours = b"".join(_block(oracle, key, c, nonce, 20) for c in range(3))
    assert buf.raw == ours


def test_seed_from_u64_42(oracle):
    key = (C.c_uint32 * 8)()
    oracle.lib().ok_seed_from_u64(42, key)
    # provisional value recorded in SURVEY appendix C (PCG32 expansion of 42)
    assert bytes(key).hex() == "a48fa17b58323d0aeab8a1cc690114b82b8cc87518b4f7548d446ea1e4df20f2"


def test_uniform_inclusive_properties(oracle):
    f = oracle.lib().ok_uniform_inclusive
    lo, hi = -2.8973, 2.8973
    assert f(lo, hi, 0) == lo                      # all-zero mantissa -> low
    top = f(lo, hi, 0xFFFFFFFFFFFFFFFF)            # max mantissa -> <= high, within 1 ulp-ish
    assert top <= hi and hi - top < 1e-15
    rng = np.random.default_rng(0)
    for bits in rng.integers(0, 2**63, size=1000, dtype=np.uint64):
        v = f(lo, hi, int(bits))
        assert lo <= v <= hi
    # only the top 52 bits matter (u64 >> 12)
    assert f(lo, hi, 0xABCDEF0123456000) == f(lo, hi, 0xABCDEF0123456FFF)


def test_restart_seeds_ur3e(oracle, chains):
    _, ch = chains["ur3e"]
    s1, s2 = oracle.restart_seed(ch, 1), oracle.restart_seed(ch, 2)
    np.testing.assert_allclose(
        s1, [1.362142, -2.092864, -0.110490, 2.175655, -2.723413, 1.719957], atol=1e-6)
    np.testing.assert_allclose(
        s2, [-1.987935, -1.292079, 0.090641, -2.383262, -1.419787, -0.183156], atol=1e-6)
    # stream id = restart index: distinct indices -> distinct seeds, within limits
    d, _ = chains["ur3e"]
    seen = set()
    for i in range(1, 200):
        s = oracle.restart_seed(ch, i)
        assert np.all(s >= d["lb"]) and np.all(s <= d["ub"])
        seen.add(tuple(s))
    assert len(seen) == 199


def test_rand_chacha_construction_kat(oracle):
    """rand_chacha's own unit test `test_chacha_construction` (src/chacha.rs, recalled from the crate:
    seed bytes [0 x8, 1, 0 x7, 2, 0 x7, 3, 0 x7], ChaCha20Rng::from_seed(seed).next_u32() ==
    137206642): pins the seed-bytes -> key-words order (little endian) and the zero counter /
    stream layout of the block function the restart seeds use (there with 8 rounds)."""
    seed = bytes([0] * 8 + [1] + [0] * 7 + [2] + [0] * 7 + [3] + [0] * 7)
    block = _block(oracle, seed, 0, 0, 20)
    assert int.from_bytes(block[:4], "little") == 137206642
