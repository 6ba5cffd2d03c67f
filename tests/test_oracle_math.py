"""The oracle's elementary functions (shared as an operation sequence with the HIP
kernel so that GPU == oracle bit-for-bit) stay within ~1 ulp of the true value."""
import ctypes as C

import mpmath as mp
import numpy as np


def test_sincos_and_atan2_accuracy(oracle):
    L = oracle.lib()
    L.ok_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ok_atan2_q1.argtypes = [C.c_double, C.c_double]
    L.ok_atan2_q1.restype = C.c_double
    mp.mp.prec = 200
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-7, 7, 4000), rng.uniform(-1e-3, 1e-3, 500),
                         rng.uniform(-1000, 1000, 500),
                         [0.0, np.pi / 2, np.pi, -np.pi, 1.5 * np.pi, 1e-300, np.pi / 4]])
    worst = 0.0
    for x in xs:
        s, c = C.c_double(), C.c_double()
        L.ok_sincos(float(x), C.byref(s), C.byref(c))
        ts, tc = mp.sin(mp.mpf(float(x))), mp.cos(mp.mpf(float(x)))
        for got, true in ((s.value, ts), (c.value, tc)):
            u = np.spacing(abs(float(true))) or 5e-324
            worst = max(worst, float(abs(mp.mpf(got) - true) / u))
    assert worst < 1.0, worst
    worst = 0.0
    for _ in range(4000):
        y = float(rng.uniform(1e-3, 1))
        x = float(rng.uniform(0, 1))
        r = rng.random()
        if r < 0.05:
            x = 0.0
        elif r < 0.1:
            x = float(rng.uniform(0, 1e-6))
        true = mp.atan2(mp.mpf(y), mp.mpf(x))
        worst = max(worst, float(abs(mp.mpf(L.ok_atan2_q1(y, x)) - true) / np.spacing(float(true))))
    assert worst < 1.5, worst
