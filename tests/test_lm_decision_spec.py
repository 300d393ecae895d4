"""The device-side LM decision (block_reduce_publish, dvm_slam_amd/csrc/ba_kernels.hip) forms g2o's
`alpha = 1 - pow(2 * rho - 1, 3)` (optimization_algorithm_levenberg.cpp:120-122) without libm: t*t and (t*t)*t with the rounding
error of both products carried along by FMAs.  The host keeps what the device enqueued only when its own std::pow result has the
same bits, so a disagreement costs speed, never correctness -- this test pins how rare it is: the device formula restated with
exact rational arithmetic must equal the correctly rounded cube always, and libm's pow almost always."""
import math
from fractions import Fraction

import numpy as np


def _fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))   # one rounding, like v_fma_f64


def _device_cube(t):
    t2 = t * t
    e2 = _fma(t, t, -t2)
    t3 = t2 * t
    e3 = _fma(t2, t, -t3)
    return t3 + (e3 + e2 * t)


def test_device_cube_is_the_correctly_rounded_cube_and_matches_libm():
    rng = np.random.default_rng(5)
    # 2 * rho - 1 for gain ratios of accepted trials: rho in (0, ~2]; plus values near the interval ends
    ts = np.concatenate([rng.uniform(-1.0, 3.0, 20000), rng.uniform(-1e-3, 1e-3, 2000), 1.0 - rng.uniform(0, 1e-6, 2000)])
    exact_miss = libm_miss = 0
    for t in ts.tolist():
        d = _device_cube(t)
        exact = float(Fraction(t) ** 3)
        exact_miss += d != exact
        libm_miss += d != math.pow(t, 3)
    assert exact_miss <= len(ts) // 2000, exact_miss      # (a double-rounding tie is possible in principle)
    assert libm_miss <= len(ts) // 500, libm_miss
