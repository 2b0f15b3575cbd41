"""The reciprocal-table division the ME and LF kernels use for per-lane divisions by small wave-uniform divisors
(me_udiv / lf_udiv: inv = floor((2^32 - 1) / d) + 1, quotient = mulhi(t, inv), d = 1 -> inv wraps to 0 -> t):
exact for every divisor of the tables and every dividend the kernels can produce (t < 2^16)."""
import numpy as np


def test_magic_division_exact():
    t = np.arange(1 << 16, dtype=np.uint64)
    for d in range(1, 257):
        inv = (0xFFFFFFFF // d + 1) & 0xFFFFFFFF
        q = t if inv == 0 else (t * np.uint64(inv)) >> np.uint64(32)
        assert np.array_equal(q, t // np.uint64(d)), d


def test_magic_division_bound():
    # the identity holds while t * d < 2^32; the kernels stay far below (t < 2^16, d <= 256)
    rng = np.random.default_rng(3)
    for d in (3, 7, 21, 53, 255, 256):
        inv = np.uint64((0xFFFFFFFF // d + 1) & 0xFFFFFFFF)
        t = rng.integers(0, (1 << 32) // d, 200000, dtype=np.uint64)
        assert np.array_equal((t * inv) >> np.uint64(32), t // np.uint64(d)), d
