"""Exhaustive proof of the division-free depth conversion used by the HIP kernel (csrc/fusion.hip, depth_to_metres):
for EVERY u16 depth d the sequence  q0 = d*r ; e = fma(-q0, 1000, d) ; q = fma(e, r, q0)  with r = fl32(1/1000)
is bit-identical to the reference's  float(d) / 1000.0f  (src/NativeUtils/depthprocessing.cpp:149-150).
Exact rational arithmetic, no floating-point shortcuts."""
from fractions import Fraction

import numpy as np


def rn32(F):
    """Correct rounding of a Fraction to float32 (nearest, ties to even)."""
    if F == 0:
        return np.float32(0)
    c = np.float32(float(F))
    best = None
    for x in (np.nextafter(c, np.float32(-np.inf)), c, np.nextafter(c, np.float32(np.inf))):
        err = abs(Fraction(float(x)) - F)
        if best is None or err < best[0] or (err == best[0] and (int(x.view(np.uint32)) & 1) == 0):
            best = (err, x)
    return best[1]


def test_three_instruction_quotient_is_exact_for_every_u16_depth():
    r = np.float32(1.0) / np.float32(1000.0)
    assert int(r.view(np.uint32)) == 0x3A83126F            # the literal in fusion.hip
    R = Fraction(float(r))
    for d in range(1, 65536):
        D = Fraction(d)
        want = np.float32(d) / np.float32(1000.0)           # IEEE-754 division
        assert want == rn32(D / 1000)
        q0 = rn32(D * R)
        e = rn32(D - Fraction(float(q0)) * 1000)
        q = rn32(Fraction(float(q0)) + Fraction(float(e)) * R)
        assert q == want, d


def test_integer_triangle_threshold_equals_the_double_expression():
    """csrc/fusion.hip computes checkTriangleConstraints' depth_thr (src/NativeUtils/meshGenerator.cpp:26,
    (int)((v0+v1+v2) / 3.0 * 0.00272 + 7.273) in double) as (272 s + 2181900) / 300000 in integers: equal for every
    possible sum s of three u16 depths."""
    s = np.arange(0, 3 * 65535 + 1, dtype=np.int64)
    ref = np.floor(s.astype(np.float64) / 3.0 * 0.00272 + 7.273).astype(np.int64)   # truncation = floor (positive)
    assert np.array_equal((272 * s + 2181900) // 300000, ref)


def test_division_free_edge_test_equals_the_threshold_compare():
    """csrc/mesh.hip (edges_pass): `metric < (272 s + 2181900) / 300000` is evaluated as `18750 * metric <= 17 * s + 117618` in 32-bit
    unsigned arithmetic on 24-bit factors.  For every sum s of three u16 depths the largest metric the product form accepts is exactly
    threshold - 1 (both forms are monotone in the metric, so the boundary decides every metric), nothing overflows, and the factors fit
    v_mul_u32_u24's 24 bits for every metric an edge can have (|vA - vB| <= 65535 bounds the minimum of its three differences)."""
    s = np.arange(0, 3 * 65535 + 1, dtype=np.int64)
    thr = (272 * s + 2181900) // 300000
    rhs = 17 * s + 117618
    largest_accepted = rhs // 18750                       # max metric with 18750 * metric <= rhs
    assert np.array_equal(largest_accepted, thr - 1)
    assert int(rhs.max()) < 2 ** 32 and 18750 * 65535 < 2 ** 32
    assert 65535 < 2 ** 24 and 18750 < 2 ** 24 and int(s.max()) < 2 ** 24 and 17 < 2 ** 24
    # spot check of the two predicates themselves around the boundary and at the extremes
    for m in (0, 1, 7, 8, 184, 185, 186, 65535):
        assert np.array_equal(m < thr, 18750 * m <= rhs), m
    # max(m0, m1, m2) < thr  <=>  every metric < thr
    rng = np.random.default_rng(5)
    m3 = rng.integers(0, 260, size=(200000, 3))
    t = rng.integers(0, 3 * 65535 + 1, size=200000)
    th = (272 * t + 2181900) // 300000
    assert np.array_equal((m3 < th[:, None]).all(axis=1), 18750 * m3.max(axis=1) <= 17 * t + 117618)
