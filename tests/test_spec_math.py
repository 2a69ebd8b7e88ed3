"""Accuracy of the spec's own math routines (they replace libm so that CPU and GPU agree bit for bit)."""
import numpy as np
import pytest


def test_sincos_accuracy(orc):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-200, 200, 200000), rng.uniform(-4, 4, 100000), np.linspace(-7000, 7000, 50001)]).astype(np.float32)
    s, c = orc.sincos(x)
    xd = x.astype(np.float64)
    assert np.max(np.abs(s - np.sin(xd))) < 3e-7
    assert np.max(np.abs(c - np.cos(xd))) < 3e-7
    assert np.max(np.abs(s.astype(np.float64) ** 2 + c.astype(np.float64) ** 2 - 1)) < 5e-7
    s0, c0 = orc.sincos(np.zeros(1, np.float32))
    assert s0[0] == 0.0 and c0[0] == 1.0


def test_atan2_accuracy_and_quadrants(orc):
    rng = np.random.default_rng(1)
    y = rng.standard_normal(300000).astype(np.float32) * 10 ** rng.uniform(-6, 6, 300000).astype(np.float32)
    x = rng.standard_normal(300000).astype(np.float32) * 10 ** rng.uniform(-6, 6, 300000).astype(np.float32)
    r = orc.atan2(y, x)
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.max(np.abs(r - ref)) < 4e-7
    pts = np.array([[0, 1], [1, 0], [0, -1], [-1, 0], [1, 1], [-1, -1], [1, -1], [-1, 1], [0, 0]], dtype=np.float32)
    r = orc.atan2(pts[:, 0], pts[:, 1])
    assert np.allclose(r[:8], np.arctan2(pts[:8, 0], pts[:8, 1]), atol=2e-7) and r[8] == 0.0


def test_log2_accuracy(orc):
    rng = np.random.default_rng(2)
    v = (10 ** rng.uniform(-9, 9, 200000)).astype(np.float32)
    r = orc.log2(v)
    ref = np.log2(v.astype(np.float64))
    assert np.max(np.abs(r - ref)) < 3e-6
    assert np.max(np.abs(r - ref) / np.maximum(1, np.abs(ref))) < 4e-7


def test_rsqrt_accuracy(orc):
    """spec rule 10: three Newton steps from the exponent-halving seed"""
    rng = np.random.default_rng(4)
    # every mantissa of two adjacent binades (the seed's error has period two in the exponent) + a wide range
    m = (np.arange(1 << 24, dtype=np.uint32) | np.uint32(0x3F000000)).view(np.float32)
    v = np.concatenate([m, np.exp(rng.uniform(-80, 80, 1 << 20)).astype(np.float32)])
    r = orc.rsqrt(v).astype(np.float64)
    rel = np.abs(r * np.sqrt(v.astype(np.float64)) - 1.0)
    assert rel.max() < 1.5e-7, rel.max()
    assert orc.rsqrt(np.float32([1.0, 4.0, 0.25])).tolist() == pytest.approx([1.0, 0.5, 2.0], rel=2e-7)


def test_recip_accuracy(orc):
    """spec rule 11: three Newton steps from the exponent-negating seed"""
    rng = np.random.default_rng(5)
    m = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3F800000)).view(np.float32)         # every float of [1, 2)
    v = np.concatenate([m, np.exp(rng.uniform(-60, 60, 1 << 20)).astype(np.float32)])
    r = orc.recip(v).astype(np.float64)
    rel = np.abs(r * v.astype(np.float64) - 1.0)
    assert rel.max() < 1.5e-7, rel.max()
    assert not np.isfinite(orc.recip(np.float32([0.0]))[0])                                   # 1/0 still is no number


def test_fft64_spec_matches_numpy_and_libm_mode(orc):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((500, 64)) + 1j * rng.standard_normal((500, 64))).astype(np.complex64)
    a = orc.fft64(x, orc.MATH_SPEC)
    b = orc.fft64(x, orc.MATH_LIBM)
    ref = np.fft.fftshift(np.fft.fft(x.astype(np.complex128), axis=1), axes=1)
    scale = np.abs(ref).max()
    assert np.max(np.abs(a - ref)) / scale < 5e-7
    assert np.max(np.abs(b - ref)) / scale < 2e-7
    # an impulse at n=1 gives the twiddle row exactly (bin i <-> k = i-32)
    e = np.zeros((1, 64), np.complex64); e[0, 1] = 1
    k = np.arange(-32, 32)
    assert np.allclose(orc.fft64(e)[0], np.exp(-2j * np.pi * k / 64), atol=1e-7)


def test_viterbi_corrects_errors(orc):
    from wifirx import txgen
    rng = np.random.default_rng(4)
    for trial in range(20):
        bits = rng.integers(0, 2, 400).astype(np.uint8)
        bits[-6:] = 0
        coded = txgen.conv_encode(bits[None])[0].copy()
        assert np.array_equal(orc.viterbi(coded, 400), bits)
        pos = rng.choice(800, 12, replace=False)          # sparse errors are corrected
        pos = pos[np.argsort(pos)]
        if np.min(np.diff(pos)) > 14:
            c2 = coded.copy(); c2[pos] ^= 1
            assert np.array_equal(orc.viterbi(c2, 400), bits)
        c3 = coded.copy(); c3[rng.choice(800, 150, replace=False)] = 2     # erasures (punctured bits)
        assert np.array_equal(orc.viterbi(c3, 400), bits)
