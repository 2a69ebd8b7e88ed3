"""The device channel `synth_kernel` (csrc/wr_synth.hip), which makes config 2's and config 3's inputs: unit-variance
complex AWGN, signal gain sqrt(10^(snr/10)) against it (gnu_radio/IRS_tranceiver.py:283,294), per-slot CFO uniform in
+-cfo_max (the range of IRS_tranceiver.py:121), and -- noise off -- the same samples txgen.impair() puts into a slot."""
import numpy as np
import pytest

from wifirx import txgen

pytestmark = pytest.mark.gpu

SLOT, LEAD = 4608, 160


@pytest.fixture(scope="module")
def rx():
    from wifirx import capi
    r = capi.WifiRx(max_sym=50, llr_bits=0)
    yield r
    r.close()


@pytest.fixture(scope="module")
def tx():
    return txgen.encode_psdus(txgen.make_psdus(16, 294, seed=77), 2)


def _synth(rx, tx, n_slots, snr_db, cfo_max, seed):
    slots = rx.alloc(n_slots * SLOT * 8)
    cfo = rx.alloc(n_slots * 4)
    rx.synth_slots(tx.samples, slots.ptr, SLOT, n_slots, LEAD, snr_db, cfo_max, seed, cfo.ptr)
    x = slots.download(np.complex64, n_slots * SLOT).reshape(n_slots, SLOT)
    c = cfo.download(np.float32, n_slots)
    slots.free()
    cfo.free()
    return x, c


def test_noise_is_unit_variance_white_gaussian(rx, tx):
    n = 1024
    x, _ = _synth(rx, tx, n, 20.0, 0.037, 4242)
    flen = tx.samples.shape[1]
    noise = np.concatenate([x[:, :LEAD], x[:, LEAD + flen:]], axis=1).astype(np.complex128).reshape(-1)
    N = noise.size
    assert N > 200_000
    p = np.mean(np.abs(noise) ** 2)
    assert abs(p - 1.0) < 0.01, p                                   # noise_voltage = 1 (IRS_tranceiver.py:283)
    assert abs(np.var(noise.real) - 0.5) < 0.01 and abs(np.var(noise.imag) - 0.5) < 0.01
    assert abs(np.mean(noise)) < 5 / np.sqrt(N)
    assert abs(np.mean(noise.real * noise.imag)) < 5 * 0.5 / np.sqrt(N)           # I and Q uncorrelated
    assert abs(np.mean(np.abs(noise) ** 4) - 2.0) < 0.05                           # Gaussian: E|n|^4 = 2
    rows = x[:, LEAD + flen:].astype(np.complex128)
    lag1 = np.mean(rows[:, 1:] * np.conj(rows[:, :-1]))
    assert abs(lag1) < 5 / np.sqrt(rows.size)                                      # white
    # slots are independent draws: the same sample position of two slots is uncorrelated
    assert abs(np.mean(rows[:-1] * np.conj(rows[1:]))) < 5 / np.sqrt(rows.size)


def test_signal_gain_against_the_noise(rx, tx):
    n = 512
    flen = tx.samples.shape[1]
    p_tx = np.mean(np.abs(tx.samples.astype(np.complex128)) ** 2, axis=1)           # per template
    for snr_db in (5.0, 20.0, 30.0):
        x, _ = _synth(rx, tx, n, snr_db, 0.037, 99)
        p = np.mean(np.abs(x[:, LEAD:LEAD + flen].astype(np.complex128)) ** 2, axis=1)
        g2 = np.mean((p - 1.0) / p_tx[np.arange(n) % 16])
        assert abs(g2 / 10 ** (snr_db / 10) - 1.0) < 0.01, (snr_db, g2)             # IRS_tranceiver.py:294


def test_cfo_is_uniform_within_the_range(rx, tx):
    from scipy import stats
    cmax = 0.037
    n = 100_000
    slots = rx.alloc(n * 64 * 8)              # tiny slots: only the per-slot CFO draw matters here
    cfo = rx.alloc(n * 4)
    rx.synth_slots(tx.samples[:, :32], slots.ptr, 64, n, 0, 20.0, cmax, 31337, cfo.ptr)
    c = cfo.download(np.float32, n).astype(np.float64)
    slots.free(); cfo.free()
    assert c.min() >= -cmax and c.max() <= cmax
    assert c.min() < -0.999 * cmax and c.max() > 0.999 * cmax
    assert abs(c.mean()) < 5 * cmax / np.sqrt(3 * n)
    assert abs(c.var() - cmax ** 2 / 3) < 0.02 * cmax ** 2 / 3
    assert stats.kstest(c, stats.uniform(loc=-cmax, scale=2 * cmax).cdf).pvalue > 1e-3
    # a different seed gives different draws, the same seed the same ones
    cfo2 = rx.alloc(n * 4)
    slots = rx.alloc(n * 64 * 8)
    rx.synth_slots(tx.samples[:, :32], slots.ptr, 64, n, 0, 20.0, cmax, 31337, cfo2.ptr)
    assert np.array_equal(cfo2.download(np.float32, n).astype(np.float64), c)
    rx.synth_slots(tx.samples[:, :32], slots.ptr, 64, n, 0, 20.0, cmax, 31338, cfo2.ptr)
    assert abs(np.corrcoef(cfo2.download(np.float32, n), c)[0, 1]) < 0.02
    slots.free(); cfo2.free()


def test_noiseless_part_equals_txgen_impair(rx, tx):
    """snr_db = NaN switches the AWGN off (gain 1): without CFO the slot is bit for bit what txgen.impair(snr_db=None)
    makes; with CFO it is the same up to the float32 rotation angle cfo*m of the device kernel."""
    n = 64
    x, c = _synth(rx, tx, n, float("nan"), 0.0, 5)
    ref = txgen.impair(tx.samples[np.arange(n) % 16], None, cfo=0.0, lead=LEAD, total=SLOT)
    assert np.array_equal(c, np.zeros(n, np.float32))
    assert np.array_equal(x, ref)
    x, c = _synth(rx, tx, n, float("nan"), 0.037, 5)
    ref = txgen.impair(tx.samples[np.arange(n) % 16], None, cfo=c.astype(np.float64), lead=LEAD, total=SLOT)
    assert np.abs(c).max() > 0.02
    # |cfo*m| reaches 160 rad, where a float32 angle has an ulp of 1.5e-5 rad
    assert np.abs(x - ref).max() <= 3e-5 * np.abs(ref).max()
    assert np.array_equal(x[:, :LEAD], np.zeros((n, LEAD), np.complex64))


def test_noise_does_not_depend_on_the_signal(rx, tx):
    """counter-based RNG: the noise of a slot depends on (seed, slot, sample) only -- the noisy slot minus the
    noiseless one is the same whatever the gain"""
    n = 32
    clean, _ = _synth(rx, tx, n, float("nan"), 0.0, 8)
    a, _ = _synth(rx, tx, n, 0.0, 0.0, 8)            # gain 1
    b, _ = _synth(rx, tx, n, 20.0, 0.0, 8)           # gain 10
    na = a.astype(np.complex128) - clean
    nb = b.astype(np.complex128) - 10.0 * clean
    assert np.abs(na - nb).max() < 2e-5               # float32 rounding of the sums only
