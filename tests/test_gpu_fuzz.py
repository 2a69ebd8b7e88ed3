"""Hostile inputs: the chain must neither fault nor hang, and for finite samples stay bit-identical to the
oracle (false alarms, repeated preambles, extreme amplitudes incl. the denormal range, clipped signals)."""
import numpy as np
import pytest

from wifirx import txgen

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(2024)
    n, L = 32, 2048

    def noise(s):
        return ((rng.standard_normal((n, L)) + 1j * rng.standard_normal((n, L))) * s).astype(np.complex64)
    yield "noise_unit", noise(1.0)
    yield "noise_1e30", noise(1e15)                       # |x|^2 near the top of the float range
    yield "noise_denormal", noise(1e-21)                  # |x|^2 ~ 1e-42: denormal products
    psdu = txgen.make_psdus(n, 100, seed=1)
    tx = txgen.encode_psdus(psdu, 4)
    fr = txgen.impair(tx.samples, 18.0, cfo=rng.uniform(-0.05, 0.05, n), lead=150, total=L, seed=3)
    yield "frames_tiny", (fr * np.float32(1e-18)).astype(np.complex64)
    yield "frames_huge", (fr * np.float32(1e12)).astype(np.complex64)
    clipped = fr.copy()
    clipped.real = np.clip(clipped.real, -3, 3); clipped.imag = np.clip(clipped.imag, -3, 3)
    yield "frames_clipped", clipped
    sts = tx.samples[:, :160]
    rep = np.tile(sts, (1, L // 160 + 1))[:, :L] * 5 + noise(0.3)          # short preamble forever: plateau everywhere
    yield "sts_forever", rep.astype(np.complex64)
    lts_only = np.zeros((n, L), np.complex64)
    lts_only[:, 200:200 + 160] = tx.samples[:, 160:320] * 5
    yield "lts_without_sts", (lts_only + noise(0.2)).astype(np.complex64)
    const = np.full((n, L), 3 + 4j, np.complex64)                           # DC: autocorrelation = power
    yield "dc_only", const
    alt = (fr * 10).astype(np.complex64)
    alt[:, 700:] = 0                                                          # frame cut to exact zeros mid-way
    yield "cut_to_zero", alt


CASES = dict(_cases())


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("chan_est", [0, 1])
def test_fuzz_bit_exact(orc, name, chan_est, decode_path):
    from wifirx import capi
    iq = np.ascontiguousarray(CASES[name]).reshape(-1)
    L = CASES[name].shape[1]
    rx = capi.WifiRx(max_sym=40, llr_bits=6, want_carrier=True, chan_est=chan_est)
    r = rx.demod_batch(iq, L, decode=True, psdu_stride=512, want_csi=False)
    prm = orc.make_params(max_sym=40, llr_bits=6, chan_est=chan_est)
    o = orc.demod_batch(iq, L, prm, want_eq=True)
    opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=512)
    assert np.array_equal(r["frames"], o["frames"]), (name, r["frames"][:4], o["frames"][:4])
    assert np.array_equal(r["idx"], o["idx"])
    fin = np.isfinite(o["eq"])
    assert np.array_equal(np.isfinite(r["carrier"]), fin)
    assert np.array_equal(r["carrier"][fin], o["eq"][fin])
    lf = np.isfinite(o["llr"])
    assert np.array_equal(r["llr"][lf], o["llr"][lf])
    dec = (o["frames"]["flags"] & orc.F_DECODED) != 0
    for k in np.nonzero(dec)[0]:
        n = int(o["frames"]["psdu_len"][k])
        assert np.array_equal(r["psdu"][k, :n], opsdu[k, :n])
    rx.close()


def test_nan_and_inf_do_not_hang_or_fault():
    from wifirx import capi
    rng = np.random.default_rng(1)
    x = ((rng.standard_normal(64 * 1024) + 1j * rng.standard_normal(64 * 1024))).astype(np.complex64)
    x[::97] = np.nan
    x[5::1013] = np.inf
    rx = capi.WifiRx(max_sym=16, llr_bits=2)
    r = rx.demod_batch(x, 1024, decode=True, psdu_stride=512)
    assert r["frames"].shape == (64,)
    rx.push(x); rx.push(np.zeros(0, np.complex64))
    rx.poll()
    rx.close()
