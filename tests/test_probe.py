"""probe_mpsk_snr_est (wifirx/probe.py), the stand-in for digital.probe_mpsk_snr_est_c(0, 1000, 0.05) of
gnu_radio/IRS_AP.py:275,312: estimator arithmetic, message cadence, per-frame input against sample-by-sample input,
and the oracle's per-frame moments (spec rule 13) against float64."""
import numpy as np
import pytest

from wifirx import grshim, probe


def _qpsk(n, snr_db, seed):
    rng = np.random.default_rng(seed)
    s = (rng.choice([-1, 1], n) + 1j * rng.choice([-1, 1], n)) / np.sqrt(2)
    sigma2 = 10 ** (-snr_db / 10)
    return s + (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.sqrt(sigma2 / 2)


@pytest.mark.parametrize("typ", [probe.SNR_EST_SIMPLE, probe.SNR_EST_M2M4])
@pytest.mark.parametrize("snr_db", [10.0, 20.0])
def test_estimators_find_the_snr_of_qpsk(typ, snr_db):
    p = probe.probe_mpsk_snr_est(typ, 1000, 0.001)
    p.update(_qpsk(60000, snr_db, 1))
    assert abs(p.snr() - snr_db) < 0.6, p.snr()
    assert p.signal() > 0 and p.noise() > 0


def test_message_cadence_and_ports():
    p = probe.probe_mpsk_snr_est(0, 1000, 0.05)
    got = {"snr": [], "signal": [], "noise": []}
    for port in got:
        grshim.msg_connect(p, port, grshim.sink_block(got[port].append), "in")
    y = _qpsk(48 * 100, 15.0, 2)
    for k in range(100):                                    # 48-point PDUs, as the `symbols` / `carrier` port sends them
        p._handle_pdu(({}, y[48 * k:48 * k + 48].astype(np.complex64)))
    # upstream's loop `while (count > nsamples)`: 4800 samples, a message whenever more than 1000 have piled up
    assert len(got["snr"]) == 4 and len(got["signal"]) == 4 and len(got["noise"]) == 4
    assert all(abs(v - 15.0) < 2.0 for v in got["snr"])
    assert p.type() == 0 and p.msg_nsamples() == 1000 and p.alpha() == 0.05


def test_frame_input_composes_like_sample_input():
    y = _qpsk(48 * 50 * 40, 18.0, 3).reshape(40, 50 * 48)            # 40 frames of 50 symbols
    for typ, tol in ((probe.SNR_EST_SIMPLE, 1e-9), (probe.SNR_EST_M2M4, 0.5)):
        a = probe.probe_mpsk_snr_est(typ, 1000, 0.001)
        b = probe.probe_mpsk_snr_est(typ, 1000, 0.001)
        for f in y:
            a.update(f)
            m = np.abs(f)
            b.update_frame(m.sum(), (m ** 2).sum(), (m ** 4).sum(), m.size)
        assert abs(a.snr() - b.snr()) < tol, (typ, a.snr(), b.snr())
    with pytest.raises(ValueError):
        probe.probe_mpsk_snr_est(1, 1000, 0.05)                      # skewness estimator: not provided


def test_oracle_moments_match_float64(orc):
    rng = np.random.default_rng(5)
    eq = (rng.standard_normal((6, 20, 48)) + 1j * rng.standard_normal((6, 20, 48))).astype(np.complex64)
    n_out = np.array([20, 1, 0, 7, 20, 13])
    st = orc.sym_stats(eq, n_out)
    for i, n in enumerate(n_out):
        m = np.abs(eq[i, :n].astype(np.complex128)).reshape(-1)
        ref = np.array([m.sum(), (m ** 2).sum(), (m ** 4).sum(), 0.0])
        assert np.allclose(st[i], ref, rtol=2e-6, atol=0), (i, st[i], ref)
