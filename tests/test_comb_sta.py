"""Row f4: the COMB and STA equalisers (ieee802_11.COMB / ieee802_11.STA, gnu_radio/IRS_AP.py:139-141).

Upstream's sources are not vendored by the reference, so both follow the definitions of DESIGN.md section 4.11
(comb-pilot interpolation with temporal smoothing; spectral-temporal averaging after Fernandez et al.).  The CPU
tests pin what those definitions must deliver (loop-back at every rate, tracking of a frequency-selective
channel for STA, the known weakness of four-pilot interpolation for COMB); the GPU tests are bit-exact parity
against the oracle, batch and stream.
"""
import numpy as np
import pytest

from helpers import make_slots

TAPS = np.array([1.0, 0.35 + 0.2j, 0.15 - 0.1j, 0.05j], dtype=np.complex64)
TAPS /= np.linalg.norm(TAPS)


def _run(orc, iq, slot_len, tx, chan_est, **kw):
    prm = orc.make_params(max_sym=tx.n_sym, chan_est=chan_est, **kw)
    o = orc.demod_batch(iq, slot_len, prm, want_eq=True)
    psdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=2048)
    return o, psdu


@pytest.mark.parametrize("chan_est", [2, 3])
@pytest.mark.parametrize("encoding", range(8))
def test_loopback_every_rate(orc, chan_est, encoding):
    iq, slot_len, tx = make_slots(8, encoding, snr_db=32.0, seed=90 + encoding, psdu_len=150)
    o, psdu = _run(orc, iq, slot_len, tx, chan_est)
    assert (o["frames"]["flags"] & orc.F_CRC_OK).all()
    assert np.array_equal(psdu[:, :150], tx.psdu)
    assert np.array_equal(o["idx"][:, :tx.n_sym], tx.data_idx)


def test_sta_follows_a_selective_channel_and_comb_does_not(orc):
    """four taps over 150 ns: STA (per-bin estimates, +-2 bins) decodes 16-QAM 3/4 like LS; COMB's piecewise-linear
    estimate from four pilots cannot follow the ripple at 64-QAM"""
    iq, slot_len, tx = make_slots(32, 5, snr_db=25.0, seed=16, taps=TAPS, psdu_len=200)
    ok = {}
    for ce in (0, 2, 3):
        o, _ = _run(orc, iq, slot_len, tx, ce)
        ok[ce] = int(((o["frames"]["flags"] & orc.F_CRC_OK) != 0).sum())
    assert ok[0] == 32 and ok[3] == 32 and ok[2] >= 30
    iq, slot_len, tx = make_slots(32, 7, snr_db=32.0, seed=18, taps=TAPS, psdu_len=200)
    o, _ = _run(orc, iq, slot_len, tx, 3)
    assert ((o["frames"]["flags"] & orc.F_CRC_OK) != 0).sum() == 32
    o, _ = _run(orc, iq, slot_len, tx, 2)
    assert ((o["frames"]["flags"] & orc.F_CRC_OK) != 0).sum() < 8


def test_comb_and_sta_track_a_drifting_channel(orc):
    """a flat channel whose gain moves over the frame (Doppler-like): the LS estimate from the preamble goes stale,
    the per-symbol pilot estimate (COMB) and the decision-directed average (STA) follow"""
    from wifirx import txgen
    psdu = txgen.make_psdus(24, 600, seed=31)
    tx = txgen.encode_psdus(psdu, 4)                       # 16-QAM 1/2, 51 symbols
    n = tx.samples.shape[1]
    t = np.arange(n) / n
    drift = (1.0 + 0.45 * np.exp(2j * np.pi * 0.35 * t)).astype(np.complex64)      # flat fading that moves by ~1 rad
    slot_len = ((160 + n + 320 + 63) // 64) * 64
    iq = txgen.impair(tx.samples * drift[None, :], 30.0, cfo=0.0, lead=160, total=slot_len, seed=77).reshape(-1)
    ok = {}
    for ce in (0, 2, 3):
        o, _ = _run(orc, iq, slot_len, tx, ce)
        ok[ce] = int(((o["frames"]["flags"] & orc.F_CRC_OK) != 0).sum())
    assert ok[0] <= 4 and ok[2] == 24 and ok[3] == 24


@pytest.mark.gpu
@pytest.mark.parametrize("chan_est", [2, 3])
@pytest.mark.parametrize("encoding", range(8))
def test_gpu_bit_exact(orc, chan_est, encoding, decode_path):
    from wifirx import capi
    iq, slot_len, tx = make_slots(24, encoding, snr_db=24.0, seed=100 + encoding, taps=TAPS if encoding % 2 else None)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=6, want_carrier=True, chan_est=chan_est)
    r = rx.demod_batch(iq, slot_len, want_csi=True)
    prm = orc.make_params(max_sym=tx.n_sym, llr_bits=6, chan_est=chan_est)
    o = orc.demod_batch(iq, slot_len, prm, want_eq=True, want_csi=True)
    assert np.array_equal(r["frames"], o["frames"]) and np.array_equal(r["idx"], o["idx"])
    assert np.array_equal(r["llr"], o["llr"]) and np.array_equal(r["carrier"], o["eq"])
    assert np.array_equal(r["csi"], o["csi"])
    d = rx.demod_batch(iq, slot_len, decode=True)                      # + decode_mac on the device
    opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=2048)
    assert np.array_equal(d["frames"]["flags"], o["frames"]["flags"])
    for k in np.nonzero(d["frames"]["flags"] & capi.F_DECODED)[0]:
        L = int(d["frames"]["psdu_len"][k])
        assert np.array_equal(d["psdu"][k, :L], opsdu[k, :L])
    rx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("chan_est", [2, 3])
def test_gpu_stream_mode(orc, chan_est):
    from wifirx import capi
    from test_gpu_stream import build_stream
    x, psdus = build_stream(seed=9)
    prm = orc.make_params(max_sym=511, chan_est=chan_est)
    o = orc.demod_stream(x, prm, cap=64)
    rx = capi.WifiRx(max_sym=511, chan_est=chan_est)
    rx.push(x)
    rx._check(capi.lib().wifirx_push(rx._h, None, 0, 0))
    got = rx.poll(cap=64, want_idx=True)
    assert np.array_equal(got["frames"]["flags"] & 0x9f, o["frames"]["flags"] & 0x9f)
    for k in range(len(o["frames"])):
        n = int(o["frames"]["n_sym_out"][k])
        assert np.array_equal(got["idx"][k, :n], o["idx"][k, :n])
    opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=2048)
    ok_g = (got["frames"]["flags"] & capi.F_CRC_OK) != 0
    assert np.array_equal(ok_g, (o["frames"]["flags"] & orc.F_CRC_OK) != 0)
    for k in np.nonzero(ok_g)[0]:
        L = int(got["frames"]["psdu_len"][k])
        assert np.array_equal(got["psdu"][k, :L], opsdu[k, :L])
    rx.close()
