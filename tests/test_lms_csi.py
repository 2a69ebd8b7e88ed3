"""Row f4: the LMS equaliser (ieee802_11.LMS, gnu_radio/IRS_AP.py:139-141) and the CSI export."""
import numpy as np
import pytest

from helpers import make_slots
from wifirx import txgen


def test_point_of_is_the_transmit_constellation(orc):
    """the decision-directed update divides by the same points the transmitter maps to"""
    import ctypes as C
    # through the LMS path: a clean frame must leave H unchanged up to rounding, i.e. decode perfectly
    for enc in range(8):
        iq, slot_len, tx = make_slots(6, enc, snr_db=40.0, seed=60 + enc)
        prm = orc.make_params(max_sym=tx.n_sym, chan_est=1)
        o = orc.demod_batch(iq, slot_len, prm, want_eq=True)
        assert np.array_equal(o["idx"][:, :tx.n_sym], tx.data_idx)
        pts = txgen.constellation_points(txgen.RATE_TABLE[enc][0])
        err = np.abs(o["eq"][:, :tx.n_sym] - pts[tx.data_idx])
        assert err.max() < 0.08


@pytest.mark.parametrize("encoding", [0, 2, 5, 7])
def test_lms_loopback_both_modes(orc, encoding):
    iq, slot_len, tx = make_slots(10, encoding, snr_db=30.0, seed=70 + encoding)
    out = {}
    for mode in (orc.MATH_SPEC, orc.MATH_LIBM):
        prm = orc.make_params(max_sym=tx.n_sym, math_mode=mode, chan_est=1)
        o = orc.demod_batch(iq, slot_len, prm, want_eq=True, want_csi=True)
        psdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=512)
        assert (o["frames"]["flags"] & orc.F_CRC_OK).all() and np.array_equal(psdu[:, :294], tx.psdu)
        out[mode] = o
    a, b = out[orc.MATH_SPEC], out[orc.MATH_LIBM]
    assert np.array_equal(a["idx"], b["idx"])
    assert np.abs(a["eq"] - b["eq"]).max() / np.abs(b["eq"]).max() < 1e-4
    assert np.abs(a["csi"] - b["csi"]).max() / np.abs(b["csi"]).max() < 1e-4


def test_csi_is_the_channel(orc):
    """flat unit channel: |H| equal on all 52 bins; two-tap channel: H follows the tap spectrum"""
    psdu = txgen.make_psdus(4, 100, seed=3)
    tx = txgen.encode_psdus(psdu, 2)
    taps = np.array([1.0, 0.0, 0.5j], dtype=np.complex64)
    iq = txgen.impair(tx.samples, 35.0, cfo=0.0, lead=160, total=2048, seed=5, taps=taps).reshape(-1)
    o = orc.demod_batch(iq, 2048, orc.make_params(max_sym=tx.n_sym), want_csi=True)
    assert (o["frames"]["flags"] & orc.F_COMPLETE).all()
    k = np.array([i - 32 for i in range(6, 59) if i != 32])
    Hk = 1.0 + 0.5j * np.exp(-2j * np.pi * k * 2 / 64.0)
    csi = o["csi"] / np.abs(o["csi"]).mean(axis=1, keepdims=True) * np.abs(Hk).mean()
    # timing offset of the LTS search leaves a linear phase: compare magnitudes
    assert np.abs(np.abs(csi) - np.abs(Hk)).max() < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("encoding", range(8))
def test_lms_and_csi_bit_exact_on_gpu(orc, encoding):
    from wifirx import capi
    iq, slot_len, tx = make_slots(24, encoding, snr_db=24.0, seed=80 + encoding)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=6, want_carrier=True, chan_est=capi.EQ_LMS)
    r = rx.demod_batch(iq, slot_len, want_csi=True)
    o = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, llr_bits=6, chan_est=1), want_eq=True, want_csi=True)
    assert np.array_equal(r["frames"], o["frames"]) and np.array_equal(r["idx"], o["idx"])
    assert np.array_equal(r["llr"], o["llr"]) and np.array_equal(r["carrier"], o["eq"])
    assert np.array_equal(r["csi"], o["csi"])
    rx.close()
    # and the LS chain delivers the same CSI
    rx = capi.WifiRx(max_sym=tx.n_sym)
    r2 = rx.demod_batch(iq, slot_len, want_csi=True)
    assert np.array_equal(r2["csi"], o["csi"])
    rx.close()


@pytest.mark.gpu
def test_lms_stream_mode(orc):
    """LMS through push/poll (stream kernels are templated on the equaliser too)"""
    from wifirx import capi
    from test_gpu_stream import build_stream
    x, psdus = build_stream(seed=8)
    prm = orc.make_params(max_sym=511, chan_est=1)
    o = orc.demod_stream(x, prm, cap=64)
    rx = capi.WifiRx(max_sym=511, chan_est=capi.EQ_LMS)
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
    assert ok_g.sum() >= len(psdus) - 2          # decision-directed LMS can lose a long 64-QAM frame at 24 dB
    for k in np.nonzero(ok_g)[0]:
        L = int(got["frames"]["psdu_len"][k])
        assert np.array_equal(got["psdu"][k, :L], opsdu[k, :L])
    rx.close()


def test_llr_csi_weight_is_the_preamble_channel_power(orc):
    """WIFIRX_P_LLR_CSI (row a7, optional weight): every LLR times |H|^2 of its sub-carrier, H = the LS estimate the
    CSI export delivers; signs, hence hard decisions, unchanged"""
    taps = np.array([[1.0, 0.4 + 0.3j, 0.2j]], dtype=np.complex64)
    for enc in (0, 2, 5, 7):
        iq, slot_len, tx = make_slots(6, enc, snr_db=28.0, seed=120 + enc, taps=np.repeat(taps, 6, axis=0))
        nb = txgen.RATE_TABLE[enc][0]
        a = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, llr_bits=nb), want_csi=True)
        b = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, llr_bits=nb, llr_csi=1), want_csi=True)
        assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["frames"], b["frames"])
        data_bins = [i for i in range(6, 59) if i not in (11, 25, 32, 39, 53)]
        csi_pos = {i: k for k, i in enumerate(i for i in range(6, 59) if i != 32)}
        w = np.abs(a["csi"][:, [csi_pos[i] for i in data_bins]].astype(np.complex128)) ** 2          # [frames, 48]
        la = a["llr"].reshape(6, tx.n_sym, 48, nb)
        lb = b["llr"].reshape(6, tx.n_sym, 48, nb)
        assert np.allclose(lb, la * w[:, None, :, None], rtol=2e-6, atol=0)
        assert np.array_equal(np.signbit(la), np.signbit(lb))


@pytest.mark.gpu
@pytest.mark.parametrize("chan_est", [0, 3])
@pytest.mark.parametrize("encoding", [0, 3, 4, 7])
def test_llr_csi_bit_exact_on_gpu(orc, encoding, chan_est):
    from wifirx import capi
    iq, slot_len, tx = make_slots(20, encoding, snr_db=22.0, seed=130 + encoding)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=6, chan_est=chan_est)
    rx.set_param(capi.P_LLR_CSI, 1)
    r = rx.demod_batch(iq, slot_len)
    o = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, llr_bits=6, chan_est=chan_est, llr_csi=1))
    assert np.array_equal(r["frames"], o["frames"]) and np.array_equal(r["idx"], o["idx"])
    assert np.array_equal(r["llr"].view(np.uint32), o["llr"].view(np.uint32))
    rx.set_param(capi.P_LLR_CSI, 0)
    r0 = rx.demod_batch(iq, slot_len)
    o0 = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, llr_bits=6, chan_est=chan_est))
    assert np.array_equal(r0["llr"].view(np.uint32), o0["llr"].view(np.uint32))
    assert not np.array_equal(r0["llr"], r["llr"])
    rx.close()
