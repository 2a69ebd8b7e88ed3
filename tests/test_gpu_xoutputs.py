"""The other output sets through the constellation loops (round 4): the reference's own set --
the equalised points of every data symbol on the `carrier` port (`gnu_radio/IRS_AP.py:293,312-313`:
frame_equalizer.symbols -> pdu_to_tagged_stream -> probe_mpsk_snr_est) --, LLRs weighted by the channel state
(WIFIRX_P_LLR_CSI, spec rule 12), the probe's moments (`wifirx_out.sym_stats`) and the bit planes alone (what the stream
path asks for).  Until round 4 any of these sent a wave to the general loop with per-bin stores; now a wave whose rows
share a constellation runs a loop compiled for it whose rows leave as whole 16-byte pieces (`csrc/wr_quad.h`:
`store_rows_x`).  Exact equality with the oracle for every rate x every equaliser x every output set, with ragged rows,
partial last waves, and buffers whose alignment rules the wide stores out.

PARITY UNPINNED: "the oracle" is this repository's CPU restatement."""
import numpy as np
import pytest

from helpers import make_slots
from test_gpu_hbits import planes_of
from test_gpu_plain_outputs import _ragged

pytestmark = pytest.mark.gpu
N_BPSC = (1, 1, 2, 2, 4, 4, 6, 6)


@pytest.mark.parametrize("chan_est", [0, 1, 2, 3])
@pytest.mark.parametrize("encoding", range(8))
def test_carrier_stats_weights_every_rate_and_equaliser(orc, encoding, chan_est):
    from wifirx import capi
    n = 70                                             # 17 full waves of four frames + one with two
    iq, slot_len, tx = make_slots(n, encoding, snr_db=14.0 + 2.5 * encoding, seed=900 + 8 * chan_est + encoding, psdu_len=150)
    nb = N_BPSC[encoding]
    # (a) the reference's own set + LLRs: decisions, LLRs, equalised points -- without and with the planes, and with the moments
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=nb, chan_est=chan_est, want_carrier=True)
    o = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, llr_bits=nb, chan_est=chan_est), want_eq=True)
    assert (o["frames"]["flags"] & orc.F_COMPLETE).sum() >= n - 2
    stats = orc.sym_stats(o["eq"], o["frames"]["n_sym_out"])
    for hb, st in ((False, False), (True, False), (False, True), (True, True)):
        r = rx.demod_batch(iq, slot_len, want_hbits=hb, want_stats=st)
        assert np.array_equal(r["frames"], o["frames"])
        assert np.array_equal(r["idx"], o["idx"])
        assert np.array_equal(r["llr"], o["llr"])
        assert np.array_equal(r["carrier"].view(np.uint32), o["eq"].view(np.uint32))
        if hb:
            assert np.array_equal(r["hbits"], planes_of(r["frames"], r["idx"], tx.n_sym))
        if st:
            assert np.array_equal(r["sym_stats"], stats)
    # (b) weighted LLRs, with the points
    rx.set_param(capi.P_LLR_CSI, 1)
    ow = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, llr_bits=nb, chan_est=chan_est, llr_csi=1), want_eq=True)
    r = rx.demod_batch(iq, slot_len)
    assert np.array_equal(r["frames"], ow["frames"]) and np.array_equal(r["idx"], ow["idx"])
    assert np.array_equal(r["llr"].view(np.uint32), ow["llr"].view(np.uint32))
    assert np.array_equal(r["carrier"].view(np.uint32), ow["eq"].view(np.uint32))
    rx.close()
    # (c) weighted LLRs without the points; (d) the reference's set as it is: decisions + points, no LLRs
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=nb, chan_est=chan_est)
    rx.set_param(capi.P_LLR_CSI, 1)
    r = rx.demod_batch(iq, slot_len, want_stats=True)
    assert np.array_equal(r["frames"], ow["frames"]) and np.array_equal(r["idx"], ow["idx"])
    assert np.array_equal(r["llr"].view(np.uint32), ow["llr"].view(np.uint32))
    assert np.array_equal(r["sym_stats"], stats)
    rx.close()
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0, chan_est=chan_est, want_carrier=True)
    o0 = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, llr_bits=0, chan_est=chan_est), want_eq=True)
    for hb in (False, True):
        r = rx.demod_batch(iq, slot_len, want_hbits=hb)
        assert np.array_equal(r["frames"], o0["frames"]) and np.array_equal(r["idx"], o0["idx"])
        assert np.array_equal(r["carrier"].view(np.uint32), o0["eq"].view(np.uint32))
        if hb:
            assert np.array_equal(r["hbits"], planes_of(r["frames"], r["idx"], tx.n_sym))
    rx.close()


@pytest.mark.parametrize("chan_est", [0, 3])
@pytest.mark.parametrize("encoding", [0, 3, 4, 7])
def test_planes_alone_and_points_alone(orc, encoding, chan_est):
    """device buffers handed over one by one: the planes and records alone (the stream path's request), the planes + the
    points, the points alone"""
    from wifirx import capi
    C = capi.C
    n = 46
    iq, slot_len, tx = make_slots(n, encoding, snr_db=24.0, seed=77 + encoding, psdu_len=180)
    nb, per = N_BPSC[encoding], tx.n_sym * 48
    o = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, llr_bits=0, chan_est=chan_est), want_eq=True)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0, chan_est=chan_est, want_carrier=True)
    d_iq = rx.alloc(iq.nbytes).upload(iq)
    d_fr, d_hb, d_car = rx.alloc(n * 32), rx.alloc(n * per), rx.alloc(n * per * 8)
    for want_hb, want_car in ((True, False), (True, True), (False, True)):
        for b in (d_fr, d_hb, d_car):
            b.upload(np.zeros(b.nbytes, np.uint8))
        out = capi.Out(d_fr.ptr, None, None, d_car.ptr if want_car else None, None, 0, 1, None, None, d_hb.ptr if want_hb else None)
        rx._check(capi.lib().wifirx_demod_batch(rx._h, d_iq.ptr, 1, slot_len, n, C.byref(out)))
        rx.sync()
        fr = d_fr.download(capi.FRAME_DTYPE, n)
        assert np.array_equal(fr, o["frames"])
        hb = d_hb.download(np.uint32, n * tx.n_sym * 12).reshape(n, -1)
        car = d_car.download(np.complex64, n * per).reshape(n, tx.n_sym, 48)
        if want_hb:
            assert np.array_equal(hb, planes_of(fr, o["idx"], tx.n_sym))
        else:
            assert not hb.any()
        if want_car:
            assert np.array_equal(car.view(np.uint32), o["eq"].view(np.uint32))
        else:
            assert not car.view(np.uint32).any()
    for b in (d_iq, d_fr, d_hb, d_car):
        b.free()
    rx.close()


@pytest.mark.parametrize("encoding", [0, 2, 5, 7])
def test_ragged_rows_with_points_and_moments(orc, encoding):
    from wifirx import capi
    n = 42
    iq, slot_len, n_sym = _ragged(encoding, (30, 260, 120, 333, 77), n, seed=177 + encoding)
    nb, ms = N_BPSC[encoding], int(n_sym.max())
    rx = capi.WifiRx(max_sym=ms, llr_bits=nb, want_carrier=True)
    r = rx.demod_batch(iq, slot_len, want_hbits=True, want_stats=True)
    o = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=ms, llr_bits=nb), want_eq=True)
    assert np.array_equal(r["frames"], o["frames"])
    assert np.array_equal(r["frames"]["n_sym_out"], n_sym)
    assert np.array_equal(r["idx"], o["idx"]) and np.array_equal(r["llr"], o["llr"])      # incl. the zeros behind a frame's end
    assert np.array_equal(r["carrier"].view(np.uint32), o["eq"].view(np.uint32))
    assert np.array_equal(r["hbits"], planes_of(r["frames"], r["idx"], ms))
    assert np.array_equal(r["sym_stats"], orc.sym_stats(o["eq"], o["frames"]["n_sym_out"]))
    rx.close()


@pytest.mark.parametrize("shift", [(1, 4, 8), (4, 16, 16), (4, 16, 8), (16, 4, 16)])
@pytest.mark.parametrize("encoding", [0, 2, 4, 7])
def test_points_into_buffers_of_any_alignment(orc, encoding, shift):
    """decisions / LLRs / points `shift` bytes behind a 256-byte boundary: only (4, 16, 16) allows the 16-byte pieces; the
    values must not depend on it, and nothing may be written in front of or behind the rows"""
    from wifirx import capi
    C = capi.C
    n = 38
    iq, slot_len, tx = make_slots(n, encoding, snr_db=27.0, seed=131 + encoding, psdu_len=120)
    nb = N_BPSC[encoding]
    per = tx.n_sym * 48
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=nb, want_carrier=True)
    d_iq = rx.alloc(iq.nbytes).upload(iq)
    d_fr, d_idx, d_llr, d_car = rx.alloc(n * 32), rx.alloc(n * per + 64), rx.alloc(n * per * nb * 4 + 64), rx.alloc(n * per * 8 + 64)
    for b in (d_fr, d_idx, d_llr, d_car):
        b.upload(np.zeros(b.nbytes, np.uint8))
    si, sl, sc = shift
    out = capi.Out(d_fr.ptr, d_idx.ptr + si, d_llr.ptr + sl, d_car.ptr + sc, None, 0, 1, None)
    rx._check(capi.lib().wifirx_demod_batch(rx._h, d_iq.ptr, 1, slot_len, n, C.byref(out)))
    rx.sync()
    fr = d_fr.download(capi.FRAME_DTYPE, n)
    idx_all = d_idx.download(np.uint8, n * per + 64)
    llr_all = d_llr.download(np.uint32, n * per * nb + 16)
    car_all = d_car.download(np.uint32, n * per * 2 + 16)
    o = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, llr_bits=nb), want_eq=True)
    assert np.array_equal(fr, o["frames"]) and (fr["flags"] & capi.F_COMPLETE).all()
    assert np.array_equal(idx_all[si:si + n * per], o["idx"].reshape(-1))
    assert np.array_equal(llr_all[sl // 4:sl // 4 + n * per * nb], o["llr"].reshape(-1).view(np.uint32))
    assert np.array_equal(car_all[sc // 4:sc // 4 + n * per * 2], o["eq"].reshape(-1).view(np.uint32))
    assert not idx_all[:si].any() and not idx_all[si + n * per:].any()
    assert not llr_all[:sl // 4].any() and not llr_all[sl // 4 + n * per * nb:].any()
    assert not car_all[:sc // 4].any() and not car_all[sc // 4 + n * per * 2:].any()
    for b in (d_iq, d_fr, d_idx, d_llr, d_car):
        b.free()
    rx.close()
