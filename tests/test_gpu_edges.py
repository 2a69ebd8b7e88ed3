"""GPU vs oracle on the edge-case batches (bit for bit, through the C ABI)."""
import numpy as np
import pytest

from cases import edge_cases

pytestmark = pytest.mark.gpu
CASES = {c[0]: c for c in edge_cases()}


@pytest.mark.parametrize("name", sorted(CASES))
def test_edge_case_bit_exact(orc, name, decode_path):
    from wifirx import capi
    _, iq, slot_len, max_sym, exp = CASES[name]
    rx = capi.WifiRx(max_sym=max_sym, llr_bits=6, want_carrier=True)
    r = rx.demod_batch(iq, slot_len, decode=True, psdu_stride=2048)
    prm = orc.make_params(max_sym=max_sym, llr_bits=6)
    o = orc.demod_batch(iq, slot_len, prm, want_eq=True)
    opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=2048)
    assert np.array_equal(r["frames"], o["frames"]), (r["frames"], o["frames"])
    assert np.array_equal(r["idx"], o["idx"])
    assert np.array_equal(r["llr"], o["llr"])
    assert np.array_equal(r["carrier"], o["eq"])
    dec = (o["frames"]["flags"] & orc.F_DECODED) != 0
    for k in np.nonzero(dec)[0]:
        L = int(o["frames"]["psdu_len"][k])
        assert np.array_equal(r["psdu"][k, :L], opsdu[k, :L])
    rx.close()


def test_empty_batch_and_argument_errors():
    from wifirx import capi
    rx = capi.WifiRx(max_sym=8)
    r = rx.demod_batch(np.zeros(0, np.complex64), 1024)
    assert r["frames"].shape == (0,)
    with pytest.raises(capi.WifiRxError):
        rx.set_param(99, 1.0)
    with pytest.raises(capi.WifiRxError):
        rx.set_param(capi.P_CHAN_EST, 4)
    for eq in (capi.EQ_STA, capi.EQ_COMB, capi.EQ_LMS, capi.EQ_LS):
        rx.set_param(capi.P_CHAN_EST, eq)
    st = rx.stats()
    assert st["samples_in"] == 0
    rx.close()


def test_full_size_properties():
    """BASELINE config-2 shape at reduced frame count but full slot geometry, checked through
    size-independent properties: every frame complete + CRC ok, decoded PSDU == transmitted PSDU of its
    template, hard decisions == transmitted symbols except where the channel flipped them (BER sane),
    and the result is independent of how the batch is split into calls."""
    from wifirx import capi, txgen
    n, n_t = 20000, 64
    psdu = txgen.make_psdus(n_t, 294, seed=77)
    tx = txgen.encode_psdus(psdu, 2)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=2)
    slots = rx.alloc(n * 4608 * 8)
    rx.synth_slots(tx.samples, slots.ptr, 4608, n, 160, 20.0, 0.037, 99)
    dev = rx.alloc_out(n, psdu_stride=320)
    rx.demod_batch_dev(slots.ptr, 4608, n, dev)
    rx.decode_batch_dev(n, dev)
    rx.sync()
    r = rx.download_out(dev, n)
    fl = r["frames"]["flags"]
    assert ((fl & capi.F_COMPLETE) != 0).all() and ((fl & capi.F_CRC_OK) != 0).mean() > 0.999
    good = (fl & capi.F_CRC_OK) != 0
    assert np.array_equal(r["psdu"][good][:, :294], psdu[np.arange(n) % n_t][good])
    ser = (r["idx"] != tx.data_idx[np.arange(n) % n_t]).mean()
    assert ser < 2e-3
    # split invariance: second half as its own call gives the same bytes
    dev2 = rx.alloc_out(n // 2, psdu_stride=320)
    rx.demod_batch_dev(slots.ptr + (n // 2) * 4608 * 8, 4608, n // 2, dev2)
    rx.sync()
    r2 = rx.download_out(dev2, n // 2)
    assert np.array_equal(r2["idx"], r["idx"][n // 2:]) and np.array_equal(r2["llr"], r["llr"][n // 2:])
    rx.free_out(dev); rx.free_out(dev2); slots.free(); rx.close()


def test_decode_with_unaligned_buffers(orc, decode_path):
    """decode_mac's fast paths (16-byte tile copies of the decisions, dword PSDU stores) are taken only when the
    caller's buffers allow it: an odd PSDU stride and a decisions buffer that is merely 4-byte aligned must give
    the same bytes"""
    from helpers import make_slots
    from wifirx import capi
    C = capi.C
    iq, slot_len, tx = make_slots(150, 5, snr_db=26.0, seed=21, psdu_len=333)
    n = 150
    rx = capi.WifiRx(max_sym=tx.n_sym)
    d_iq = rx.alloc(iq.nbytes).upload(iq)
    d_fr, d_idx = rx.alloc(n * 32), rx.alloc(n * tx.n_sym * 48 + 64)
    stride = 337
    d_psdu = rx.alloc(n * stride + 64)
    for b in (d_fr, d_idx, d_psdu):
        b.upload(np.zeros(b.nbytes, np.uint8))
    out = capi.Out(d_fr.ptr, d_idx.ptr + 4, None, None, d_psdu.ptr + 1, stride, 1, None)
    rx._check(capi.lib().wifirx_demod_batch(rx._h, d_iq.ptr, 1, slot_len, n, C.byref(out)))
    rx._check(capi.lib().wifirx_decode_batch(rx._h, n, C.byref(out)))
    rx.sync()
    fr = d_fr.download(capi.FRAME_DTYPE, n)
    psdu = d_psdu.download(np.uint8, n * stride + 1)[1:].reshape(n, stride)
    prm = orc.make_params(max_sym=tx.n_sym)
    o = orc.demod_batch(iq, slot_len, prm)
    opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=stride)
    assert np.array_equal(fr, o["frames"])
    assert (fr["flags"] & capi.F_CRC_OK).all()
    assert np.array_equal(psdu[:, :333], opsdu[:, :333]) and np.array_equal(psdu[:, :333], tx.psdu)
    for b in (d_iq, d_fr, d_idx, d_psdu):
        b.free()
    rx.close()


def test_slots_of_unequal_length(orc):
    """wifirx_demod_batch_v (the slot_off[] form of SURVEY.md 8(b)): recordings of different lengths back to back, one
    of them empty, one too short for a frame -- every slot equals the oracle's result for that slot alone."""
    from wifirx import capi, txgen
    rng = np.random.default_rng(3)
    parts, refs = [], []
    encs = [0, 2, 5, 7, 3, 6, 2, 4, 1]
    for k, enc in enumerate(encs):
        plen = int(rng.integers(40, 200))
        tx = txgen.encode_psdus(txgen.make_psdus(1, plen, seed=40 + k), enc)
        total = int(rng.integers(tx.samples.shape[1] + 100, tx.samples.shape[1] + 900))
        parts.append(txgen.impair(tx.samples, 25.0, cfo=float(rng.uniform(-0.03, 0.03)), lead=int(rng.integers(20, 90)),
                                  total=total, seed=k)[0])
    parts.insert(3, np.zeros(0, np.complex64))                       # an empty slot
    parts.insert(6, parts[0][:250].copy())                           # a slot shorter than the LTS search needs
    off = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.uint64)
    iq = np.concatenate(parts)
    max_sym = 140
    rx = capi.WifiRx(max_sym=max_sym, llr_bits=6, want_carrier=True)
    r = rx.demod_batch_var(iq, off)
    rx.close()
    prm = orc.make_params(max_sym=max_sym, llr_bits=6)
    n_ok = 0
    for k, p in enumerate(parts):
        if p.size == 0:
            assert r["frames"]["flags"][k] == 0 and r["frames"]["trigger"][k] == -1
            continue
        o = orc.demod_batch(p, p.size, prm, want_eq=True)
        assert np.array_equal(r["frames"][k:k + 1], o["frames"]), k
        assert np.array_equal(r["idx"][k], o["idx"][0]) and np.array_equal(r["llr"][k], o["llr"][0]), k
        assert np.array_equal(r["carrier"][k], o["eq"][0]), k
        n_ok += int((o["frames"]["flags"][0] & orc.F_COMPLETE) != 0)
    assert n_ok == len(encs)


@pytest.mark.parametrize("encoding", [0, 2, 4, 7])
def test_demod_with_minimally_aligned_output_buffers(orc, encoding):
    """The C ABI promises nothing about the callers' output buffers beyond the alignment of their element types: LLRs on
    a 4-byte, equalised points on an 8-byte, decisions on a 1-byte boundary must give the same values as aligned
    buffers (the kernel's wide stores -- 16 bytes per 16-QAM carrier -- then simply split)."""
    from helpers import make_slots
    from wifirx import capi
    C = capi.C
    n = 96
    iq, slot_len, tx = make_slots(n, encoding, snr_db=27.0, seed=31 + encoding, psdu_len=120)
    n_bpsc = (1, 1, 2, 2, 4, 4, 6, 6)[encoding]
    per = tx.n_sym * 48
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=n_bpsc, want_carrier=True)
    d_iq = rx.alloc(iq.nbytes).upload(iq)
    d_fr, d_idx = rx.alloc(n * 32), rx.alloc(n * per + 64)
    d_llr, d_car = rx.alloc(n * per * n_bpsc * 4 + 64), rx.alloc(n * per * 8 + 64)
    for b in (d_fr, d_idx, d_llr, d_car):
        b.upload(np.zeros(b.nbytes, np.uint8))
    out = capi.Out(d_fr.ptr, d_idx.ptr + 3, d_llr.ptr + 4, d_car.ptr + 8, None, 0, 1, None)
    rx._check(capi.lib().wifirx_demod_batch(rx._h, d_iq.ptr, 1, slot_len, n, C.byref(out)))
    rx.sync()
    fr = d_fr.download(capi.FRAME_DTYPE, n)
    idx = d_idx.download(np.uint8, n * per + 3)[3:].reshape(n, per)
    llr = d_llr.download(np.float32, n * per * n_bpsc + 1)[1:].reshape(n, per * n_bpsc)
    car = d_car.download(np.complex64, n * per + 1)[1:].reshape(n, per)
    prm = orc.make_params(max_sym=tx.n_sym, llr_bits=n_bpsc)
    o = orc.demod_batch(iq, slot_len, prm, want_eq=True)
    assert np.array_equal(fr, o["frames"]) and (fr["flags"] & capi.F_COMPLETE).all()
    assert np.array_equal(idx, o["idx"].reshape(n, per))
    assert np.array_equal(llr, o["llr"].reshape(n, per * n_bpsc))
    assert np.array_equal(car, o["eq"].reshape(n, per))
    for b in (d_iq, d_fr, d_idx, d_llr, d_car):
        b.free()
    rx.close()
