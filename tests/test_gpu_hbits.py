"""`wifirx_out.hbits`: the hard decisions as bit planes, the form decode_mac reads (include/wifirx.h).  The demod kernels
write them next to `idx`; wifirx_decode_batch takes them instead of `idx` (no pass over the decisions between the two
kernels).  Pinned here: planes == the bit-plane transposition of `idx` (itself == the oracle's decisions) at every
rate and for every equaliser, host and device buffers, zero where a bin carries no data, nothing written behind a
frame's last symbol; decode_mac from planes alone == the oracle's PSDUs, uniform and mixed rates, full decode waves."""
import numpy as np
import pytest

from helpers import make_slots

pytestmark = pytest.mark.gpu

DATA_BINS = np.array([i for i in range(6, 59) if i not in (11, 25, 32, 39, 53)])       # data carrier c -> FFT bin (shifted)
N_BPSC = np.array([1, 1, 2, 2, 4, 4, 6, 6])


@pytest.fixture(scope="module")
def capi():
    from wifirx import capi
    return capi


def planes_of(frames, idx, max_sym):
    """numpy statement of the layout in include/wifirx.h: word 2 b + h of symbol q = bit b of bins 32 h .. 32 h + 31"""
    n = len(frames)
    out = np.zeros((n, max_sym * 12), np.uint32)
    for f in range(n):
        nb = int(N_BPSC[frames["encoding"][f] & 7]) if frames["n_bpsc"][f] else 0
        for q in range(int(frames["n_sym_out"][f])):
            dec = idx[f, q].astype(np.uint32)
            for b in range(nb):
                bits = np.zeros(64, np.uint64)
                bits[DATA_BINS] = (dec >> b) & 1
                w = int((bits << np.arange(64, dtype=np.uint64)).sum())
                out[f, q * 2 * nb + 2 * b] = w & 0xffffffff
                out[f, q * 2 * nb + 2 * b + 1] = w >> 32
    return out


@pytest.mark.parametrize("encoding", range(8))
def test_planes_equal_transposed_idx_all_rates(capi, orc, encoding):
    iq, slot_len, tx = make_slots(24, encoding, snr_db=12.0 + 2.5 * encoding, seed=900 + encoding)
    ms = tx.n_sym + 1
    rx = capi.WifiRx(max_sym=ms, llr_bits=0)
    r = rx.demod_batch(iq, slot_len, want_hbits=True)                     # host buffers
    o = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=ms))
    assert np.array_equal(r["frames"], o["frames"]) and np.array_equal(r["idx"], o["idx"])
    assert (r["frames"]["n_sym_out"] == tx.n_sym).all()
    want = planes_of(r["frames"], r["idx"], ms)
    assert want.any()
    assert np.array_equal(r["hbits"], want)                                # incl. zeros behind the last symbol
    rx.close()


@pytest.mark.parametrize("chan_est", [1, 2, 3])
def test_planes_other_equalisers(capi, orc, chan_est):
    iq, slot_len, tx = make_slots(16, 5, snr_db=20.0, seed=950 + chan_est)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0, chan_est=chan_est)
    r = rx.demod_batch(iq, slot_len, want_hbits=True)
    o = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, chan_est=chan_est))
    assert np.array_equal(r["idx"], o["idx"])
    assert np.array_equal(r["hbits"], planes_of(r["frames"], r["idx"], tx.n_sym))
    rx.close()


def _mixed_batch(seed=5):
    """frames of all eight rates and several lengths in one batch, slots of one length"""
    from wifirx import txgen
    rng = np.random.default_rng(seed)
    encs = np.array([0, 1, 2, 3, 4, 5, 6, 7] * 6)
    lens = rng.integers(30, 200, encs.size)
    frames, psdus = [], []
    for e, l in zip(encs, lens):
        t = txgen.encode_psdus(txgen.make_psdus(1, int(l), seed=int(rng.integers(1 << 30))), int(e))
        frames.append(t.samples[0])
        psdus.append(t.psdu[0])
    slot_len = max(len(f) for f in frames) + 400
    iq = np.concatenate([txgen.impair(f[None, :], 24.0, cfo=0.001 * (k % 5), lead=160, total=slot_len, seed=seed + k)
                         for k, f in enumerate(frames)], axis=0)
    return iq.reshape(-1), slot_len, encs, lens, psdus


def test_planes_mixed_rates_and_decode_from_planes_only(capi, orc, monkeypatch):
    """one wave of the throughput decoder holds frames of all rates (the per-lane look-up path): planes on the device,
    no `idx` buffer at all"""
    monkeypatch.setenv("WIFIRX_DECODE_SMALL_MAX", "0")
    monkeypatch.setenv("WIFIRX_DECODE_FPW", "128")
    iq, slot_len, encs, lens, psdus = _mixed_batch()
    n = encs.size
    ms = 140
    rx = capi.WifiRx(max_sym=ms, llr_bits=0)
    d_iq = rx.alloc(iq.nbytes).upload(np.ascontiguousarray(iq, np.complex64))
    dev = rx.alloc_out(n, psdu_stride=256, want_hbits=True, want_idx=False)
    rx.demod_batch_dev(d_iq.ptr, slot_len, n, dev)
    rx.decode_batch_dev(n, dev)
    rx.sync()
    r = rx.download_out(dev, n)
    prm = orc.make_params(max_sym=ms)
    o = orc.demod_batch(iq, slot_len, prm)
    opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=256)
    assert np.array_equal(r["frames"], o["frames"])
    assert (r["frames"]["flags"] & orc.F_CRC_OK).all()
    assert np.array_equal(r["hbits"], planes_of(o["frames"], o["idx"], ms))
    for k in range(n):
        assert np.array_equal(r["psdu"][k, :lens[k]], opsdu[k, :lens[k]])
        assert np.array_equal(r["psdu"][k, :lens[k]], psdus[k])
    d_iq.free()
    rx.free_out(dev)
    rx.close()


@pytest.mark.parametrize("small", [True, False])
def test_planes_and_idx_give_the_same_bytes(capi, orc, monkeypatch, small):
    """the same device batch decoded from the planes and from `idx` (the pack pre-pass), both decode kernels"""
    monkeypatch.setenv("WIFIRX_DECODE_SMALL_MAX", "1000000000" if small else "0")
    iq, slot_len, tx = make_slots(70, 7, psdu_len=180, snr_db=19.0, seed=31)      # bit errors: tie-breaks matter
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0)
    d_iq = rx.alloc(iq.nbytes).upload(np.ascontiguousarray(iq, np.complex64))
    dev = rx.alloc_out(70, psdu_stride=192, want_hbits=True)
    rx.demod_batch_dev(d_iq.ptr, slot_len, 70, dev)
    rx.decode_batch_dev(70, dev)
    rx.sync()
    a = rx.download_out(dev, 70)
    planes = dev.pop("hbits")
    dev["hbits"] = None
    dev["psdu"].upload(np.zeros(70 * 192, np.uint8))
    rx.demod_batch_dev(d_iq.ptr, slot_len, 70, dev)
    rx.decode_batch_dev(70, dev)
    rx.sync()
    b = rx.download_out(dev, 70)
    assert np.array_equal(a["frames"], b["frames"]) and np.array_equal(a["psdu"], b["psdu"])
    prm = orc.make_params(max_sym=tx.n_sym)
    o = orc.demod_batch(iq, slot_len, prm)
    opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=192)
    assert np.array_equal(a["frames"], o["frames"])
    dec = (o["frames"]["flags"] & orc.F_DECODED) != 0
    assert dec.any() and np.array_equal(a["psdu"][dec][:, :180], opsdu[dec][:, :180])
    planes.free()
    d_iq.free()
    rx.free_out(dev)
    rx.close()


def test_misaligned_planes_pointer_is_refused(capi):
    rx = capi.WifiRx(max_sym=8, llr_bits=0)
    dev = rx.alloc_out(4, psdu_stride=64, want_hbits=True)
    out = rx._out_struct(dev)
    out.hbits = dev["hbits"].ptr + 4
    iq = rx.alloc(4 * 1024 * 8)
    assert capi.lib().wifirx_demod_batch(rx._h, iq.ptr, 1, 1024, 4, capi.C.byref(out)) == -1
    assert capi.lib().wifirx_decode_batch(rx._h, 4, capi.C.byref(out)) == -1
    iq.free()
    rx.free_out(dev)
    rx.close()


def test_mid_size_batch_partial_decode_waves(capi):
    """90 000 frames: the throughput decoder cuts them into tasks of 88 frames (64 lanes with two frames... 24 of them,
    40 with one) -- planes from the demod kernel, every PSDU back, the last task ragged"""
    from wifirx import txgen
    n, n_t = 90001, 64
    psdu = txgen.make_psdus(n_t, 150, seed=78)
    tx = txgen.encode_psdus(psdu, 4)
    slot = 2048
    assert 160 + tx.samples.shape[1] <= slot
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0)
    slots = rx.alloc(n * slot * 8)
    rx.synth_slots(tx.samples, slots.ptr, slot, n, 160, 26.0, 0.037, 5)
    dev = rx.alloc_out(n, psdu_stride=160, want_hbits=True, want_idx=False)
    rx.demod_batch_dev(slots.ptr, slot, n, dev)
    rx.decode_batch_dev(n, dev)
    rx.sync()
    r = rx.download_out(dev, n)
    ok = (r["frames"]["flags"] & capi.F_CRC_OK) != 0
    assert ok.mean() > 0.999
    assert np.array_equal(r["psdu"][ok][:, :150], psdu[np.arange(n) % n_t][ok])
    assert ok[-1] and ok[-88:].all()
    rx.free_out(dev); slots.free(); rx.close()
