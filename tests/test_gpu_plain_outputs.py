"""The contract's usual output set -- decisions + LLRs and nothing else -- is what `bench.py` times, and the demod kernel
serves it by code of its own: when every row of a wave with data symbols carries the same constellation and wants exactly
these outputs, the wave runs a data loop compiled for that constellation whose rows leave as whole 16-byte pieces through
LDS (`csrc/wr_quad.h`: `store_bins_lines`; SURVEY 8(a) rows a5-a7, `IRS_AP.py:271` frame_equalizer + decision makers).
Most parity tests also ask for the equalised points (or for no LLRs) and therefore take the general loop; these tests pin
the fast path itself against the oracle, exact equality, for every rate x every equaliser, with and without the bit
planes, with rows that end at different symbols, a last wave that is not full, frames cut short by `max_sym`, and output
buffers whose alignment rules the wide stores out (the same loops then fall back to per-bin stores)."""
import numpy as np
import pytest

from helpers import make_slots
from test_gpu_hbits import planes_of

pytestmark = pytest.mark.gpu
N_BPSC = (1, 1, 2, 2, 4, 4, 6, 6)


@pytest.mark.parametrize("chan_est", [0, 1, 2, 3])
@pytest.mark.parametrize("encoding", range(8))
def test_plain_outputs_every_rate_and_equaliser(orc, encoding, chan_est):
    from wifirx import capi
    n = 70                                             # 17 full waves of four frames + one with two
    iq, slot_len, tx = make_slots(n, encoding, snr_db=14.0 + 2.5 * encoding, seed=400 + 8 * chan_est + encoding, psdu_len=150)
    nb = N_BPSC[encoding]
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=nb, chan_est=chan_est)
    prm = orc.make_params(max_sym=tx.n_sym, llr_bits=nb, chan_est=chan_est)
    o = orc.demod_batch(iq, slot_len, prm)
    assert (o["frames"]["flags"] & orc.F_COMPLETE).sum() >= n - 2
    for hb in (False, True):                           # the kernel instance without / with plane output
        r = rx.demod_batch(iq, slot_len, want_hbits=hb)
        assert np.array_equal(r["frames"], o["frames"])
        assert np.array_equal(r["idx"], o["idx"])
        assert np.array_equal(r["llr"], o["llr"])
        if hb:
            assert np.array_equal(r["hbits"], planes_of(r["frames"], r["idx"], tx.n_sym))
    rx.close()


def _ragged(encoding, lens, n, seed, snr_db=26.0):
    """frames of ONE rate and several lengths, slots of one length: the rows of a wave end at different symbols"""
    from wifirx import txgen
    rng = np.random.default_rng(seed)
    sig, n_sym = [], []
    for k in range(n):
        t = txgen.encode_psdus(txgen.make_psdus(1, int(lens[k % len(lens)]), seed=int(rng.integers(1 << 30))), encoding)
        sig.append(t.samples[0])
        n_sym.append(int(t.n_sym))
    slot_len = ((160 + max(len(s) for s in sig) + 320 + 63) // 64) * 64
    iq = np.zeros((n, slot_len), np.complex64)
    for k, s in enumerate(sig):
        cfo = float(rng.uniform(-0.03, 0.03))
        iq[k] = txgen.impair(s[None, :], snr_db, cfo=np.array([cfo]), lead=160, total=slot_len, seed=int(rng.integers(1 << 30)))[0]
    return iq.reshape(-1), slot_len, np.array(n_sym)


@pytest.mark.parametrize("encoding", [0, 2, 5, 7])
def test_rows_of_a_wave_end_at_different_symbols(orc, encoding):
    from wifirx import capi
    n = 42
    iq, slot_len, n_sym = _ragged(encoding, (30, 260, 120, 333, 77), n, seed=77 + encoding)
    nb, ms = N_BPSC[encoding], int(n_sym.max())
    rx = capi.WifiRx(max_sym=ms, llr_bits=nb)
    r = rx.demod_batch(iq, slot_len, want_hbits=True)
    o = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=ms, llr_bits=nb))
    assert np.array_equal(r["frames"], o["frames"])
    assert np.array_equal(r["frames"]["n_sym_out"], n_sym)                # every frame complete, each at its own length
    assert np.array_equal(r["idx"], o["idx"]) and np.array_equal(r["llr"], o["llr"])      # incl. the zeros behind a frame's end
    assert np.array_equal(r["hbits"], planes_of(r["frames"], r["idx"], ms))
    rx.close()


@pytest.mark.parametrize("encoding", [1, 3, 4, 6])
def test_frames_cut_short_by_max_sym(orc, encoding):
    from wifirx import capi
    n = 22
    iq, slot_len, tx = make_slots(n, encoding, snr_db=28.0, seed=610 + encoding, psdu_len=200)
    nb, ms = N_BPSC[encoding], tx.n_sym - 3
    rx = capi.WifiRx(max_sym=ms, llr_bits=nb)
    r = rx.demod_batch(iq, slot_len)
    o = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=ms, llr_bits=nb))
    assert np.array_equal(r["frames"], o["frames"])
    assert (r["frames"]["flags"] & 0x80).all()            # WIFIRX_F_TRUNCATED
    assert (r["frames"]["n_sym_out"] == ms).all()
    assert np.array_equal(r["idx"], o["idx"]) and np.array_equal(r["llr"], o["llr"])
    rx.close()


@pytest.mark.parametrize("shift", [(1, 4), (4, 16), (16, 4)])
@pytest.mark.parametrize("encoding", [0, 2, 4, 7])
def test_plain_outputs_into_buffers_of_any_alignment(orc, encoding, shift):
    """decisions / LLRs `shift` bytes behind a 256-byte boundary: (1, 4) and (16, 4) rule the 16-byte pieces out (LLRs on a
    4-byte boundary only), (4, 16) is the least the line stores need -- the values must not depend on it"""
    from wifirx import capi
    C = capi.C
    n = 38
    iq, slot_len, tx = make_slots(n, encoding, snr_db=27.0, seed=31 + encoding, psdu_len=120)
    nb = N_BPSC[encoding]
    per = tx.n_sym * 48
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=nb)
    d_iq = rx.alloc(iq.nbytes).upload(iq)
    d_fr, d_idx, d_llr = rx.alloc(n * 32), rx.alloc(n * per + 64), rx.alloc(n * per * nb * 4 + 64)
    for b in (d_fr, d_idx, d_llr):
        b.upload(np.zeros(b.nbytes, np.uint8))
    si, sl = shift
    out = capi.Out(d_fr.ptr, d_idx.ptr + si, d_llr.ptr + sl, None, None, 0, 1, None)
    rx._check(capi.lib().wifirx_demod_batch(rx._h, d_iq.ptr, 1, slot_len, n, C.byref(out)))
    rx.sync()
    fr = d_fr.download(capi.FRAME_DTYPE, n)
    idx_all = d_idx.download(np.uint8, n * per + 64)
    llr_all = d_llr.download(np.float32, n * per * nb + 16)
    o = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, llr_bits=nb))
    assert np.array_equal(fr, o["frames"]) and (fr["flags"] & capi.F_COMPLETE).all()
    assert np.array_equal(idx_all[si:si + n * per], o["idx"].reshape(-1))
    assert np.array_equal(llr_all[sl // 4:sl // 4 + n * per * nb], o["llr"].reshape(-1))
    # nothing in front of or behind the rows
    assert not idx_all[:si].any() and not idx_all[si + n * per:].any()
    assert not llr_all[:sl // 4].any() and not llr_all[sl // 4 + n * per * nb:].any()
    for b in (d_iq, d_fr, d_idx, d_llr):
        b.free()
    rx.close()
