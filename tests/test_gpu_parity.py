"""GPU parity: the HIP chain (through the C ABI) against the oracle's spec mode -- bit for bit."""
import numpy as np
import pytest

from helpers import make_slots

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from wifirx import capi
    return capi


def _compare(r, o, n_bpsc_max):
    fr_g, fr_o = r["frames"], o["frames"]
    for name in fr_g.dtype.names:
        a, b = fr_g[name], fr_o[name]
        assert np.array_equal(a, b), "frame field %s differs: %s vs %s" % (name, a[:8], b[:8])
    assert np.array_equal(r["idx"], o["idx"]), "hard decisions differ"
    if o["llr"] is not None:
        assert np.array_equal(r["llr"].view(np.uint32) & 0x7fffffff | 0, o["llr"].view(np.uint32) & 0x7fffffff | 0) or \
            np.array_equal(r["llr"], o["llr"])
        assert np.array_equal(r["llr"], o["llr"]), "LLRs differ"
    if o["eq"] is not None:
        assert np.array_equal(r["carrier"], o["eq"]), "equalised symbols differ"


@pytest.mark.parametrize("encoding", range(8))
def test_batch_bit_exact_all_rates(capi, orc, encoding):
    iq, slot_len, tx = make_slots(48, encoding, snr_db=22.0, seed=encoding)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=6, want_carrier=True)
    r = rx.demod_batch(iq, slot_len)
    prm = orc.make_params(max_sym=tx.n_sym, llr_bits=6)
    o = orc.demod_batch(iq, slot_len, prm, want_eq=True)
    assert (o["frames"]["flags"] & orc.F_COMPLETE).all()
    _compare(r, o, 6)
    rx.close()


@pytest.mark.parametrize("encoding,snr", [(0, 25), (1, 25), (2, 25), (3, 25), (4, 25), (5, 25), (6, 28), (7, 28),
                                          (2, 4.0), (5, 14.0), (7, 19.0)])
def test_decode_mac_bit_exact(capi, orc, encoding, snr, decode_path):
    """decode_mac (Viterbi, descramble, CRC) on the device == oracle, also where the channel
    leaves bit errors (survivor tie-breaks must agree)."""
    iq, slot_len, tx = make_slots(40, encoding, snr_db=snr, seed=100 + encoding)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0)
    r = rx.demod_batch(iq, slot_len, decode=True, psdu_stride=320)
    prm = orc.make_params(max_sym=tx.n_sym)
    o = orc.demod_batch(iq, slot_len, prm)
    opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=320)
    assert np.array_equal(r["frames"], o["frames"])
    dec = (o["frames"]["flags"] & orc.F_DECODED) != 0
    assert np.array_equal(r["psdu"][dec][:, :294], opsdu[dec][:, :294])
    ok = (o["frames"]["flags"] & orc.F_CRC_OK) != 0
    if snr >= 25:
        assert ok.all()
        assert np.array_equal(r["psdu"][:, :294], tx.psdu)
    rx.close()


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.parametrize("encoding,plen,snr", [(2, 294, 25.0), (2, 294, 5.0), (0, 60, 25.0), (0, 60, 6.0), (7, 294, 19.5), (5, 100, 13.0)])
def test_decode_mac_full_waves(capi, orc, monkeypatch, encoding, plen, snr):
    """The throughput decoder with all 128 frame positions of a wave filled (WIFIRX_DECODE_FPW: as in the 1M-frame
    batches): its uniform traceback (blocks of 96 steps, six decoded bits at a time; trellis lengths with and without a
    remainder mod 96) and, in the last wave, the predicated one -- byte for byte the oracle's PSDUs, also where the
    channel leaves bit errors."""
    monkeypatch.setenv("WIFIRX_DECODE_SMALL_MAX", "0")
    monkeypatch.setenv("WIFIRX_DECODE_FPW", "128")
    n = 300                                                   # two full waves + one with 44 frames
    iq, slot_len, tx = make_slots(n, encoding, psdu_len=plen, snr_db=snr, seed=200 + encoding)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0)
    r = rx.demod_batch(iq, slot_len, decode=True, psdu_stride=320)
    rx.close()
    prm = orc.make_params(max_sym=tx.n_sym)
    o = orc.demod_batch(iq, slot_len, prm, n_threads=8)
    opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=320, n_threads=8)
    assert np.array_equal(r["frames"], o["frames"])
    dec = (o["frames"]["flags"] & orc.F_DECODED) != 0
    assert dec.sum() > 20
    assert np.array_equal(r["psdu"][dec][:, :plen], opsdu[dec][:, :plen])
    if snr >= 25:
        assert ((o["frames"]["flags"] & orc.F_CRC_OK) != 0).all() and np.array_equal(r["psdu"][:, :plen], tx.psdu)


@pytest.mark.parametrize("overlap", [0, 1, 2])
@pytest.mark.parametrize("encoding,plen,snr", [(2, 294, 25.0), (2, 294, 5.0), (2, 294, 2.5), (0, 60, 6.0), (7, 294, 19.5), (5, 100, 13.0), (4, 333, 11.0)])
def test_decode_mac_four_frames_per_lane(capi, orc, monkeypatch, encoding, plen, snr, overlap):
    """decode_q_kernel (four frames per lane, byte path metrics, 256 frames per wave: what batches of a million frames
    take) forced onto a small batch with every frame position of its waves filled: byte for byte the oracle's PSDUs, at
    every constellation incl. the 48-row instance for 64-QAM, also where the channel leaves bit errors and ties."""
    monkeypatch.setenv("WIFIRX_DECODE_SMALL_MAX", "0")
    monkeypatch.setenv("WIFIRX_DECODE_Q", "1")
    monkeypatch.setenv("WIFIRX_DECODE_FPW", "256")
    # overlap = 1: the trace-back of a task runs interleaved with the add-compare-select of the wave's next task; 1400 frames = 6
    # tasks on 3 waves: deferred, interleaved and drained trace-backs all occur.  overlap = 2 (what a million-frame batch takes since
    # round 5): every block of 96 steps is walked back speculatively under the task's own next block, the true trace-back joins the
    # chain of those walks -- at 25 dB after one block, at 2.5 .. 6 dB (start states off the final path, broken links) after many
    monkeypatch.setenv("WIFIRX_DECODE_OVL", str(overlap))
    n = 1400 if overlap else 600                              # full waves + one partly filled
    iq, slot_len, tx = make_slots(n, encoding, psdu_len=plen, snr_db=snr, seed=300 + encoding)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0)
    r = rx.demod_batch(iq, slot_len, decode=True, psdu_stride=384)
    rx.close()
    prm = orc.make_params(max_sym=tx.n_sym)
    o = orc.demod_batch(iq, slot_len, prm, n_threads=8)
    opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=384, n_threads=8)
    assert np.array_equal(r["frames"], o["frames"])
    dec = (o["frames"]["flags"] & orc.F_DECODED) != 0
    assert dec.sum() > 40
    assert np.array_equal(r["psdu"][dec][:, :plen], opsdu[dec][:, :plen])
    if snr >= 25:
        assert ((o["frames"]["flags"] & orc.F_CRC_OK) != 0).all() and np.array_equal(r["psdu"][:, :plen], tx.psdu)


@pytest.mark.parametrize("fail_allocs,budget", [(1, 0), (2, 0), (4, 0), (0, 3_000_000), (0, 700_000)])
def test_decode_mac_scratch_fallbacks(capi, orc, monkeypatch, fail_allocs, budget):
    """the survivor scratch of the throughput decoder is sized from what the device has free and, when its
    allocation fails all the same, retried without the overlap, then with half the waves, and again (the kernels are
    grid-stride over their tasks) -- instead of returning ENOMEM for a batch that fits with fewer waves.  The test hooks fail
    the next k scratch allocations / cap the budget; the PSDUs stay the oracle's, byte for byte."""
    monkeypatch.setenv("WIFIRX_DECODE_SMALL_MAX", "0")
    monkeypatch.setenv("WIFIRX_DECODE_Q", "1")
    monkeypatch.setenv("WIFIRX_DECODE_FPW", "16")            # 1400 frames = 88 tasks: room to halve the waves several times
    monkeypatch.setenv("WIFIRX_DECODE_OVL", "1")
    if fail_allocs:
        monkeypatch.setenv("WIFIRX_TEST_FAIL_DECODE_SCRATCH", str(fail_allocs))
    if budget:
        monkeypatch.setenv("WIFIRX_TEST_DECODE_BUDGET", str(budget))
    n, plen = 1400, 120
    iq, slot_len, tx = make_slots(n, 3, psdu_len=plen, snr_db=9.0, seed=77)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0)
    r = rx.demod_batch(iq, slot_len, decode=True, psdu_stride=128)
    rx.close()
    prm = orc.make_params(max_sym=tx.n_sym)
    o = orc.demod_batch(iq, slot_len, prm, n_threads=8)
    opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=128, n_threads=8)
    assert np.array_equal(r["frames"], o["frames"])
    dec = (o["frames"]["flags"] & orc.F_DECODED) != 0
    assert dec.sum() > 1000
    assert np.array_equal(r["psdu"][dec][:, :plen], opsdu[dec][:, :plen])


def test_decode_mac_scratch_exhausted_is_enomem(capi, monkeypatch):
    """... and when even four waves do not fit, the call says so (WIFIRX_ENOMEM), it does not crash or decode garbage"""
    monkeypatch.setenv("WIFIRX_DECODE_SMALL_MAX", "0")
    monkeypatch.setenv("WIFIRX_DECODE_FPW", "16")
    monkeypatch.setenv("WIFIRX_TEST_FAIL_DECODE_SCRATCH", "50")
    iq, slot_len, tx = make_slots(1400, 2, psdu_len=100, snr_db=25.0, seed=5)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0)
    with pytest.raises(capi.WifiRxError) as e:
        rx.demod_batch(iq, slot_len, decode=True, psdu_stride=128)
    assert e.value.code == -3
    rx.close()
