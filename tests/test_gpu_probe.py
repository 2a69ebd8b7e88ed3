"""f4: the probe_mpsk_snr_est statistics on the device (wifirx_out.sym_stats, wifirx_poll_ex) against the oracle's
spec rule 13 -- exact equality -- and the wifi_phy_rx probe against the SNR the frames were made with."""
import numpy as np
import pytest

from helpers import make_slots
from wifirx import txgen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("encoding,chan_est", [(0, 0), (2, 0), (5, 0), (7, 0), (2, 1), (4, 2), (2, 3)])
def test_batch_sym_stats_equal_the_oracle(orc, encoding, chan_est):
    from wifirx import capi
    iq, slot_len, tx = make_slots(37, encoding, snr_db=21.0, seed=60 + encoding)        # 37: a last wave with one frame
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0, want_carrier=True, chan_est=chan_est)
    r = rx.demod_batch(iq, slot_len, want_stats=True)
    rx.close()
    o = orc.demod_batch(iq, slot_len, orc.make_params(max_sym=tx.n_sym, chan_est=chan_est), want_eq=True)
    assert np.array_equal(r["frames"], o["frames"]) and np.array_equal(r["carrier"], o["eq"])
    ref = orc.sym_stats(o["eq"], o["frames"]["n_sym_out"])
    assert np.array_equal(r["sym_stats"], ref)
    assert (r["sym_stats"][:, 1] > 0).all()
    # and through device buffers + decode (the bench path)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0, chan_est=chan_est)
    r2 = rx.demod_batch(iq, slot_len, decode=True, psdu_stride=320, want_stats=True)
    rx.close()
    assert np.array_equal(r2["sym_stats"], ref)


def test_stream_sym_stats_and_block_probe(orc):
    """stream mode delivers the same moments (wifirx_poll_ex); the block's probe, fed per frame from the device, reports
    the SNR the stream was made with, and its messages come at the reference's cadence (one per 1000 samples)"""
    from wifirx import block, capi, grshim
    snr_db = 17.0
    tx = txgen.encode_psdus(txgen.make_psdus(40, 294, seed=9), 2)
    g = np.float32(np.sqrt(10 ** (snr_db / 10)))
    x = np.concatenate([np.concatenate([np.zeros(100, np.complex64), s * g, np.zeros(600, np.complex64)]) for s in tx.samples])
    rng = np.random.default_rng(4)
    x = (x + (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size)) * np.sqrt(0.5)).astype(np.complex64)
    rx = capi.WifiRx(max_sym=tx.n_sym, want_carrier=True)
    rx.push(x)
    rx._check(capi.lib().wifirx_push(rx._h, None, 0, 0))
    got = rx.poll(cap=64, want_stats=True)
    rx.close()
    assert len(got["frames"]) == 40
    assert np.array_equal(got["sym_stats"], orc.sym_stats(got["carrier"], got["frames"]["n_sym_out"]))
    for typ in (0, 2):
        msgs = []
        blk = block.wifi_phy_rx(bandwidth=20e6, frequency=5.89e9, max_sym=tx.n_sym, publish_carrier=False,
                                snr_probe=(typ, 1000, 0.05))
        grshim.msg_connect(blk.snr_probe, "snr", grshim.sink_block(msgs.append), "in")
        grshim.run_stream(blk, x, chunk=8192)
        # the equalised points carry the channel noise times 1/|H|^2 = 1/gain^2 and the estimation error of H on top
        assert abs(blk.get_probe_snr() - snr_db) < 1.5, (typ, blk.get_probe_snr())
        assert len(msgs) == (40 * 50 * 48 - 1) // 1000
        blk.close()
