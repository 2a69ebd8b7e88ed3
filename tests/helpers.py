"""Shared helpers of the test-suite: synthetic slots with the reference's loop-back channel."""
import numpy as np

from wifirx import txgen


def make_slots(n_frames, encoding, psdu_len=294, snr_db=25.0, cfo_max=2e-5 * 5.89e9 / 20e6 * 2 * np.pi,
               lead=160, tail=320, seed=7, taps=None, jitter=0):
    """Returns (iq [n_frames*slot_len] complex64, slot_len, TxBatch)."""
    psdu = txgen.make_psdus(n_frames, psdu_len, seed=2025 + seed)
    tx = txgen.encode_psdus(psdu, encoding)
    n = tx.samples.shape[1]
    slot_len = ((lead + n + tail + 63) // 64) * 64
    rng = np.random.default_rng(seed)
    cfo = rng.uniform(-cfo_max, cfo_max, n_frames)
    iq = txgen.impair(tx.samples, snr_db, cfo=cfo, lead=lead, total=slot_len, seed=1234 + seed, taps=taps)
    return iq.reshape(-1), slot_len, tx
