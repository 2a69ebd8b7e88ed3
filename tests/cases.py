"""Edge-case batches shared by the oracle tests (CPU) and the GPU parity tests."""
import os

import numpy as np

from wifirx import txgen

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _noise(shape, rng, sigma=1.0):
    return ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * (sigma * np.sqrt(0.5))).astype(np.complex64)


def edge_cases():
    """yields (name, iq, slot_len, max_sym, expectations dict)"""
    rng = np.random.default_rng(11)
    # noise only / zeros only: nothing may be delivered
    yield "noise_only", _noise(8 * 2048, rng), 2048, 16, dict(complete=0)
    yield "zeros_only", np.zeros(4 * 1024, np.complex64), 1024, 16, dict(complete=0, detected=0)
    # ragged slot length (not a multiple of 16 or 64), frame right at the slot start, minimal tail
    psdu = txgen.make_psdus(6, 60, seed=1)
    tx = txgen.encode_psdus(psdu, 3)
    n = tx.samples.shape[1]
    iq = txgen.impair(tx.samples, 25.0, cfo=0.01, lead=0, total=n + 7, seed=3)
    yield "ragged_lead0", iq.reshape(-1), n + 7, tx.n_sym, dict(complete=6)
    iq = txgen.impair(tx.samples, 25.0, cfo=-0.02, lead=37, total=n + 37 + 3, seed=4)
    yield "ragged_lead37", iq.reshape(-1), n + 40, tx.n_sym, dict(complete=6)
    # truncated by the slot end: 3 data symbols missing
    iq = txgen.impair(tx.samples, 25.0, cfo=0.0, lead=100, total=100 + n - 240, seed=5)
    yield "truncated_slot", iq.reshape(-1), 100 + n - 240, tx.n_sym, dict(complete=0, truncated=6)
    # output capacity smaller than the frame
    iq = txgen.impair(tx.samples, 25.0, cfo=0.0, lead=100, total=100 + n + 100, seed=6)
    yield "capacity_limit", iq.reshape(-1), n + 200, tx.n_sym - 2, dict(complete=0, truncated=6)
    # slot shorter than the LTS search window
    yield "tiny_slots", iq.reshape(-1)[:6 * 300], 300, 8, dict(complete=0)
    # largest frame decode_mac accepts: PSDU 1528 at BPSK 1/2 -> 511 symbols
    psdu = txgen.make_psdus(2, 1528, seed=2)
    txl = txgen.encode_psdus(psdu, 0)
    assert txl.n_sym == 511
    nl = txl.samples.shape[1]
    # (small CFO: the equalizer's sampling-offset compensation assumes a sampling clock locked to the
    #  carrier, App. A.5 step 1; impair() offsets the carrier only, which over 511 symbols matters)
    iq = txgen.impair(txl.samples, 22.0, cfo=[0.004, -0.004], lead=160, total=160 + nl + 64, seed=7)
    yield "max_frame_511_symbols", iq.reshape(-1), 160 + nl + 64, 511, dict(complete=2, crc_ok=2, psdu=psdu)
    # PSDUs too short to hold an FCS, and LENGTH 0
    for plen in (0, 1, 3):
        p = np.zeros((3, plen), np.uint8) + 0x5A
        t = txgen.encode_psdus(p, 2)
        iq = txgen.impair(t.samples, 25.0, cfo=0.005, lead=120, total=1024, seed=8 + plen)
        yield "short_psdu_%d" % plen, iq.reshape(-1), 1024, 4, dict(complete=3, crc_ok=0)
    # 64-QAM 3/4 through the SV-derived multipath taps (BASELINE.json config 3)
    taps = np.load(os.path.join(GOLD, "sv_taps.npy"))[:24]
    psdu = txgen.make_psdus(24, 294, seed=3)
    t7 = txgen.encode_psdus(psdu, 7)
    iq = txgen.impair(t7.samples, 30.0, cfo=rng.uniform(-0.03, 0.03, 24), lead=160, total=1472, seed=12, taps=taps)
    yield "sv_multipath_64qam", iq.reshape(-1), 1472, t7.n_sym, dict(min_complete=20)
    # strong second burst inside the slot is ignored in batch mode (first frame only)
    two = np.concatenate([iq.reshape(24, 1472)[:4], iq.reshape(24, 1472)[4:8]], axis=1)
    yield "two_frames_per_slot", two.reshape(-1), 2944, t7.n_sym, dict(min_complete=3)
    # low SNR: many SIGNAL / CRC failures, whatever they are the two sides must agree
    psdu = txgen.make_psdus(32, 100, seed=4)
    t4 = txgen.encode_psdus(psdu, 4)
    iq = txgen.impair(t4.samples, 6.0, cfo=rng.uniform(-0.03, 0.03, 32), lead=160, total=1600, seed=13)
    yield "low_snr_16qam", iq.reshape(-1), 1600, t4.n_sym, dict()
