"""IEEE 802.11a/g OFDM frame generator (CPU, NumPy) -- SURVEY.md section 8 row f1.

Functional equivalent of the TX half of the reference's ``wifi_phy_hier`` block plus the
``ieee802_11.mac`` and ``foo.packet_pad2`` blocks in front of / behind it:

* MAC framing ...... gnu_radio/IRS_user.py:192 (``ieee802_11.mac(src, dst, bss)``)
* mapper ........... gnu_radio/wifi_phy_hier.grc:570-586  (scramble, conv-encode, puncture, interleave)
* SIGNAL header .... gnu_radio/wifi_phy_hier.grc:425-441,35-39
* chunks->symbols .. gnu_radio/wifi_phy_hier.grc:316-335,518-532
* carrier allocator  gnu_radio/wifi_phy_hier.grc:336-405 (48 data + 4 pilots, 4 sync words)
* IFFT 1/sqrt(52) .. gnu_radio/wifi_phy_hier.grc:459-479
* cyclic prefixer .. gnu_radio/wifi_phy_hier.grc:406-424 (CP 16, roll-off 2)
* packet_pad2 ...... gnu_radio/IRS_user.py:193 (100 zeros in front, 1000 behind)

It is the synthetic source for tests and bench.py; the TX stays on the CPU exactly as in the
reference (BASELINE.json config 5).  Everything is vectorised over a batch of equal-length frames.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass

import numpy as np

# encoding enum of the reference (gnu_radio/IRS_user.py:130-132)
BPSK_1_2, BPSK_3_4, QPSK_1_2, QPSK_3_4, QAM16_1_2, QAM16_3_4, QAM64_2_3, QAM64_3_4 = range(8)

#            n_bpsc n_cbps n_dbps rate_field puncture
RATE_TABLE = {
    0: (1, 48, 24, 0x0D, "1/2"),
    1: (1, 48, 36, 0x0F, "3/4"),
    2: (2, 96, 48, 0x05, "1/2"),
    3: (2, 96, 72, 0x07, "3/4"),
    4: (4, 192, 96, 0x09, "1/2"),
    5: (4, 192, 144, 0x0B, "3/4"),
    6: (6, 288, 192, 0x01, "2/3"),
    7: (6, 288, 216, 0x03, "3/4"),
}

DATA_BINS = np.array([i for i in range(6, 59) if i not in (11, 25, 32, 39, 53)], dtype=np.int64)
PILOT_BINS = np.array([11, 25, 39, 53], dtype=np.int64)

_LTS_M26_26 = [1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
               1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1]
_STS_M26_26 = [0, 0, 1, 0, 0, 0, -1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 0, 0, 1, 0, 0, 0, 0,
               0, 0, 0, -1, 0, 0, 0, -1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0]


def n_sym_for(psdu_len: int, encoding: int) -> int:
    n_dbps = RATE_TABLE[encoding][2]
    return int(math.ceil((16 + 8 * psdu_len + 6) / float(n_dbps)))


def frame_samples(psdu_len: int, encoding: int) -> int:
    """Samples of one frame out of the cyclic prefixer: (4 sync + SIGNAL + n_sym) * 80 + 1 roll-off."""
    return (5 + n_sym_for(psdu_len, encoding)) * 80 + 1


def polarity_sequence() -> np.ndarray:
    state, out = 0x7F, []
    for _ in range(127):
        fb = ((state >> 6) ^ (state >> 3)) & 1
        out.append(1 - 2 * fb)
        state = ((state << 1) & 0x7E) | fb
    return np.array(out, dtype=np.float64)


def scrambler_sequence(seed: int, n: int) -> np.ndarray:
    """x^7+x^4+1 scrambler output for initial state `seed` (1..127)."""
    state = seed & 0x7F
    period = np.empty(127, dtype=np.uint8)
    for i in range(127):
        fb = ((state >> 6) ^ (state >> 3)) & 1
        period[i] = fb
        state = ((state << 1) & 0x7E) | fb
    reps = (n + 126) // 127
    return np.tile(period, reps)[:n]


def constellation_points(n_bpsc: int) -> np.ndarray:
    """Point for every index; LSB of the index = first transmitted bit."""
    if n_bpsc == 1:
        return np.array([-1.0, 1.0], dtype=np.complex128)
    half = n_bpsc // 2
    if n_bpsc == 2:
        lvl, axis = math.sqrt(0.5), {0: -1, 1: 1}
    elif n_bpsc == 4:
        # axis bits (b0, b1): 00->-3 01->-1 11->+1 10->+3 ; index bit0 = b0, bit1 = b1
        lvl, axis = math.sqrt(0.1), {0b00: -3, 0b10: -1, 0b11: 1, 0b01: 3}
    elif n_bpsc == 6:
        # (b0,b1,b2): 000->-7 001->-5 011->-3 010->-1 110->1 111->3 101->5 100->7 ; bit k of key = b_k
        lvl = math.sqrt(1.0 / 42.0)
        axis = {0b000: -7, 0b100: -5, 0b110: -3, 0b010: -1, 0b011: 1, 0b111: 3, 0b101: 5, 0b001: 7}
    else:
        raise ValueError(n_bpsc)
    pts = np.empty(1 << n_bpsc, dtype=np.complex128)
    mask = (1 << half) - 1
    for idx in range(1 << n_bpsc):
        pts[idx] = lvl * (axis[idx & mask] + 1j * axis[(idx >> half) & mask])
    return pts


def interleaver_map(n_cbps: int, n_bpsc: int) -> np.ndarray:
    """j[k]: the coded bit k of one OFDM symbol is sent at position j[k] (802.11 17.3.5.7)."""
    s = max(n_bpsc // 2, 1)
    k = np.arange(n_cbps)
    i = (n_cbps // 16) * (k % 16) + k // 16
    j = s * (i // s) + (i + n_cbps - (16 * i) // n_cbps) % s
    return j


def conv_encode(bits: np.ndarray) -> np.ndarray:
    """Rate-1/2 K=7 (133,171) encoder, zero initial state. bits: [F, n] -> [F, 2n] (A0 B0 A1 B1 ...)."""
    F, n = bits.shape
    pad = np.zeros((F, n + 6), dtype=np.uint8)
    pad[:, 6:] = bits

    def d(k):  # bits delayed by k
        return pad[:, 6 - k:6 - k + n]
    a = d(0) ^ d(2) ^ d(3) ^ d(5) ^ d(6)
    b = d(0) ^ d(1) ^ d(2) ^ d(3) ^ d(6)
    out = np.empty((F, 2 * n), dtype=np.uint8)
    out[:, 0::2] = a
    out[:, 1::2] = b
    return out


def puncture(coded: np.ndarray, rate: str) -> np.ndarray:
    n = coded.shape[1]
    i = np.arange(n)
    if rate == "1/2":
        return coded
    if rate == "2/3":
        keep = (i % 4) != 3
    elif rate == "3/4":
        keep = ~np.isin(i % 6, (3, 4))
    else:
        raise ValueError(rate)
    return coded[:, keep]


def bytes_to_bits_lsb(data: np.ndarray) -> np.ndarray:
    """[F, n] uint8 -> [F, 8n] bits, LSB of every byte first."""
    return np.unpackbits(data, axis=1, bitorder="little")


def mac_frame(payload: bytes, seq: int = 0,
              src=(0x23,) * 6, dst=(0x42,) * 6, bss=(0xFF,) * 6) -> bytes:
    """PSDU as ``ieee802_11.mac`` builds it: 24-byte data header + payload + CRC-32 (little endian).
    Argument order (src, dst, bss) as at gnu_radio/IRS_user.py:192; addr1=dst, addr2=src, addr3=bss."""
    hdr = bytearray(24)
    hdr[0:2] = (0x0008).to_bytes(2, "little")      # frame control: data frame
    hdr[2:4] = (0).to_bytes(2, "little")           # duration
    hdr[4:10] = bytes(dst)
    hdr[10:16] = bytes(src)
    hdr[16:22] = bytes(bss)
    hdr[22:24] = ((seq & 0xFFF) << 4).to_bytes(2, "little")
    body = bytes(hdr) + bytes(payload)
    return body + (zlib.crc32(body) & 0xFFFFFFFF).to_bytes(4, "little")


def signal_bits(encoding: int, length: int) -> np.ndarray:
    rf = RATE_TABLE[encoding][3]
    b = np.zeros(24, dtype=np.uint8)
    b[0], b[1], b[2], b[3] = (rf >> 3) & 1, (rf >> 2) & 1, (rf >> 1) & 1, rf & 1
    for i in range(12):
        b[5 + i] = (length >> i) & 1
    b[17] = b[:17].sum() & 1
    return b


@dataclass
class TxBatch:
    samples: np.ndarray       # [F, (5+n_sym)*80+1] complex64, unit average power
    data_idx: np.ndarray      # [F, n_sym, 48] uint8 constellation indices as transmitted
    signal_idx: np.ndarray    # [F, 48] uint8 (BPSK) SIGNAL symbol indices
    psdu: np.ndarray          # [F, psdu_len] uint8
    encoding: int
    n_sym: int
    seeds: np.ndarray         # [F] scrambler seeds


def _freq_symbol_to_time(X: np.ndarray) -> np.ndarray:
    """[..., 64] shifted spectrum -> [..., 64] time, un-normalised IDFT with the 1/sqrt(52) window."""
    return np.fft.ifft(np.fft.ifftshift(X, axes=-1), axis=-1) * (64.0 / math.sqrt(52.0))


def sync_words() -> np.ndarray:
    """The 4 sync words of the reference's carrier allocator, shifted order (grc:377-398)."""
    sts = np.zeros(64, dtype=np.complex128)
    lts = np.zeros(64, dtype=np.complex128)
    for k, (s, l) in zip(range(-26, 27), zip(_STS_M26_26, _LTS_M26_26)):
        sts[k + 32] = math.sqrt(13.0 / 6.0) * s * (1 + 1j)
        lts[k + 32] = l
    k = np.arange(-32, 32)
    lts_adv = lts * np.exp(-1j * np.pi * k / 2.0)   # LTS rotated by 16 samples: CP+symbol = GI2+first part
    lts_adv = np.round(lts_adv.real) + 1j * np.round(lts_adv.imag)
    return np.stack([sts, sts, lts_adv, lts])


def encode_psdus(psdu: np.ndarray, encoding: int, seeds=None) -> TxBatch:
    """psdu: [F, L] uint8 (all frames the same length). Returns base-band frames as the reference TX
    would emit them at the hier block's ``samp_out`` port (before the x0.5 gain and packet_pad2)."""
    psdu = np.ascontiguousarray(psdu, dtype=np.uint8)
    F, L = psdu.shape
    n_bpsc, n_cbps, n_dbps, _, rate = RATE_TABLE[encoding]
    n_sym = n_sym_for(L, encoding)
    n_bits = n_sym * n_dbps
    if seeds is None:
        seeds = (np.arange(F) % 127) + 1            # mapper: seed increments per frame, 1..127
    seeds = np.asarray(seeds, dtype=np.int64)

    bits = np.zeros((F, n_bits), dtype=np.uint8)
    bits[:, 16:16 + 8 * L] = bytes_to_bits_lsb(psdu)
    scr = np.empty_like(bits)
    cache = {}
    for f in range(F):
        sd = int(seeds[f])
        if sd not in cache:
            cache[sd] = scrambler_sequence(sd, n_bits)
        scr[f] = cache[sd]
    bits ^= scr
    bits[:, 16 + 8 * L:16 + 8 * L + 6] = 0          # tail bits forced to zero after scrambling
    coded = puncture(conv_encode(bits), rate)        # [F, n_sym*n_cbps]
    assert coded.shape[1] == n_sym * n_cbps
    coded = coded.reshape(F, n_sym, n_cbps)
    j = interleaver_map(n_cbps, n_bpsc)
    inter = np.empty_like(coded)
    inter[:, :, j] = coded
    sym_bits = inter.reshape(F, n_sym, 48, n_bpsc)
    weights = (1 << np.arange(n_bpsc)).astype(np.uint8)
    data_idx = (sym_bits * weights).sum(axis=3).astype(np.uint8)

    # SIGNAL symbol
    sig = np.stack([signal_bits(encoding, L)] * F)
    sig_coded = conv_encode(sig)
    js = interleaver_map(48, 1)
    sig_inter = np.empty_like(sig_coded)
    sig_inter[:, js] = sig_coded
    signal_idx = sig_inter.astype(np.uint8)

    pol = polarity_sequence()
    pts = constellation_points(n_bpsc)
    n_tot = 5 + n_sym
    X = np.zeros((F, n_tot, 64), dtype=np.complex128)
    X[:, 0:4, :] = sync_words()[None]
    X[:, 4, DATA_BINS] = 2.0 * signal_idx - 1.0
    X[:, 5:, :][:, :, DATA_BINS] = pts[data_idx]
    p = pol[np.arange(n_sym + 1) % 127]             # SIGNAL uses p0, data symbol n uses p_{n+1}
    X[:, 4:, 11] = p
    X[:, 4:, 25] = p
    X[:, 4:, 39] = p
    X[:, 4:, 53] = -p
    x = _freq_symbol_to_time(X)                      # [F, n_tot, 64]

    # cyclic prefixer, CP 16, roll-off 2: first CP sample = half own + half previous symbol's continuation
    out = np.zeros((F, n_tot * 80 + 1), dtype=np.complex128)
    sym = np.concatenate([x[:, :, 48:], x], axis=2)  # [F, n_tot, 80]
    sym[:, :, 0] *= 0.5
    sym[:, 1:, 0] += 0.5 * x[:, :-1, 0]
    out[:, :n_tot * 80] = sym.reshape(F, n_tot * 80)
    out[:, n_tot * 80] = 0.5 * x[:, -1, 0]
    return TxBatch(out.astype(np.complex64), data_idx, signal_idx, psdu, encoding, n_sym, seeds)


def make_psdus(n_frames: int, psdu_len: int, seed: int = 2025, seq0: int = 0) -> np.ndarray:
    """Synthetic PSDUs: reference MAC header (src 0x23.., dst 0x42.., bss 0xff..) + PCG64 payload + FCS."""
    rng = np.random.Generator(np.random.PCG64(seed))
    payload_len = psdu_len - 28
    assert payload_len >= 0
    out = np.empty((n_frames, psdu_len), dtype=np.uint8)
    pay = rng.integers(0, 256, size=(n_frames, payload_len), dtype=np.uint8)
    for f in range(n_frames):
        out[f] = np.frombuffer(mac_frame(pay[f].tobytes(), seq=seq0 + f), dtype=np.uint8)
    return out


def packet_pad(frames: np.ndarray, pad_front: int = 100, pad_tail: int = 1000, gain: float = 1.0) -> np.ndarray:
    """foo.packet_pad2 equivalent: [F, n] -> one stream with zeros around every burst."""
    F, n = frames.shape
    out = np.zeros((F, pad_front + n + pad_tail), dtype=np.complex64)
    out[:, pad_front:pad_front + n] = frames * np.float32(gain)
    return out.reshape(-1)


def impair(frames: np.ndarray, snr_db: float | None, cfo: np.ndarray | float = 0.0,
           lead: int = 0, total: int | None = None, seed: int = 1234,
           taps: np.ndarray | None = None) -> np.ndarray:
    """Channel of the reference's loop-back flowgraph (gnu_radio/IRS_tranceiver.py:282-294): the signal
    is scaled by sqrt(10^(snr/10)) against unit-variance complex noise, rotated by `cfo` rad/sample
    (per frame), optionally convolved with per-frame `taps` [F, L]; every frame is placed `lead`
    samples into a slot of `total` samples that is otherwise noise only."""
    F, n = frames.shape
    total = total if total is not None else n + lead
    x = frames.astype(np.complex128)
    if taps is not None:
        taps = np.asarray(taps, dtype=np.complex128)
        if taps.ndim == 1:
            taps = np.broadcast_to(taps, (F, taps.shape[0]))
        y = np.zeros_like(x)
        for l in range(taps.shape[1]):
            y[:, l:] += taps[:, l:l + 1] * x[:, :n - l]
        x = y
    cfo = np.broadcast_to(np.asarray(cfo, dtype=np.float64), (F,))
    idx = np.arange(n)
    x = x * np.exp(1j * cfo[:, None] * idx[None, :])
    slot = np.zeros((F, total), dtype=np.complex128)
    m = max(0, min(n, total - lead))                 # a frame longer than the slot is cut off
    if snr_db is None:
        slot[:, lead:lead + m] = x[:, :m]
        return slot.astype(np.complex64)
    rng = np.random.Generator(np.random.PCG64(seed))
    g = math.sqrt(10.0 ** (snr_db / 10.0))
    noise = (rng.standard_normal((F, total)) + 1j * rng.standard_normal((F, total))) * math.sqrt(0.5)
    slot[:] = noise
    slot[:, lead:lead + m] += g * x[:, :m]
    return slot.astype(np.complex64)
