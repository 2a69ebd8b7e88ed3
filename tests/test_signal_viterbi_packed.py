"""The SIGNAL decoder of the demod kernels (csrc/wr_quad.h viterbi_signal4) keeps the path metrics of the four frames of a
wave in the four bytes of one register: start value 64 instead of the oracle's 2^28, compare by a guard-bit subtraction
((m0 + 0x7f7f7f7f) - m1: top bit of byte f = m1 < m0), survivor select by the resulting byte mask, final state by a
minimum over (metric << 6 | state).  Pinned here on the CPU, step by step in numpy with exactly those word operations:
the decoded bits equal the oracle's Viterbi (orc_viterbi, oracle/wifirx_oracle.c viterbi_decode) on random coded
sequences -- clean, noisy and pure noise --, and no byte ever reaches 128.  (The kernel itself is compared with the
oracle on every frame of the -m gpu suite; this test states WHY the packed form may replace the 32-bit one.)"""
import numpy as np


def parity(x):
    return bin(x).count("1") & 1


S = np.arange(64)
P0, P1 = S >> 1, (S >> 1) | 32
F0, F1 = (P0 << 1) | (S & 1), (P1 << 1) | (S & 1)
A0 = np.array([parity(int(f) & 0o155) for f in F0], np.uint32) * np.uint32(0x01010101)
B0 = np.array([parity(int(f) & 0o117) for f in F0], np.uint32) * np.uint32(0x01010101)
A1 = np.array([parity(int(f) & 0o155) for f in F1], np.uint32) * np.uint32(0x01010101)
B1 = np.array([parity(int(f) & 0o117) for f in F1], np.uint32) * np.uint32(0x01010101)


def packed_viterbi4(cb):
    """cb: four 48-bit integers (bit j = coded bit j).  Returns the four 24-bit results and the largest metric byte seen."""
    pm = np.where(S == 0, np.uint32(0), np.uint32(0x40404040)).astype(np.uint32)
    surv = np.zeros((4, 24), np.uint64)
    top_byte = 0
    for t in range(24):
        ra = rb = 0
        for f in range(4):
            two = (cb[f] >> (2 * t)) & 3
            ra |= (two & 1) << (8 * f)
            rb |= (two >> 1) << (8 * f)
        ra, rb = np.uint32(ra), np.uint32(rb)
        m0 = pm[P0] + ((ra ^ A0) + (rb ^ B0))
        m1 = pm[P1] + ((ra ^ A1) + (rb ^ B1))
        top = ((m0 + np.uint32(0x7f7f7f7f)) - m1) & np.uint32(0x80808080)
        mask = (top - (top >> np.uint32(7))) | top
        pm = (m1 & mask) | (m0 & ~mask)
        top_byte = max(top_byte, int(((pm[:, None] >> (8 * np.arange(4, dtype=np.uint32))) & 0xff).max()))
        for f in range(4):
            bits = ((top >> np.uint32(8 * f + 7)) & 1).astype(np.uint64)
            surv[f, t] = (bits << S.astype(np.uint64)).sum()
    out = []
    for f in range(4):
        key = (((pm >> np.uint32(8 * f)) & 0xff) << 6) | S.astype(np.uint32)
        st = int(key.min()) & 63
        bits = 0
        for t in range(23, -1, -1):
            bits |= (st & 1) << t
            h = (int(surv[f, t]) >> st) & 1
            st = (st >> 1) | (h << 5)
        out.append(bits)
    return out, top_byte


def conv_encode(bits24):
    state, out = 0, []
    for b in bits24:
        reg = (int(b) << 6) | state
        out += [parity(reg & 0o155), parity(reg & 0o117)]
        state = reg >> 1
    return out


def test_packed_byte_metrics_decode_like_the_oracle(orc):
    rng = np.random.default_rng(424)
    worst = 0
    for trial in range(300):
        cbs, want = [], []
        for f in range(4):
            kind = (trial + f) % 3
            if kind == 2:
                coded = rng.integers(0, 2, 48)                                   # noise only: ties and unreachable states matter
            else:
                msg = rng.integers(0, 2, 24)
                msg[18:] = 0                                                     # tail bits, as a SIGNAL field has them
                coded = np.array(conv_encode(msg))
                flips = rng.random(48) < (0.0 if kind == 0 else 0.12)
                coded = coded ^ flips
            cbs.append(int(sum(int(c) << j for j, c in enumerate(coded))))
            want.append(orc.viterbi(coded.astype(np.uint8), 24))
        got, top = packed_viterbi4(cbs)
        worst = max(worst, top)
        for f in range(4):
            assert [(got[f] >> t) & 1 for t in range(24)] == list(want[f]), (trial, f)
    assert worst < 128                                                            # the guard bit of the compare stays free
