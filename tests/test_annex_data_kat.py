"""Known answers for the DATA field path of the transmitter every loop-back test relies on (wifirx/txgen.py), from IEEE Std
802.11 clause 17/18 (OFDM PHY) and its Annex worked example (36 Mb/s, 100-octet PSDU, scrambler state 1011101) -- stated
only where the value can be given with confidence, each with its provenance -- and direct float64 checks of the tables
the oracle and the kernels share (include/wifirx_tables.h), so that a wrong entry is not common-mode.

This does NOT pin the receiver to the reference (the reference holds no vectors: parity unpinned);
it pins the transmitter and the shared constants to the standard instead of to this repository's own receiver."""
import math
import os
import re

import numpy as np

from wifirx import txgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Annex example message: MAC header + "Joy, bright spark of divinity,\nDaughter of Elysium,\nFire-insired we trea" + FCS
ANNEX_PSDU = bytes.fromhex("0402002e006008cd37a60020d6013cf1006008ad3baf0000") + \
    b"Joy, bright spark of divinity,\nDaughter of Elysium,\nFire-insired we trea" + bytes.fromhex("673321b6")

# Annex table "scrambling sequence for seed 1011101" (127 bits).  Provenance: the table as remembered; the memory is
# confirmed by the LFSR x^7 + x^4 + 1 started in state 1011101 reproducing all 127 bits (test below does it again).
ANNEX_SCRAMBLE_SEQ = ("0110110000011001101010011100111101101000010101011111010010100011"
                      "011100011111110000111011110010110010010000001000100110001011101")


def test_annex_scrambling_sequence_for_state_1011101():
    state, out = [1, 0, 1, 1, 1, 0, 1], []            # x7 .. x1
    for _ in range(127):
        fb = state[0] ^ state[3]
        out.append(fb)
        state = state[1:] + [fb]
    assert "".join(map(str, out)) == ANNEX_SCRAMBLE_SEQ
    assert "".join(map(str, txgen.scrambler_sequence(0b1011101, 127))) == ANNEX_SCRAMBLE_SEQ


def test_annex_data_field_geometry_and_scrambled_service():
    """36 Mb/s, LENGTH 100: N_DBPS 144, 6 OFDM symbols, 864 DATA bits = 16 SERVICE + 800 PSDU + 6 tail + 42 pad; the
    SERVICE field is all zeros before scrambling, so the first 16 scrambled bits are the first 16 of the sequence
    (0110 1100 0001 1001: how the Annex's table of scrambled bits opens, and what a receiver recovers the state from)."""
    assert len(ANNEX_PSDU) == 100
    assert txgen.RATE_TABLE[5][:3] == (4, 192, 144)
    assert txgen.n_sym_for(100, 5) == 6
    assert 6 * 144 - (16 + 800 + 6) == 42
    tx = txgen.encode_psdus(np.frombuffer(ANNEX_PSDU, np.uint8)[None], 5, seeds=[0b1011101])
    assert tx.n_sym == 6 and tx.samples.shape[1] == (5 + 6) * 80 + 1
    # undo the transmitter's own steps for the first symbol with independent code: constellation index -> bits ->
    # de-interleave (standard's two permutations written out) -> 192 coded bits of symbol 0
    idx = tx.data_idx[0, 0]
    bits = ((idx[:, None] >> np.arange(4)) & 1).reshape(-1)                   # position j = 4 * carrier + bit
    n_cbps, s = 192, 2
    coded = np.empty(192, np.uint8)
    for k in range(192):
        i = (n_cbps // 16) * (k % 16) + k // 16                               # first permutation (eq. 18-18)
        j = s * (i // s) + (i + n_cbps - (16 * i) // n_cbps) % s              # second permutation (eq. 18-19)
        coded[k] = bits[j]
    # rate 3/4: of every six coded bits A1 B1 A2 B2 A3 B3 the pair B2, A3 is stolen; a rate-1/2 encoder of the scrambled
    # bits must reproduce the 192 transmitted ones
    seq = np.array(list(map(int, ANNEX_SCRAMBLE_SEQ)), np.uint8)
    data = np.concatenate([np.zeros(16, np.uint8), np.unpackbits(np.frombuffer(ANNEX_PSDU, np.uint8), bitorder="little"),
                           np.zeros(6 + 42, np.uint8)])
    scr = data ^ np.resize(seq, data.size)
    scr[16 + 800:16 + 800 + 6] = 0                                            # the six tail bits are zeroed after scrambling
    assert "".join(map(str, scr[:16])) == "0110110000011001"
    reg, mother = [0] * 6, []
    for b in scr[:144]:
        w = [int(b)] + reg
        mother += [w[0] ^ w[2] ^ w[3] ^ w[5] ^ w[6], w[0] ^ w[1] ^ w[2] ^ w[3] ^ w[6]]      # g0 = 133, g1 = 171 (octal)
        reg = w[:6]
    kept = [mother[i] for i in range(288) if i % 6 not in (3, 4)]
    assert kept == coded.tolist()


def test_rate_field_bits_of_all_eight_rates():
    """SIGNAL RATE field R1-R4 (standard's rate table): 6: 1101, 9: 1111, 12: 0101, 18: 0111, 24: 1001, 36: 1011,
    48: 0001, 54: 0011; LENGTH LSB first in bits 5..16; bit 17 even parity; bits 18..23 zero tail"""
    want = {0: "1101", 1: "1111", 2: "0101", 3: "0111", 4: "1001", 5: "1011", 6: "0001", 7: "0011"}
    for enc, r in want.items():
        b = txgen.signal_bits(enc, 0x5A3)
        assert "".join(map(str, b[:4])) == r and b[4] == 0
        assert sum(int(b[5 + i]) << i for i in range(12)) == 0x5A3
        assert b[:18].sum() % 2 == 0 and not b[18:].any()
    assert "".join(map(str, txgen.signal_bits(5, 100))) == "101100010011000000000000"       # the Annex's SIGNAL bits


def test_gray_mappings_and_normalisation():
    """Constellation tables of the standard: BPSK 0 -> -1; QPSK b0 -> I, b1 -> Q, 0 -> -1; 16-QAM (b0 b1) 00 -> -3,
    01 -> -1, 11 -> +1, 10 -> +3 on I, (b2 b3) likewise on Q; 64-QAM (b0 b1 b2) 000 -> -7, 001 -> -5, 011 -> -3,
    010 -> -1, 110 -> +1, 111 -> +3, 101 -> +5, 100 -> +7; K_MOD = 1, 1/sqrt 2, 1/sqrt 10, 1/sqrt 42.  Index bit k = b_k."""
    ax16 = {(0, 0): -3, (0, 1): -1, (1, 1): 1, (1, 0): 3}
    ax64 = {(0, 0, 0): -7, (0, 0, 1): -5, (0, 1, 1): -3, (0, 1, 0): -1, (1, 1, 0): 1, (1, 1, 1): 3, (1, 0, 1): 5, (1, 0, 0): 7}
    p = txgen.constellation_points(1)
    assert p[0] == -1 and p[1] == 1
    p = txgen.constellation_points(2)
    for i in range(4):
        assert abs(p[i] - ((2 * (i & 1) - 1) + 1j * (2 * (i >> 1) - 1)) / math.sqrt(2)) < 1e-15
    p = txgen.constellation_points(4)
    for i in range(16):
        b = [(i >> k) & 1 for k in range(4)]
        assert abs(p[i] - (ax16[(b[0], b[1])] + 1j * ax16[(b[2], b[3])]) / math.sqrt(10)) < 1e-15
    p = txgen.constellation_points(6)
    for i in range(64):
        b = [(i >> k) & 1 for k in range(6)]
        assert abs(p[i] - (ax64[tuple(b[:3])] + 1j * ax64[tuple(b[3:])]) / math.sqrt(42)) < 1e-15
    for nb in (1, 2, 4, 6):
        assert abs(np.mean(np.abs(txgen.constellation_points(nb)) ** 2) - 1.0) < 1e-12     # unit average power


def test_puncturing_patterns():
    x = (np.arange(24, dtype=np.uint8) % 251)[None]
    assert txgen.puncture(x, "1/2")[0].tolist() == list(range(24))
    assert txgen.puncture(x, "2/3")[0].tolist() == [i for i in range(24) if i % 4 != 3]          # A1 B1 A2 (B2 stolen)
    assert txgen.puncture(x, "3/4")[0].tolist() == [i for i in range(24) if i % 6 not in (3, 4)]  # A1 B1 A2 B3


# ---- the tables oracle and kernels share, against float64 formulas (a wrong entry would be common-mode otherwise) ----
def _table(name):
    txt = open(os.path.join(ROOT, "include", "wifirx_tables.h")).read()
    m = re.search(r"%s\[\d+\] = \{(.*?)\};" % name, txt, re.S)
    vals = [v.strip() for v in m.group(1).replace("\n", " ").split(",") if v.strip()]
    return np.array([float.fromhex(v.rstrip("f")) if "x" in v else float(v.rstrip("f")) for v in vals])


def _define(name):
    txt = open(os.path.join(ROOT, "include", "wifirx_tables.h")).read()
    return float.fromhex(re.search(r"#define %s (\S+?)f?\n" % name, txt).group(1))


def test_slicer_thresholds_and_levels():
    f32 = lambda v: float(np.float32(v))
    a16, a64 = 1 / math.sqrt(10), 1 / math.sqrt(42)
    assert _define("WR_LEVEL_QPSK") == f32(1 / math.sqrt(2))
    assert _define("WR_LEVEL_16QAM") == f32(a16) and _define("WR_LEVEL_64QAM") == f32(a64)
    # thresholds formed in float32 from the float32 level, like the upstream decision makers: 2a, 4a, 6a
    assert _define("WR_T16_2") == f32(np.float32(2) * np.float32(a16))
    assert _define("WR_T64_2") == f32(np.float32(2) * np.float32(a64))
    assert _define("WR_T64_4") == f32(np.float32(4) * np.float32(a64))
    assert _define("WR_T64_6") == f32(np.float32(6) * np.float32(a64))
    # and they sit half way between neighbouring levels
    assert abs(_define("WR_T16_2") - 2 * a16) < 1e-7 and abs(_define("WR_T64_6") - 6 * a64) < 1e-7


def test_comb_interpolation_weights():
    w, u = _table("WR_COMB_W"), _table("WR_COMB_U")
    nodes = [0, 11, 25, 39, 53, 64]
    for i in range(64):
        k = 0 if i <= 11 else 1 if i <= 25 else 2 if i <= 39 else 3 if i <= 53 else 4
        ww = (i - nodes[k]) / (nodes[k + 1] - nodes[k])
        assert abs(w[i] - ww) <= 6e-8 and abs(u[i] - (1 - ww)) <= 6e-8 and 0 <= w[i] <= 1
    for n in (11, 25, 39, 53):                               # on a pilot the estimate IS the pilot
        assert w[n] == 1.0 and u[n] == 0.0


def test_sampling_offset_factor_table():
    t = _table("WR_T4_64")
    assert len(t) == 520
    s = np.arange(520, dtype=np.float64)
    assert np.array_equal(t, (((2 * math.pi) * s) * 80.0) / 64.0)          # the C expression's operation order, exactly
    assert np.allclose(t, 2 * math.pi * s * 80 / 64, rtol=1e-15)


def test_lts_correlation_operand_table_matches_the_lts():
    """WR_LTS_MFMA_B (kernels only) against WR_LTS_TIME through the layout rule of DESIGN.md rule 6, written out
    independently of tools/gen_tables.py: entry [j][lane][s'] multiplies float phi = 16 j + 4 (lane >> 4) + s' of a row."""
    lts = _table("WR_LTS_TIME").reshape(64, 2)
    b = _table("WR_LTS_MFMA_B").reshape(9, 64, 4)
    for j in range(9):
        for lane in range(64):
            for sp in range(4):
                phi = 16 * j + 4 * (lane >> 4) + sp
                m, part, col = phi // 2, phi % 2, lane & 15
                k = m - (col & 7)
                want = 0.0
                if 0 <= k < 64:
                    lr, li = lts[k]
                    want = (li if part else lr) if col < 8 else (lr if part else -li)
                assert b[j, lane, sp] == want or (want == 0 and b[j, lane, sp] == 0)
    # the LTS itself: 64-point inverse DFT of L(-26..26) with the reference's 1/sqrt(52) scale
    L = np.zeros(64)
    for k, v in zip(range(-26, 27), [1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
                                     1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1]):
        L[k % 64] = v
    ref = np.fft.ifft(L) * 64 / math.sqrt(52)
    assert np.abs(lts[:, 0] + 1j * lts[:, 1] - ref).max() < 1e-7


def _int_table(name):
    txt = open(os.path.join(ROOT, "include", "wifirx_tables.h")).read()
    m = re.search(r"%s\[\d+\] = \{(.*?)\};" % name, txt, re.S)
    return np.array([int(v.strip().rstrip("u"), 0) for v in m.group(1).replace("\n", " ").split(",") if v.strip()], dtype=np.int64)


def test_integer_lts_tables_match_the_lts():
    """spec rule 6, stage 1: WR_LTS_Q8 = rint(64 l), and WR_LTS_MFMA_A8 (kernels only: the taps as the row-side operand of
    v_mfma_i32_16x16x64_i8) against it through the instruction's layout written out independently of tools/gen_tables.py:
    byte j of entry [t][lane] multiplies byte phi = 64 t + 16 (lane >> 4) + j of a sample column; row = lane & 15 stands for
    the lag offset b = 2 (row >> 2) + ((row & 3) >> 1) inside a block of 8 and the part row & 1 (0 real, 1 imaginary sum);
    tap k = (phi >> 1) - b, zero outside the 64 taps"""
    lts = _table("WR_LTS_TIME")
    q8 = _int_table("WR_LTS_Q8")
    assert np.array_equal(q8, np.rint(64.0 * lts).astype(np.int64)) and np.abs(q8).max() <= 127
    words = _int_table("WR_LTS_MFMA_A8").reshape(3, 64, 4)
    seen = set()
    for t in range(3):
        for lane in range(64):
            by = np.array([(int(words[t, lane, j >> 2]) >> (8 * (j & 3))) & 0xff for j in range(16)], dtype=np.uint8).view(np.int8)
            row = lane & 15
            b, outpart = 2 * (row >> 2) + ((row & 3) >> 1), row & 1
            seen.add((b, outpart))
            for j in range(16):
                phi = 64 * t + 16 * (lane >> 4) + j
                m, part = phi // 2, phi % 2
                k = m - b
                want = 0
                if 0 <= k < 64:
                    qr, qi = q8[2 * k], q8[2 * k + 1]
                    want = (qi if part else qr) if outpart == 0 else (qr if part else -qi)
                assert by[j] == want, (t, lane, j)
    assert seen == {(b, p) for b in range(8) for p in range(2)}


def test_integer_lts_search_finds_the_peaks_of_the_float_search(orc):
    """the two-stage search (rule 6) against the exhaustive float search of the upstream-literal mode: same frame start
    and flags on frames from clean to barely detectable, flat and multipath"""
    from wifirx import txgen
    rng = np.random.default_rng(3)
    taps = (rng.standard_normal((64, 6)) + 1j * rng.standard_normal((64, 6))) * np.exp(-0.6 * np.arange(6))
    taps /= np.sqrt((np.abs(taps) ** 2).sum(axis=1, keepdims=True))
    for snr, tp in ((30.0, None), (8.0, None), (3.0, None), (18.0, taps), (9.0, taps)):
        tx = txgen.encode_psdus(txgen.make_psdus(64, 60, seed=int(snr)), 0)
        iq = txgen.impair(tx.samples, snr, cfo=rng.uniform(-0.03, 0.03, 64), lead=200, total=2048, seed=int(snr) + 9, taps=tp)
        a = orc.demod_batch(iq.reshape(-1), 2048, orc.make_params(max_sym=tx.n_sym, math_mode=orc.MATH_SPEC))["frames"]
        b = orc.demod_batch(iq.reshape(-1), 2048, orc.make_params(max_sym=tx.n_sym, math_mode=orc.MATH_LIBM))["frames"]
        assert np.array_equal(a["frame_start"], b["frame_start"]) and np.array_equal(a["flags"], b["flags"])
        assert ((a["flags"] & orc.F_SYNC) != 0).mean() > (0.9 if snr > 5 else 0.3)
