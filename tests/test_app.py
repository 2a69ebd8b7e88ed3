"""Application layer (row f3): the piece wire format and image rebuild of the reference, pinned by
datagrams made with the reference's own image_detach_rebuild.detach_image (tests/golden)."""
import os
import pickle
import struct

import numpy as np

from wifirx import app, txgen

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def golden_datagrams():
    z = np.load(os.path.join(GOLD, "pieces_kodim01.npz"))
    return [bytes(z["datagrams"][i, :z["lengths"][i]]) for i in range(len(z["lengths"]))], int(z["n_pieces_total"])


def test_piece_wire_format_matches_reference():
    grams, n_total = golden_datagrams()
    img = np.load(os.path.join(GOLD, "kodim_300.npz"))["kodim01"]
    pcs = app.detach_image_sorted(img)
    assert n_total == len(pcs) == 2700
    for g, p in zip(grams, pcs):
        (n,) = struct.unpack("=L", g[:4])
        assert n == len(g) - 4 and 262 <= n <= 268
        key, arr = app.load_piece(g[4:])
        assert key == p[0] and arr.shape == (10, 10, 1) and arr.dtype == np.uint8 and np.array_equal(arr, p[1])
        k2, a2 = app.load_piece(app.pack_piece(p)[4:])
        assert k2 == key and np.array_equal(a2, arr)


def test_load_piece_refuses_foreign_pickles():
    import pytest
    evil = pickle.dumps(os.system)
    with pytest.raises(pickle.UnpicklingError):
        app.load_piece(evil)


def test_extract_pics_strips_mac_header_and_length():
    got = []
    blk = app.extract_pics(sink=got.append)
    payload = app.pack_piece(((0, 10, 2), np.arange(100, dtype=np.uint8).reshape(10, 10, 1)))
    psdu = txgen.mac_frame(payload)
    blk.handle_msg(({"snr": 20.0}, np.frombuffer(psdu[:-4], dtype=np.uint8)))
    assert got == [payload[4:]]
    key, arr = app.load_piece(got[0])
    assert key == (0, 10, 2)


def test_image_round_trip_through_the_oracle_stream(orc):
    """config-1 in miniature: pieces -> MAC -> BPSK-1/2 frames with packet_pad2 gaps -> RX -> redraw."""
    img = np.load(os.path.join(GOLD, "kodim_300.npz"))["kodim01"]
    pcs = app.detach_image_sorted(img)[:12]
    stream = []
    for k, p in enumerate(pcs):
        psdu = np.frombuffer(txgen.mac_frame(app.pack_piece(p), seq=k), dtype=np.uint8)[None]
        tx = txgen.encode_psdus(psdu, 0, seeds=[k + 1])
        stream.append(txgen.packet_pad(tx.samples * 8.0))
    x = np.concatenate(stream)
    x = x + ((np.random.default_rng(0).standard_normal(x.size) + 1j * np.random.default_rng(1).standard_normal(x.size)) * 0.5).astype(np.complex64)
    prm = orc.make_params(max_sym=128)
    o = orc.demod_stream(x, prm)
    psdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=512)
    out = np.zeros_like(img)
    got = []
    blk = app.extract_pics(sink=got.append)
    for f, p in zip(o["frames"], psdu):
        if f["flags"] & orc.F_CRC_OK:
            blk.handle_msg(({}, p[:int(f["psdu_len"]) - 4]))
    assert len(got) == len(pcs)
    for g in got:
        app.redraw_image(app.load_piece(g), out)
    for (y, x_, c), piece in pcs:
        assert np.array_equal(out[y:y + 10, x_:x_ + 10, c:c + 1], piece)
