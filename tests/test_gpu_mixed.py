"""Waves whose four / sixty-four frames differ: mixed rates and lengths inside one batch, bit for bit
against the oracle (demod quad kernel: rows with different constellations; decode kernel: lanes with
different trellis lengths and puncturing)."""
import numpy as np
import pytest

from wifirx import txgen

pytestmark = pytest.mark.gpu


def mixed_batch(n=150, slot_len=3200, seed=5):
    rng = np.random.default_rng(seed)
    iq = np.zeros((n, slot_len), np.complex64)
    meta = []
    for k in range(n):
        enc = int(rng.integers(0, 8))
        plen = int(rng.integers(30, 120)) if enc < 2 else int(rng.integers(30, 400))
        psdu = txgen.make_psdus(1, plen, seed=seed * 1000 + k)
        tx = txgen.encode_psdus(psdu, enc, seeds=[k % 127 + 1])
        if tx.samples.shape[1] + 200 > slot_len:
            plen = 60
            psdu = txgen.make_psdus(1, plen, seed=seed * 1000 + k)
            tx = txgen.encode_psdus(psdu, enc, seeds=[k % 127 + 1])
        snr = 30.0 if k % 7 else 9.0              # some frames fail SIGNAL / CRC
        lead = int(rng.integers(20, 200))
        iq[k] = txgen.impair(tx.samples, snr, cfo=rng.uniform(-0.03, 0.03), lead=lead, total=slot_len, seed=seed + k)[0]
        meta.append((enc, plen, psdu[0]))
    return iq.reshape(-1), slot_len, meta


def test_mixed_rates_in_one_batch(orc, decode_path):
    from wifirx import capi
    iq, slot_len, meta = mixed_batch()
    max_sym = 140
    rx = capi.WifiRx(max_sym=max_sym, llr_bits=6, want_carrier=True)
    r = rx.demod_batch(iq, slot_len, decode=True, psdu_stride=512)
    prm = orc.make_params(max_sym=max_sym, llr_bits=6)
    o = orc.demod_batch(iq, slot_len, prm, want_eq=True)
    opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=512)
    assert np.array_equal(r["frames"], o["frames"])
    assert np.array_equal(r["idx"], o["idx"]) and np.array_equal(r["llr"], o["llr"]) and np.array_equal(r["carrier"], o["eq"])
    dec = (o["frames"]["flags"] & orc.F_DECODED) != 0
    for k in np.nonzero(dec)[0]:
        L = int(o["frames"]["psdu_len"][k])
        assert np.array_equal(r["psdu"][k, :L], opsdu[k, :L])
    ok = (r["frames"]["flags"] & capi.F_CRC_OK) != 0
    assert ok.sum() > 100
    for k in np.nonzero(ok)[0]:
        enc, plen, psdu = meta[k]
        assert r["frames"]["encoding"][k] == enc and r["frames"]["psdu_len"][k] == plen
        assert np.array_equal(r["psdu"][k, :plen], psdu)
    rx.close()


def test_config1_kodim01_pieces_through_the_block():
    """BASELINE config 1 in the GPU block: 1000 BPSK-1/2 frames carrying the first 1000 pieces of kodim01
    (reference wire format), packet_pad2 gaps, through wifi_phy_rx + Extract Pics -> pixels."""
    import os
    from wifirx import app, block, grshim
    gold = os.path.join(os.path.dirname(__file__), "golden")
    img = np.load(os.path.join(gold, "kodim_300.npz"))["kodim01"]
    pieces = app.detach_image_sorted(img)[:1000]
    payloads = [app.pack_piece(p) for p in pieces]
    assert len({len(p) for p in payloads}) <= 3
    rng = np.random.default_rng(1)
    chunks = []
    by_len = {}
    for k, p in enumerate(payloads):
        by_len.setdefault(len(p), []).append(k)
    streams = [None] * len(payloads)
    for L, ks in by_len.items():
        psdus = np.stack([np.frombuffer(txgen.mac_frame(payloads[k], seq=k), dtype=np.uint8) for k in ks])
        tx = txgen.encode_psdus(psdus, 0, seeds=[(k % 127) + 1 for k in ks])
        assert tx.n_sym in (99, 100)
        for row, k in enumerate(ks):
            streams[k] = tx.samples[row] * np.float32(10.0)      # 20 dB over the unit noise below
    x = np.concatenate([np.concatenate([np.zeros(100, np.complex64), s, np.zeros(1000, np.complex64)]) for s in streams])
    x = x + ((rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size)) * np.sqrt(0.5)).astype(np.complex64)
    got = []
    rx = block.wifi_phy_rx(bandwidth=20e6, frequency=5.89e9, publish_carrier=False)
    pics = app.extract_pics(sink=got.append)
    grshim.msg_connect(rx, "mac_out", pics, "MAC")
    grshim.run_stream(rx, x, chunk=1 << 20)
    assert len(got) == 1000
    out = np.zeros_like(img)
    for g in got:
        app.redraw_image(app.load_piece(g), out)
    for (y, xx, c), piece in pieces:
        assert np.array_equal(out[y:y + 10, xx:xx + 10, c:c + 1], piece)
    st = rx.stats()
    assert st["frames_crc_ok"] == 1000


def test_two_handles_on_two_threads(orc):
    """handles are independent (own HIP stream, no global state): two threads, one handle each, concurrently"""
    import threading
    from wifirx import capi
    from helpers import make_slots
    jobs = []
    for enc, seed in ((2, 11), (5, 12)):
        iq, slot_len, tx = make_slots(200, enc, snr_db=25.0, seed=seed)
        prm = orc.make_params(max_sym=tx.n_sym, llr_bits=6)
        o = orc.demod_batch(iq, slot_len, prm)
        jobs.append((iq, slot_len, tx, o))
    results = [None, None]

    def run(k):
        iq, slot_len, tx, _ = jobs[k]
        rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=6)
        for _ in range(5):
            results[k] = rx.demod_batch(iq, slot_len, decode=True, psdu_stride=320)
        rx.close()

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        _, _, tx, o = jobs[k]
        r = results[k]
        assert np.array_equal(r["idx"], o["idx"]) and np.array_equal(r["llr"], o["llr"])
        assert ((r["frames"]["flags"] & capi.F_CRC_OK) != 0).all()
        assert np.array_equal(r["psdu"][:, :294], tx.psdu)
