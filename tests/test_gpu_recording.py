"""BASELINE config 5's recording (the six Kodak images as one continuous IQ stream) cut 2- and 3-ways across ranks
(wifirx.dist.demod_recording_sharded with the GPU stream engine): the same PDU list as one rank, in
stream order, pixel-exact.  One process, the ranks one after the other; the multi-process form with the PDU all-gather
is tools/recording_sharded.py (run through torchrun on the one-GPU pool: profiles/r03_recording_sharded_*.json) and, on
the CPU with the oracle as engine, tests/test_dist_recording.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_recording(names=None, n_pieces=None, snr_amp=10.0):
    from wifirx import app, txgen
    imgs = np.load(os.path.join(ROOT, "tests", "golden", "kodim_300.npz"))
    parts, truth = [], []
    for name in sorted(imgs.files if names is None else names):
        img = imgs[name]
        pieces = app.detach_image_sorted(img)[:n_pieces]
        payloads = [app.pack_piece(p) for p in pieces]
        streams = [None] * len(payloads)
        by_len = {}
        for k, p in enumerate(payloads):
            by_len.setdefault(len(p), []).append(k)
        for L, ks in by_len.items():
            psdus = np.stack([np.frombuffer(txgen.mac_frame(payloads[k], seq=k), dtype=np.uint8) for k in ks])
            tx = txgen.encode_psdus(psdus, 0, seeds=[(k % 127) + 1 for k in ks])
            for row, k in enumerate(ks):
                streams[k] = tx.samples[row] * np.float32(snr_amp)
        parts += [np.concatenate([np.zeros(100, np.complex64), s, np.zeros(1000, np.complex64)]) for s in streams]
        truth.append((name, img, pieces))
    x = np.concatenate(parts)
    rng = np.random.default_rng(11)
    x += ((rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size)) * np.sqrt(0.5)).astype(np.complex64)
    return x, truth


@pytest.mark.timeout(900)
def test_config5_recording_cut_two_and_three_ways():
    from wifirx import app, capi, dist as wdist
    x, truth = make_recording()
    n_frames = sum(len(p) for _, _, p in truth)
    assert n_frames == 6 * 2700
    eng = wdist.gpu_stream_engine(max_sym=128, bandwidth=20e6, frequency=5.89e9)
    one_f, one_p = eng(x)
    ok1 = (one_f["flags"] & capi.F_CRC_OK) != 0
    assert int(ok1.sum()) == n_frames
    for world in (2, 3):
        res = wdist.demod_recording_sharded(x, eng, world)
        assert all(p["seam_ok"] for p in res["parts"])
        assert np.array_equal(res["frames"], one_f), world          # records incl. the absolute trigger index
        w = min(res["psdu"].shape[1], one_p.shape[1])
        assert np.array_equal(res["psdu"][:, :w], one_p[:, :w]), world
        counts = [len(p["frames"]) for p in res["parts"]]
        assert sum(counts) == len(one_f) and min(counts) > 0.2 * len(one_f)
        # the user-visible result: every image pixel-exact from the sharded PDU stream
        ok = (res["frames"]["flags"] & capi.F_CRC_OK) != 0
        rows = np.nonzero(ok)[0]
        k = 0
        for name, img, pieces in truth:
            out = np.zeros_like(img)
            for _ in pieces:
                r = rows[k]; k += 1
                L = int(res["frames"]["psdu_len"][r])
                app.redraw_image(app.load_piece(bytes(res["psdu"][r, 24 + 4:L - 4])), out)
            assert np.array_equal(out, img), (world, name)


def test_short_pre_roll_cuts_inside_frames():
    """a pre-roll of 4096 samples (instead of two maximal frames) puts every seam's start inside a frame: the seam check
    and the ownership rule still give the one-rank result"""
    from wifirx import dist as wdist
    x, _ = make_recording(names=["kodim01"], n_pieces=300)
    eng = wdist.gpu_stream_engine(max_sym=128, bandwidth=20e6, frequency=5.89e9)
    one_f, one_p = eng(x)
    for world in (2, 5):
        res = wdist.demod_recording_sharded(x, eng, world, pre_roll=4096)
        assert np.array_equal(res["frames"], one_f) and np.array_equal(res["psdu"], one_p), world
