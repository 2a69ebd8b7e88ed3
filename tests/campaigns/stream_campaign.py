#!/usr/bin/env python3
"""Randomised stream-mode parity: continuous streams of frames (random rate, length, SNR, CFO, gaps -- some gaps
shorter than sync_short's re-trigger distance, some frames cut off by the next one), pushed in random chunk sizes
with a random WIFIRX_P_STREAM_BATCH, against the oracle's stream driver: same frames, same records, same
decisions, same PSDUs.  Prints one JSON line.

    python tests/campaigns/stream_campaign.py [n_streams=40] [frames_per_stream=30] [seed=1]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
sys.path.insert(0, ROOT)
from wifirx import capi, txgen  # noqa: E402


def make_stream(rng, n_frames):
    parts = [np.zeros(int(rng.integers(0, 400)), np.complex64)]
    for k in range(n_frames):
        enc = int(rng.integers(0, 8))
        plen = int(rng.integers(30, 700))
        psdu = txgen.make_psdus(1, plen, seed=int(rng.integers(1 << 30)), seq0=k)
        tx = txgen.encode_psdus(psdu, enc, seeds=[int(rng.integers(1, 128))])
        n = tx.samples.shape[1]
        snr = float(rng.choice([8.0, 15.0, 22.0, 30.0]))
        sig = tx.samples[0] * np.exp(1j * rng.uniform(-0.04, 0.04) * np.arange(n)) * np.sqrt(10 ** (snr / 10))
        gap = int(rng.choice([0, 16, 100, 300, 480, 1000, 3000]))
        parts += [sig.astype(np.complex64), np.zeros(gap, np.complex64)]
    x = np.concatenate(parts)
    x = x + ((rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size)) * np.sqrt(0.5)).astype(np.complex64)
    return x.astype(np.complex64)


def run(n_streams=40, per=30, seed=1):
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    tot = {"streams": n_streams, "frames_sent": n_streams * per, "frames_found": 0, "crc_ok": 0, "samples": 0,
           "mismatching_records": 0, "mismatching_decisions": 0, "mismatching_psdu_bytes": 0, "frame_count_differs": 0}
    for s in range(n_streams):
        x = make_stream(rng, per)
        ce = int(rng.integers(0, 4))
        prm = orc.make_params(max_sym=511, chan_est=ce)
        o = orc.demod_stream(x, prm, cap=4 * per + 16)
        opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=2048)
        rx = capi.WifiRx(max_sym=511, chan_est=ce)
        rx.set_param(capi.P_STREAM_BATCH, int(rng.choice([0, 0, 5000, 100000])))
        got, pos = [], 0
        while pos < x.size:
            c = int(rng.choice([257, 4096, 8192, 50000, 400000]))
            rx.push(x[pos:pos + c])
            pos += c
            got.append(rx.poll(cap=256, want_idx=True))
        rx._check(capi.lib().wifirx_push(rx._h, None, 0, 0))
        got.append(rx.poll(cap=256, want_idx=True))
        rx.close()
        frames = np.concatenate([g["frames"] for g in got])
        psdu = np.concatenate([g["psdu"] for g in got])
        idx = np.concatenate([g["idx"] for g in got])
        tot["samples"] += int(x.size)
        tot["frames_found"] += len(frames)
        if len(frames) != len(o["frames"]):
            tot["frame_count_differs"] += 1
            continue
        tot["mismatching_records"] += int((frames != o["frames"]).sum())
        for k in range(len(frames)):
            n = int(frames[k]["n_sym_out"])
            tot["mismatching_decisions"] += int((idx[k, :n] != o["idx"][k, :n]).sum())
            if frames[k]["flags"] & capi.F_CRC_OK:
                L = int(frames[k]["psdu_len"])
                tot["crc_ok"] += 1
                tot["mismatching_psdu_bytes"] += int((psdu[k, :L] != opsdu[k, :L]).sum())
    tot["all_equal"] = (tot["frame_count_differs"] + tot["mismatching_records"] + tot["mismatching_decisions"] +
                        tot["mismatching_psdu_bytes"]) == 0
    tot["seconds"] = time.perf_counter() - t0
    return tot


def main():
    n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    print(json.dumps(run(n_streams, per, seed)))


if __name__ == "__main__":
    main()
