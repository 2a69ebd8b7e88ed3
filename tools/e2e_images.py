#!/usr/bin/env python3
"""BASELINE.json config 5, minus the sockets: every piece of the six Kodak images (resized 300x300 as
upload_image_udp.py:21 does; fixtures in tests/golden/kodim_300.npz) -> reference wire format ->
ieee802_11.mac framing -> BPSK-1/2 802.11a frames with packet_pad2 gaps (CPU TX, wifirx/txgen.py) ->
AWGN -> wifi_phy_rx block on the MI355X (stream mode, host buffers through work()) -> Extract Pics ->
redraw.  Prints one JSON line: pixel-exactness per image and the host-path throughput of work()."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
from wifirx import app, block, grshim, txgen  # noqa: E402


def main():
    imgs = np.load(os.path.join(ROOT, "tests", "golden", "kodim_300.npz"))
    snr_db = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 8192        # items per work() call (GNU Radio hands out thousands)
    res = {"snr_db": snr_db, "images": {}, "encoding": "BPSK_1_2", "work_chunk": chunk}
    total_samples, total_time = 0, 0.0
    for name in sorted(imgs.files):
        img = imgs[name]
        pieces = app.detach_image_sorted(img)
        payloads = [app.pack_piece(p) for p in pieces]
        streams = [None] * len(payloads)
        by_len = {}
        for k, p in enumerate(payloads):
            by_len.setdefault(len(p), []).append(k)
        for L, ks in by_len.items():
            psdus = np.stack([np.frombuffer(txgen.mac_frame(payloads[k], seq=k), dtype=np.uint8) for k in ks])
            tx = txgen.encode_psdus(psdus, 0, seeds=[(k % 127) + 1 for k in ks])
            for row, k in enumerate(ks):
                streams[k] = tx.samples[row] * np.float32(np.sqrt(10 ** (snr_db / 10)))
        x = np.concatenate([np.concatenate([np.zeros(100, np.complex64), s, np.zeros(1000, np.complex64)]) for s in streams])
        rng = np.random.default_rng(hash(name) & 0xffff)
        x = x + ((rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size)) * np.sqrt(0.5)).astype(np.complex64)
        got = []
        rx = block.wifi_phy_rx(bandwidth=20e6, frequency=5.89e9, publish_carrier=False)
        pics = app.extract_pics(sink=got.append)
        grshim.msg_connect(rx, "mac_out", pics, "MAC")
        t = time.perf_counter()
        grshim.run_stream(rx, x, chunk=chunk)
        dt = time.perf_counter() - t
        out = np.zeros_like(img)
        for g in got:
            app.redraw_image(app.load_piece(g), out)
        res["images"][name] = {"pieces": len(pieces), "pdus": len(got), "pixel_exact": bool(np.array_equal(out, img)),
                               "samples": int(x.size), "seconds": dt}
        total_samples += x.size
        total_time += dt
    res["all_pixel_exact"] = all(v["pixel_exact"] for v in res["images"].values())
    res["host_path_msamples_per_s"] = total_samples / total_time / 1e6
    print(json.dumps(res))


if __name__ == "__main__":
    main()
