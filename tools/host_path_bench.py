#!/usr/bin/env python3
"""Throughput of the drop-in path: wifi_phy_rx.work() fed with host chunks the way a GNU Radio scheduler feeds it.

    python tools/host_path_bench.py [frames=6000]

Two streams: config-2-like (QPSK-1/2, 294 B, one frame per 4608 samples) and config-1-like (BPSK-1/2, 294 B,
packet_pad2 gaps: 9420 samples per frame); work() chunk sizes 8192 / 32768 / 131072 items; batch sizes 2^20 / 2^22.
Prints one JSON line per case: Gsample/s through work() incl. PDU construction and a list-append consumer."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
from wifirx import block, grshim, txgen  # noqa: E402


def make_stream(enc, n_frames, period, lead, snr_db=20.0, seed=3):
    tx = txgen.encode_psdus(txgen.make_psdus(64, 294, seed=seed), enc)
    flen = tx.samples.shape[1]
    assert lead + flen <= period
    g = np.float32(np.sqrt(10 ** (snr_db / 10)))
    x = np.zeros((n_frames, period), np.complex64)
    x[:, lead:lead + flen] = tx.samples[np.arange(n_frames) % 64] * g
    rng = np.random.default_rng(seed)
    x = x.reshape(-1)
    x += ((rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size)) * np.sqrt(0.5)).astype(np.complex64)
    return x, tx.n_sym


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    for name, enc, period, lead in (("config-2-like", 2, 4608, 160), ("config-1-like", 0, 9420, 100)):
        x, n_sym = make_stream(enc, n_frames, period, lead)
        for batch in (1 << 20, 1 << 22):
            for chunk in (8192, 32768, 131072):
                blk = block.wifi_phy_rx(bandwidth=20e6, frequency=5.89e9, max_sym=n_sym, publish_carrier=False,
                                        batch_samples=batch)
                got = []
                grshim.msg_connect(blk, "mac_out", grshim.sink_block(got.append), "in")
                grshim.run_stream(blk, x[:period * 300], chunk=chunk)             # warm-up
                n0 = len(got)
                best = 0.0
                for _ in range(2):
                    t = time.perf_counter()
                    grshim.run_stream(blk, x, chunk=chunk)
                    dt = time.perf_counter() - t
                    best = max(best, x.size / dt / 1e9)
                print(json.dumps({"stream": name, "batch_samples": batch, "work_chunk": chunk, "gsamples_per_s": round(best, 3),
                                  "pdus": (len(got) - n0) // 2, "frames": n_frames}), flush=True)
                blk.close()


if __name__ == "__main__":
    main()
