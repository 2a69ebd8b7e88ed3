#!/usr/bin/env python3
"""Soak of the stream path: the same 4 M-sample piece of signal (frames + noise) pushed over and over; the frame
count per push must stay the same and the device memory in use must not grow.

    python tools/stream_soak.py [pushes=150]
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
from wifirx import capi, txgen  # noqa: E402


def free_bytes(hip):
    f, t = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value


def main():
    pushes = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    hip = C.CDLL("libamdhip64.so")
    psdu = txgen.make_psdus(64, 294, seed=9)
    tx = txgen.encode_psdus(psdu, 2)
    n = tx.samples.shape[1]
    slot = ((100 + n + 1000 + 63) // 64) * 64
    piece = np.tile(txgen.impair(tx.samples, 22.0, cfo=0.01, lead=100, total=slot, seed=4).reshape(-1), 12)
    frames_per_piece = 64 * 12
    rx = capi.WifiRx(max_sym=511)
    rx.set_param(capi.P_STREAM_BATCH, 1 << 20)
    counts, free = [], []
    t0 = time.perf_counter()
    for k in range(pushes):
        rx.push(piece)
        got = 0
        while True:
            r = rx.poll(cap=1024)
            if len(r["frames"]) == 0:
                break
            got += int(((r["frames"]["flags"] & capi.F_CRC_OK) != 0).sum())
        counts.append(got)
        if k % 10 == 0:
            free.append(free_bytes(hip))
    rx._check(capi.lib().wifirx_push(rx._h, None, 0, 0))
    tail = int(((rx.poll(cap=1024)["frames"]["flags"] & capi.F_CRC_OK) != 0).sum())
    dt = time.perf_counter() - t0
    st = rx.stats()
    rx.close()
    print(json.dumps({"pushes": pushes, "samples": int(piece.size) * pushes, "seconds": dt,
                      "msamples_per_s": piece.size * pushes / dt / 1e6,
                      "frames_sent": frames_per_piece * pushes, "frames_crc_ok": int(sum(counts)) + tail,
                      "crc_ok_per_push_min_max": [int(min(counts[1:])), int(max(counts[1:]))],
                      "device_free_bytes_first_last": [free[1] if len(free) > 1 else free[0], free[-1]],
                      "stats": {k: int(v) for k, v in st.items()}}))


if __name__ == "__main__":
    main()
