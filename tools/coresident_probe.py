#!/usr/bin/env python3
"""Can decode_mac of batch k run UNDER the demod kernel of batch k + 1 ("so the kernels actually co-reside")?

Alone the demod kernel fills a CU (4 waves of 128 registers per SIMD, 16 x 9.9 kB of LDS) and decode_q_kernel takes 2 waves of 233
registers per SIMD: whichever starts first shuts the other out, and two streams run the kernels one after the other (bench.py
pdu_leg.pipelined).  Here decode_mac is capped to ONE wave per SIMD (1 024 waves: its scratch budget set to 1 024 slices) on a
high-priority stream: 233 registers + 37 kB of LDS per CU-quarter leave room for two demod waves per SIMD (256 registers, 8 x 9.9 kB),
which the demod kernel of the next batch fills from a second, low-priority stream.  Prints ms per batch: the two kernels back to back
on one stream (serial), on two streams as they are (plain), and capped + prioritised (coresident).

    python tools/coresident_probe.py [--frames 1000000] [--batches 6]"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1000000)
    ap.add_argument("--batches", type=int, default=6)
    a = ap.parse_args()
    from wifirx import capi, txgen
    lib = capi.lib()
    n, slot, stride = a.frames, 4608, 320
    tx = txgen.encode_psdus(txgen.make_psdus(64, 294, seed=5), 2)
    n_sym = tx.n_sym

    def handle(prio=None, budget=None):
        for k, v in (("WIFIRX_STREAM_PRIORITY", prio), ("WIFIRX_TEST_DECODE_BUDGET", budget)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        return capi.WifiRx(max_sym=n_sym, llr_bits=2)

    rx0 = handle()
    iq = rx0.alloc(n * slot * 8)
    rx0.synth_slots(tx.samples, iq.ptr, slot, n, 160, 20.0, 0.037, 99)
    sets = []
    for _ in range(2):
        b = dict(frames=rx0.alloc(n * 32), idx=rx0.alloc(n * n_sym * 48), llr=rx0.alloc(n * n_sym * 96 * 4),
                 hbits=rx0.alloc(n * n_sym * 48), psdu=rx0.alloc(n * stride))
        sets.append(b)
    o_dem = [capi.Out(b["frames"].ptr, b["idx"].ptr, b["llr"].ptr, None, b["psdu"].ptr, stride, 1, None, None, b["hbits"].ptr) for b in sets]
    o_dec = [capi.Out(b["frames"].ptr, None, None, None, b["psdu"].ptr, stride, 1, None, None, b["hbits"].ptr) for b in sets]
    rx0.sync()

    def demod(rx, k):
        rx._check(lib.wifirx_demod_batch(rx._h, iq.ptr, 1, slot, n, C.byref(o_dem[k % 2])))

    def decode(rx, k):
        rx._check(lib.wifirx_decode_batch(rx._h, n, C.byref(o_dec[k % 2])))

    def crc_ok(k):
        fr = sets[k % 2]["frames"].download(np.uint8, n * 32).view(capi.FRAME_DTYPE)
        return int(((fr["flags"] & capi.F_CRC_OK) != 0).sum())

    # serial: one stream
    demod(rx0, 0); decode(rx0, 0); rx0.sync()
    t = time.perf_counter()
    for k in range(a.batches):
        demod(rx0, k); decode(rx0, k)
    rx0.sync()
    serial = (time.perf_counter() - t) / a.batches * 1e3
    print("serial (one stream)           %.2f ms per batch   crc ok %d" % (serial, crc_ok(a.batches - 1)), flush=True)

    def pipelined(rx_dem, rx_dec, label):
        demod(rx_dem, 0); rx_dem.sync(); decode(rx_dec, 0); rx_dec.sync()         # allocations
        filled, free_ = threading.Semaphore(0), threading.Semaphore(2)
        err = []

        def producer():
            try:
                for k in range(a.batches):
                    free_.acquire(); demod(rx_dem, k); rx_dem.sync(); filled.release()
            except Exception as e:
                err.append(e); filled.release()

        def consumer():
            try:
                for k in range(a.batches):
                    filled.acquire(); decode(rx_dec, k); rx_dec.sync(); free_.release()
            except Exception as e:
                err.append(e); free_.release()

        ta, tb = threading.Thread(target=producer), threading.Thread(target=consumer)
        t0 = time.perf_counter()
        ta.start(); tb.start(); ta.join(); tb.join()
        ms = (time.perf_counter() - t0) / a.batches * 1e3
        print("%-29s %.2f ms per batch   crc ok %d  %s" % (label, ms, crc_ok(a.batches - 1), err or ""), flush=True)
        return ms

    rx1 = handle()
    pipelined(rx0, rx1, "two streams, plain")
    rx1.close()
    # decode capped to one wave per SIMD (1 024 slices of its scratch), on a high-priority stream; demod on a low-priority one
    n_steps = 8 * 294 + 16 + 6
    n_steps = (n_steps + 47) // 48 * 48
    slice_b = n_steps * 64 * 32 + (n_steps // 32 + 2) * 256 * 4 + (n_steps // 96 + 2) * 64 * 4
    rx_dec = handle(prio="high", budget=1024 * slice_b + 4096)
    rx_dem = handle(prio="low")
    pipelined(rx_dem, rx_dec, "capped + prioritised")
    rx_dem.close(); rx_dec.close()
    rx_dec = handle(prio="high", budget=1024 * slice_b + 4096)
    pipelined(rx0, rx_dec, "capped, decode high priority")
    rx_dec.close()
    rx0.close()


if __name__ == "__main__":
    main()
