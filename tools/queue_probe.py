#!/usr/bin/env python3
"""Does the time of the demod kernel depend on WHICH HIP stream (hardware queue) of the process launches it?
Consecutive processes on one box alternate between ~11.75 and ~12.1 ms for the same launch (round 3); this probe creates
several handles (each with a stream of its own) in ONE process and times the same batch through each of them.
    python tools/queue_probe.py [n_handles=6] [frames=1000000]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
from wifirx import capi, txgen

def main():
    n_h = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    slot = 4608
    tx = txgen.encode_psdus(txgen.make_psdus(64, 294, seed=5), 2)
    hs = [capi.WifiRx(max_sym=tx.n_sym, llr_bits=2) for _ in range(n_h)]
    rx0 = hs[0]
    iq = rx0.alloc(n * slot * 8)
    rx0.synth_slots(tx.samples, iq.ptr, slot, n, 160, 20.0, 0.037, 99)
    dev = rx0.alloc_out(n)
    rx0.sync()
    C = capi.C
    out = capi.Out(dev["frames"].ptr, dev["idx"].ptr, dev["llr"].ptr, None, None, 0, 1, None)
    for rep in range(3):
        row = []
        for h in hs:
            ms = C.c_float(0)
            best = 1e9
            for _ in range(4):
                h._check(capi.lib().wifirx_time_demod(h._h, iq.ptr, slot, n, C.byref(out), 1, C.byref(ms)))
                best = min(best, ms.value)
            row.append(round(best, 3))
        print("pass %d: kernel ms per handle / stream: %s" % (rep, row), flush=True)

if __name__ == "__main__":
    main()
