#!/usr/bin/env python3
"""demod kernel time on one geometry: python tools/geo_probe.py <encoding> <psdu_len> <slot_len> [frames]  (WIFIRX_LIB selects the build)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gnuradio-wifi-imagetransfer_amd"))
from wifirx import capi, txgen
enc, plen, slot_len = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n_frames = int(sys.argv[4]) if len(sys.argv) > 4 else 1000000
n_sym, n_bpsc = txgen.n_sym_for(plen, enc), txgen.RATE_TABLE[enc][0]
tx = txgen.encode_psdus(txgen.make_psdus(256, plen, seed=5), enc)
rx = capi.WifiRx(max_sym=n_sym, llr_bits=n_bpsc)
slots = rx.alloc(n_frames * slot_len * 8)
dev = rx.alloc_out(n_frames, psdu_stride=320)
rx.synth_slots(tx.samples, slots.ptr, slot_len, n_frames, 160, 25.0, 0.037, 77)
rx.time_demod(slots.ptr, slot_len, n_frames, dev, iters=2)
ms = min(rx.time_demod(slots.ptr, slot_len, n_frames, dev, iters=3) for _ in range(3))
fr = dev["frames"].download(capi.FRAME_DTYPE, n_frames)
bpf = 8 * slot_len + 48 * n_sym * (1 + 4 * n_bpsc) + 32
print(os.environ.get("WIFIRX_LIB", "-").split("/")[-1], "ms", round(ms, 3), "frac", round(bpf * n_frames / (ms * 1e-3) / 8e12, 4),
      "complete", int(((fr["flags"] & capi.F_COMPLETE) != 0).sum()))
