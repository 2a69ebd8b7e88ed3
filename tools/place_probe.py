"""Does the demod kernel's time depend on WHERE its buffers lie?  Four sets of buffers (samples, records, decisions, LLRs) are allocated in one
process and the same launch is timed on each: on the config-3 geometry the sets differ by up to 6-8 % (6.08 / 6.12 / 6.21 / 6.48 ms in one
process, round 3), stable per set -- the two "states" consecutive processes show are placements.   python tools/place_probe.py [2|3]"""
import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
from wifirx import capi, txgen
geo = int(sys.argv[1]) if len(sys.argv) > 1 else 3
enc, slot, snr = {2: (2, 4608, 20.0), 3: (7, 1472, 30.0)}[geo]
tx = txgen.encode_psdus(txgen.make_psdus(64, 294, seed=5), enc)
nb = txgen.RATE_TABLE[enc][0]
rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=nb)
n = 1000000
C = capi.C
def make_set():
    iq = rx.alloc(n * slot * 8)
    rx.synth_slots(tx.samples, iq.ptr, slot, n, 160, snr, 0.037, 99)
    fr = rx.alloc(n * 32); idx = rx.alloc(n * tx.n_sym * 48); llr = rx.alloc(n * tx.n_sym * 48 * nb * 4)
    rx.sync()
    return iq, fr, idx, llr
def timeit(s):
    iq, fr, idx, llr = s
    out = capi.Out(fr.ptr, idx.ptr, llr.ptr, None, None, 0, 1, None)
    ms = C.c_float(0); best = 1e9
    for _ in range(4):
        rx._check(capi.lib().wifirx_time_demod(rx._h, iq.ptr, slot, n, C.byref(out), 1, C.byref(ms)))
        best = min(best, ms.value)
    return best
sets = []
for k in range(4):
    s = make_set(); sets.append(s)
    print("set %d: iq %#x llr %#x  -> %.3f ms" % (k, s[0].ptr, s[3].ptr, timeit(s)), flush=True)
for k, s in enumerate(sets):
    print("again set %d -> %.3f ms" % (k, timeit(s)), flush=True)
# free set 0 and 1, reallocate
for s in sets[:2]:
    for b in s: b.free()
s = make_set(); print("after freeing two sets, new set: iq %#x -> %.3f ms" % (s[0].ptr, timeit(s)), flush=True)
