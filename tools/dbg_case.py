import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cases import edge_cases
from wifirx import capi
from oracle import oracle as orc
name = sys.argv[1]
for c in edge_cases():
    if c[0] == name:
        _, iq, slot_len, max_sym, exp = c
rx = capi.WifiRx(max_sym=max_sym, llr_bits=6, want_carrier=True)
r = rx.demod_batch(iq, slot_len)
prm = orc.make_params(max_sym=max_sym, llr_bits=6)
o = orc.demod_batch(iq, slot_len, prm, want_eq=True)
print(r["frames"]); print(o["frames"])
d = (r["carrier"] != o["eq"])
for f in range(d.shape[0]):
    bad = np.argwhere(d[f])
    print("frame", f, "n mismatching eq values", len(bad), "first", bad[:5].tolist())
    if len(bad):
        s, k = bad[0]
        print("  gpu", r["carrier"][f, s, k], "cpu", o["eq"][f, s, k], "sym mismatch counts per symbol (first 20 nonzero):",
              [(int(i), int(v)) for i, v in enumerate(d[f].sum(1)) if v][:20])
