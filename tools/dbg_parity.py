import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import make_slots
from wifirx import capi
from oracle import oracle as orc
enc = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iq, slot_len, tx = make_slots(16, enc, snr_db=22.0, seed=enc)
rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=6, want_carrier=True)
r = rx.demod_batch(iq, slot_len)
prm = orc.make_params(max_sym=tx.n_sym, llr_bits=6)
o = orc.demod_batch(iq, slot_len, prm, want_eq=True)
for name in r["frames"].dtype.names:
    print(name, "\n  gpu", r["frames"][name], "\n  cpu", o["frames"][name])
print("cfo_coarse bits equal:", (r["frames"]["cfo_coarse"].view(np.uint32) == o["frames"]["cfo_coarse"].view(np.uint32)))
print("cfo_fine bits equal:", (r["frames"]["cfo_fine"].view(np.uint32) == o["frames"]["cfo_fine"].view(np.uint32)))
d = np.abs(r["carrier"] - o["eq"])
print("max eq diff per frame", d.reshape(16, -1).max(axis=1))
print("idx mismatches per frame", (r["idx"] != o["idx"]).reshape(16, -1).sum(axis=1))
