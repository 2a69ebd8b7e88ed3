import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd")); sys.path.insert(0, ROOT)
from wifirx import capi, txgen
from oracle import oracle as orc
snr = int(sys.argv[1]); n_frames = 100000; n_check = 2000
taps = np.load(os.path.join(ROOT, "tests", "golden", "sv_taps.npy")); n_t = taps.shape[0]
psdu = txgen.make_psdus(n_t, 294, seed=31); tx = txgen.encode_psdus(psdu, 7)
faded = txgen.impair(tx.samples, None, cfo=0.0, lead=0, total=tx.samples.shape[1] + 8, taps=taps)
slot_len, lead = 1472, 160
rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0, want_carrier=True)
slots = rx.alloc(n_frames * slot_len * 8); dev = rx.alloc_out(n_check)
rx.synth_slots(faded, slots.ptr, slot_len, n_frames, lead, float(snr), 0.037, 1000 + snr)
rx.demod_batch_dev(slots.ptr, slot_len, n_check, dev); rx.sync()
r = rx.download_out(dev, n_check)
x = slots.download(np.complex64, n_check * slot_len)
prm = orc.make_params(max_sym=tx.n_sym)
o = orc.demod_batch(x, slot_len, prm, want_eq=True)
bad = np.nonzero((o["idx"] != r["idx"]).reshape(n_check, -1).any(1) | (o["frames"] != r["frames"]))[0]
print("bad frames", bad)
for k in bad[:4]:
    print("gpu", r["frames"][k]); print("cpu", o["frames"][k])
    d = np.abs(r["carrier"][k] - o["eq"][k]); print("eq max diff per symbol", d.max(1))
    np.save(os.path.join(ROOT, "gpurun_out", "bad_slot_%d_%d.npy" % (snr, k)), x[k * slot_len:(k + 1) * slot_len])
