import sys, time
sys.path.insert(0,'/root/repo/gnuradio-wifi-imagetransfer_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from wifirx import capi, txgen
psdu = txgen.make_psdus(256, 294, seed=5)
tx = txgen.encode_psdus(psdu, 2)
n = tx.samples.shape[1]
slot = ((100 + n + 200 + 63)//64)*64
iq = txgen.impair(tx.samples, 25.0, cfo=0.0, lead=100, total=slot, seed=3).reshape(-1)
x = np.tile(iq, 16)          # 4096 frames
print("samples", x.size, "frames", 4096)
rx = capi.WifiRx(max_sym=511)
for it in range(3):
    t0=time.perf_counter()
    rx.push(x)
    t1=time.perf_counter()
    got = rx.poll(cap=8192)
    t2=time.perf_counter()
    print("push %.1f ms  poll %.1f ms  frames %d  -> %.0f Msample/s" % ((t1-t0)*1e3, (t2-t1)*1e3, len(got["frames"]), x.size/(t2-t0)/1e6))
rx.close()
