#!/usr/bin/env python3
"""Same-process A/B of decode_mac builds: python tools/ab_decode.py a.so b.so [--rounds 5] [--frames 1000000]
One demod (with planes) of the first library fills the records and planes; every library then decodes the same batch in turn
(wall time of wifirx_decode_batch + sync, ms; the first call of each is its allocation and is dropped).  Checks the good-FCS count."""
import argparse, ctypes as C, importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd")
sys.path.insert(0, PKG)

def load_capi(path, tag):
    os.environ["WIFIRX_LIB"] = os.path.abspath(path)
    spec = importlib.util.spec_from_file_location("capi_" + tag, os.path.join(PKG, "wifirx", "capi.py"))
    mod = importlib.util.module_from_spec(spec); sys.modules["capi_" + tag] = mod; spec.loader.exec_module(mod)
    return mod

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--frames", type=int, default=1000000); ap.add_argument("--snr", type=float, default=20.0)
    a = ap.parse_args()
    from wifirx import txgen
    tx = txgen.encode_psdus(txgen.make_psdus(64, 294, seed=5), 2)
    mods = [load_capi(p, str(i)) for i, p in enumerate(a.libs)]
    rxs = [m.WifiRx(max_sym=tx.n_sym, llr_bits=2) for m in mods]
    m0, rx0 = mods[0], rxs[0]
    n, slot, stride = a.frames, 4608, 320
    iq = rx0.alloc(n * slot * 8)
    rx0.synth_slots(tx.samples, iq.ptr, slot, n, 160, a.snr, 0.037, 99)
    b = dict(frames=rx0.alloc(n * 32), hbits=rx0.alloc(n * tx.n_sym * 48), psdu=rx0.alloc(n * stride))
    out = m0.Out(b["frames"].ptr, None, None, None, b["psdu"].ptr, stride, 1, None, None, b["hbits"].ptr)
    rx0._check(m0.lib().wifirx_demod_batch(rx0._h, iq.ptr, 1, slot, n, C.byref(out))); rx0.sync()
    fr0 = b["frames"].download(np.uint8, n * 32).copy()
    times = [[] for _ in rxs]
    for rnd in range(a.rounds + 1):
        row = []
        for k, (m, rx) in enumerate(zip(mods, rxs)):
            b["frames"].upload(fr0)                      # the records as the demod left them (decode sets flags in them)
            o = m.Out(b["frames"].ptr, None, None, None, b["psdu"].ptr, stride, 1, None, None, b["hbits"].ptr)
            rx0.sync(); t = time.perf_counter()
            rx._check(m.lib().wifirx_decode_batch(rx._h, n, C.byref(o))); rx.sync()
            ms = (time.perf_counter() - t) * 1e3
            if rnd: times[k].append(ms)
            row.append("%.3f" % ms)
        fr = b["frames"].download(np.uint8, n * 32).view(m0.FRAME_DTYPE)
        print("round %d: %s   crc ok %d" % (rnd, "  ".join(row), int(((fr["flags"] & m0.F_CRC_OK) != 0).sum())), flush=True)
    for p, t in zip(a.libs, times):
        print("%-28s min %.3f  median %.3f ms" % (os.path.basename(p), min(t), float(np.median(t))))

if __name__ == "__main__":
    main()
