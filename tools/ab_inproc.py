#!/usr/bin/env python3
"""Same-PROCESS A/B of builds of libwifirx.so: every library is loaded into one process, the same device batch goes
through each in turn, many alternations.  Consecutive processes on one box differ by up to 3 % in the demod kernel's
time (round 3: two states of a box), which hides changes of a per cent from tools/ab_bench.sh; inside one process the
spread is ~0.3 %.

    python tools/ab_inproc.py a.so b.so [c.so ...] [--rounds 8] [--frames 1000000] [--geometry 2|3|1] [--eq 0..3] [--planes]

Prints the kernel time (HIP events in the library, ms per launch) of every round and, per library, min / median.
geometry: 2 = config 2 (QPSK 1/2, slot 4608), 3 = config-3 geometry (64-QAM 3/4, slot 1472), 1 = config 1 (BPSK 1/2, slot 8576)."""
import argparse
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd")
sys.path.insert(0, PKG)


def load_capi(path, tag):
    """wifirx.capi once more under another module name, bound to the library at `path`"""
    os.environ["WIFIRX_LIB"] = os.path.abspath(path)
    spec = importlib.util.spec_from_file_location("capi_" + tag, os.path.join(PKG, "wifirx", "capi.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["capi_" + tag] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--frames", type=int, default=1000000)
    ap.add_argument("--geometry", type=int, default=2)
    ap.add_argument("--eq", type=int, default=0)
    ap.add_argument("--planes", action="store_true")
    ap.add_argument("--carrier", action="store_true", help="the equalised points too (want_carrier = 1: the reference's own output set)")
    ap.add_argument("--planes-only", action="store_true", help="bit planes + records alone (what the stream path asks for)")
    ap.add_argument("--csi", action="store_true", help="LLRs weighted by |H|^2 (WIFIRX_P_LLR_CSI)")
    ap.add_argument("--stats", action="store_true", help="the probe's moments (sym_stats)")
    ap.add_argument("--sv", action="store_true", help="multipath on the templates (tests/golden/sv_taps.npy: bench.py's config3_geometry)")
    ap.add_argument("--snr", type=float, default=None)
    ap.add_argument("--lead", type=int, default=160, help="noise samples in front of every frame (160 = ten 128-byte lines: frames start on a line)")
    a = ap.parse_args()
    from wifirx import txgen
    enc, slot, snr = {2: (2, 4608, 20.0), 3: (7, 1472, 30.0), 1: (0, 8576, 20.0)}[a.geometry]
    tx = txgen.encode_psdus(txgen.make_psdus(256 if a.sv else 64, 294, seed=5), enc)
    samples = tx.samples
    if a.sv:
        taps = np.load(os.path.join(ROOT, "tests", "golden", "sv_taps.npy"))[:256]
        samples = txgen.impair(tx.samples, None, cfo=0.0, lead=0, total=tx.samples.shape[1] + taps.shape[1], taps=taps)
    if a.snr is not None:
        snr = a.snr
    n_bpsc = txgen.RATE_TABLE[enc][0]
    mods = [load_capi(p, str(i)) for i, p in enumerate(a.libs)]
    rxs = [m.WifiRx(max_sym=tx.n_sym, llr_bits=n_bpsc, chan_est=a.eq, want_carrier=a.carrier) for m in mods]
    if a.csi:
        for m, rx in zip(mods, rxs):
            rx.set_param(m.P_LLR_CSI, 1)
    m0, rx0 = mods[0], rxs[0]
    n = a.frames
    iq = rx0.alloc(n * slot * 8)
    rx0.synth_slots(samples, iq.ptr, slot, n, a.lead, snr, 0.037, 99)
    planes = a.planes or a.planes_only
    dev = rx0.alloc_out(n, want_hbits=planes, want_stats=a.stats)
    rx0.sync()
    C = m0.C
    times = [[] for _ in rxs]
    for rnd in range(a.rounds):
        row = []
        for k, (m, rx) in enumerate(zip(mods, rxs)):
            out = m.Out(dev["frames"].ptr, None if a.planes_only else dev["idx"].ptr, None if a.planes_only else dev["llr"].ptr,
                        dev["carrier"].ptr if a.carrier else None, None, 0, 1, None,
                        dev["sym_stats"].ptr if a.stats else None, dev["hbits"].ptr if planes else None)
            ms = C.c_float(0)
            best = 1e9
            for _ in range(3):
                rx._check(m.lib().wifirx_time_demod(rx._h, iq.ptr, slot, n, C.byref(out), 1, C.byref(ms)))
                best = min(best, ms.value)
            times[k].append(best)
            row.append("%.3f" % best)
        print("round %d: %s" % (rnd, "  ".join(row)), flush=True)
    for p, t in zip(a.libs, times):
        print("%-28s min %.3f  median %.3f ms" % (os.path.basename(p), min(t), float(np.median(t))))


if __name__ == "__main__":
    main()
