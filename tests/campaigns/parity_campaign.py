#!/usr/bin/env python3
"""A large randomised parity run: GPU (through the C ABI) against the oracle, bit for bit, on frames that differ
in rate, length, SNR, CFO, lead-in and channel (flat / multipath), for every equaliser.  The pytest parity cases
are hand-picked and small; this is the bulk check behind them.  Prints one JSON line.

    python tests/campaigns/parity_campaign.py [n_frames=20000] [seed=1] [long | plain]

`long`: slots of 45056 samples, PSDUs up to 1530 bytes at every rate (up to 511 OFDM symbols: the longest frames
decode_mac accepts, and the renormalisation of its 16-bit path metrics over 12 000+ trellis steps).
`plain`: the output set bench.py times (decisions + LLRs alone): the kernel's constellation loops with whole-line stores.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
sys.path.insert(0, ROOT)
from wifirx import capi, txgen  # noqa: E402


SNRS = (6.0, 12.0, 20.0, 28.0, 35.0)


def make_batch(n, slot_len, seed, max_len=1200, snrs=SNRS):
    rng = np.random.default_rng(seed)
    taps_set = np.load(os.path.join(ROOT, "tests", "golden", "sv_taps.npy"))
    iq = np.zeros((n, slot_len), np.complex64)
    # frames are generated in groups that share rate and length (the NumPy transmitter is vectorised over a group)
    k = 0
    while k < n:
        g = int(min(n - k, rng.integers(8, 64)))
        enc = int(rng.integers(0, 8))
        plen = int(rng.integers(30, max_len))
        while txgen.n_sym_for(plen, enc) * 80 + 400 + 260 > slot_len:
            plen = max(30, plen // 2)
        psdu = txgen.make_psdus(g, plen, seed=seed * 100003 + k)
        tx = txgen.encode_psdus(psdu, enc, seeds=[int(s) for s in rng.integers(1, 128, g)])
        for r in range(g):
            snr = float(rng.choice(list(snrs)))
            taps = taps_set[int(rng.integers(0, taps_set.shape[0]))][None, :] if rng.random() < 0.4 else None
            iq[k + r] = txgen.impair(tx.samples[r:r + 1], snr, cfo=float(rng.uniform(-0.04, 0.04)),
                                     lead=int(rng.integers(0, 250)), total=slot_len, seed=int(rng.integers(1 << 30)),
                                     taps=taps)[0]
        k += g
    return iq.reshape(-1)


def run(n=20000, seed=1, long_frames=False, snrs=SNRS, equalisers=(0, 1, 2, 3), decode_small_max=None, plain=False):
    """One campaign; returns the result dictionary (`all_bit_exact` is the verdict).  decode_small_max: forces decode_mac's
    kernel choice (0: the throughput decoder with 128 frames per wave for every batch; None: the library's default).
    plain: the output set bench.py times -- decisions + LLRs, no equalised points, no CSI: waves whose frames share a rate
    (the generator's groups of 8..63) then run the kernel's constellation loops with whole-line stores."""
    slot_len, max_sym, llr_bits = (45056, 511, 1) if long_frames else (8192, 96, 6)
    from oracle import oracle as orc
    t0 = time.perf_counter()
    iq = make_batch(n, slot_len, seed, 1531 if long_frames else 1200, snrs)
    res = {"frames": n, "slot_len": slot_len, "seed": seed, "snrs_db": list(snrs), "generate_s": time.perf_counter() - t0,
           "decode_small_max": decode_small_max, "outputs": "decisions + LLRs" if plain else "decisions + LLRs + equalised points + CSI",
           "equalisers": {}}
    threads = os.cpu_count() or 1
    old_env = os.environ.get("WIFIRX_DECODE_SMALL_MAX")
    if decode_small_max is not None:
        os.environ["WIFIRX_DECODE_SMALL_MAX"] = str(decode_small_max)
    try:
        for ce in equalisers:
            name = ("LS", "LMS", "COMB", "STA")[ce]
            rx = capi.WifiRx(max_sym=max_sym, llr_bits=llr_bits, want_carrier=not plain, chan_est=ce)
            r = rx.demod_batch(iq, slot_len, want_csi=not plain)
            d = rx.demod_batch(iq, slot_len, decode=True, psdu_stride=2048)
            rx.close()
            prm = orc.make_params(max_sym=max_sym, llr_bits=llr_bits, chan_est=ce)
            o = orc.demod_batch(iq, slot_len, prm, want_eq=True, want_csi=True, n_threads=threads)
            of = o["frames"].copy()
            opsdu = orc.decode_batch(of, o["idx"], prm, psdu_stride=2048, n_threads=threads)
            mism = {
                "frame_records": int((r["frames"] != o["frames"]).sum()),
                "decisions": int((r["idx"] != o["idx"]).sum()),
                "llr_values": int((r["llr"].view(np.uint32) != o["llr"].view(np.uint32)).sum()),
                "equalised_points": 0 if plain else int((r["carrier"].view(np.uint64) != o["eq"].view(np.uint64)).sum()),
                "csi_values": 0 if plain else int((r["csi"].view(np.uint64) != o["csi"].view(np.uint64)).sum()),
                "flags_after_decode": int((d["frames"]["flags"] != of["flags"]).sum()),
            }
            dec = (d["frames"]["flags"] & capi.F_DECODED) != 0
            col = np.arange(d["psdu"].shape[1])[None, :] < d["frames"]["psdu_len"][:, None].astype(np.int64)
            mism["psdu_bytes"] = int(((d["psdu"] != opsdu[:, :d["psdu"].shape[1]]) & col & dec[:, None]).sum())
            fl = d["frames"]["flags"]
            res["equalisers"][name] = {
                "mismatches": mism,
                "total_mismatches": int(sum(mism.values())),
                "detected": int(((fl & capi.F_DETECTED) != 0).sum()), "signal_ok": int(((fl & capi.F_SIGNAL) != 0).sum()),
                "complete": int(((fl & capi.F_COMPLETE) != 0).sum()), "crc_ok": int(((fl & capi.F_CRC_OK) != 0).sum()),
                "decisions_compared": int(r["idx"].size), "llr_values_compared": int(r["llr"].size),
            }
    finally:
        if decode_small_max is not None:
            if old_env is None:
                os.environ.pop("WIFIRX_DECODE_SMALL_MAX", None)
            else:
                os.environ["WIFIRX_DECODE_SMALL_MAX"] = old_env
    res["all_bit_exact"] = all(v["total_mismatches"] == 0 for v in res["equalisers"].values())
    res["seconds"] = time.perf_counter() - t0
    return res


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    long_frames = len(sys.argv) > 3 and sys.argv[3] == "long"
    plain = len(sys.argv) > 3 and sys.argv[3] == "plain"
    print(json.dumps(run(n, seed, long_frames, plain=plain)))


if __name__ == "__main__":
    main()
