#!/usr/bin/env python3
"""BASELINE.json config 3: 64-QAM-3/4 frames (PSDU 294 B, 11 symbols) through the multipath taps derived from
the reference's utils/SV_channel.py (tests/golden/sv_taps.npy; the tap construction is a build decision,
SURVEY.md 8(d)), LS equaliser (argv[2]: 1 LMS, 2 COMB, 3 STA), SNR 5..30 dB.  Per point: N frames on the GPU, coded-bit BER of the hard
decisions against the transmitted interleaved bits, frame error rate after decode_mac, and -- on a subset --
the decision mismatch count against the oracle (must be 0).  One JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
sys.path.insert(0, ROOT)
from wifirx import capi, txgen  # noqa: E402


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    chan_est = int(sys.argv[2]) if len(sys.argv) > 2 else 0            # 0 LS (config 3 proper), 1 LMS, 2 COMB, 3 STA
    n_check = 2000
    taps = np.load(os.path.join(ROOT, "tests", "golden", "sv_taps.npy"))          # [1024, 8]
    n_t = taps.shape[0]
    psdu = txgen.make_psdus(n_t, 294, seed=31)
    tx = txgen.encode_psdus(psdu, 7)
    # every template frame goes through its own tap set on the host; noise and CFO are added on the GPU
    faded = txgen.impair(tx.samples, None, cfo=0.0, lead=0, total=tx.samples.shape[1] + 8, taps=taps)
    slot_len, lead = 1472, 160
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0, chan_est=chan_est)
    slots = rx.alloc(n_frames * slot_len * 8)
    dev = rx.alloc_out(n_frames, psdu_stride=320)
    bits_tx = ((tx.data_idx[..., None] >> np.arange(6)) & 1).astype(np.uint8)
    from oracle import oracle as orc
    prm = orc.make_params(max_sym=tx.n_sym, chan_est=chan_est)
    points = []
    for snr in range(5, 31):
        rx.synth_slots(faded, slots.ptr, slot_len, n_frames, lead, float(snr), 0.037, 1000 + snr)
        rx.demod_batch_dev(slots.ptr, slot_len, n_frames, dev)
        rx.decode_batch_dev(n_frames, dev)
        rx.sync()
        r = rx.download_out(dev, n_frames)
        fl = r["frames"]["flags"]
        comp = (fl & capi.F_COMPLETE) != 0
        ok = (fl & capi.F_CRC_OK) != 0
        tmpl = np.arange(n_frames) % n_t
        good = comp & (r["frames"]["encoding"] == 7) & (r["frames"]["psdu_len"] == 294)
        bits_rx = ((r["idx"][good][..., None] >> np.arange(6)) & 1).astype(np.uint8)
        ber = float((bits_rx != bits_tx[tmpl[good]]).mean()) if good.any() else None
        right = ok & (r["psdu"][:, :294] == psdu[tmpl]).all(axis=1)
        x = slots.download(np.complex64, n_check * slot_len)
        o = orc.demod_batch(x, slot_len, prm)
        gfr = r["frames"][:n_check].copy()
        gfr["flags"] &= ~np.uint32(capi.F_DECODED | capi.F_CRC_OK)
        # the kernel writes only the symbols a frame delivers: compare those (the output buffer is reused across SNRs)
        written = np.arange(tx.n_sym)[None, :] < o["frames"]["n_sym_out"][:, None]
        mism = int(((o["idx"] != r["idx"][:n_check]) & written[..., None]).sum() + (o["frames"] != gfr).sum())
        points.append({"snr_db": snr, "frames": n_frames, "detected_and_signal_ok": float(good.mean()),
                       "coded_ber": ber, "fer": float(1.0 - right.mean()), "oracle_mismatches_on_%d" % n_check: mism})
    print(json.dumps({"config": "64-QAM 3/4, PSDU 294 B, SV-derived 8-tap Rician channel (K=10), %s equaliser, CFO +-20 ppm"
                                % ("LS", "LMS", "COMB", "STA")[chan_est],
                      "points": points}))


if __name__ == "__main__":
    main()
