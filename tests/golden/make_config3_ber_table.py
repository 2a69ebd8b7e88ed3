#!/usr/bin/env python3
"""tests/golden/config3_ber_table.json from the ORACLE, on the CPU (no GPU involved): BASELINE.json config 3 -- 64-QAM 3/4,
PSDU 294 B (11 symbols), the 8-tap channel built from the reference's utils/SV_channel.py draws
(tests/golden/sv_taps.npy; tap construction: SURVEY.md 8(d), a build decision), LS equaliser, CFO uniform in +-20 ppm of
5.89 GHz at 20 MHz, SNR 5, 10, ..., 30 dB.  Frames come from wifirx/txgen.py (NumPy transmitter + the reference's
loop-back channel law, gnu_radio/IRS_tranceiver.py:282-294), the receiver is oracle/wifirx_oracle.c in SPEC mode.

    python tests/golden/make_config3_ber_table.py [frames_per_point=30000]

The GPU test (tests/test_gpu_configs.py::test_config3_reduced_ber_sweep) draws its own noise on the device and must
land within a confidence interval of this table: then its BER / FER asserts are oracle parity, not a comparison of the
GPU with itself (round 1's GPU-made table is kept beside it as config3_ber_table_gpu_r01.json, for the record).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
sys.path.insert(0, ROOT)
from wifirx import txgen  # noqa: E402  (pure NumPy; nothing of the HIP library is called here)
from oracle import oracle as orc  # noqa: E402

CFO_20PPM = 2 * np.pi * 20e-6 * 5.89e9 / 20e6


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    taps = np.load(os.path.join(ROOT, "tests", "golden", "sv_taps.npy"))
    n_t = taps.shape[0]
    psdu = txgen.make_psdus(n_t, 294, seed=31)
    tx = txgen.encode_psdus(psdu, 7)
    faded = txgen.impair(tx.samples, None, cfo=0.0, lead=0, total=tx.samples.shape[1] + 8, taps=taps)
    slot_len, lead = 1472, 160
    bits_tx = ((tx.data_idx[..., None] >> np.arange(6)) & 1).astype(np.uint8)
    prm = orc.make_params(max_sym=tx.n_sym)
    threads = os.cpu_count() or 1
    tmpl = np.arange(n) % n_t
    points = []
    for snr in (5, 10, 15, 20, 25, 30):
        rng = np.random.default_rng(7000 + snr)
        x = txgen.impair(faded[tmpl], float(snr), cfo=rng.uniform(-CFO_20PPM, CFO_20PPM, n), lead=lead, total=slot_len,
                         seed=9000 + snr).reshape(-1)
        o = orc.demod_batch(x, slot_len, prm, n_threads=threads)
        opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=320, n_threads=threads)
        fr = o["frames"]
        good = ((fr["flags"] & orc.F_COMPLETE) != 0) & (fr["encoding"] == 7) & (fr["psdu_len"] == 294)
        bits_rx = ((o["idx"][good][..., None] >> np.arange(6)) & 1).astype(np.uint8)
        per_frame = (bits_rx != bits_tx[tmpl[good]]).reshape(int(good.sum()), -1).mean(axis=1)
        ok = ((fr["flags"] & orc.F_CRC_OK) != 0) & (opsdu[:, :294] == psdu[tmpl]).all(axis=1)
        points.append({"snr_db": snr, "frames": n, "detected_and_signal_ok": float(good.mean()),
                       "coded_ber": float(per_frame.mean()), "coded_ber_se": float(per_frame.std() / np.sqrt(len(per_frame))),
                       "fer": float(1.0 - ok.mean())})
        print(points[-1], file=sys.stderr)
    out = {"provenance": "tests/golden/make_config3_ber_table.py %d: the ORACLE (oracle/wifirx_oracle.c, SPEC mode) on the CPU over frames "
                         "of wifirx/txgen.py -- 64-QAM 3/4, PSDU 294 B, tests/golden/sv_taps.npy, LS equaliser, CFO uniform in "
                         "+-20 ppm, NumPy noise; no GPU output is part of this table (round 1's GPU-made table: "
                         "config3_ber_table_gpu_r01.json)" % n,
           "points": points}
    with open(os.path.join(ROOT, "tests", "golden", "config3_ber_table.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
