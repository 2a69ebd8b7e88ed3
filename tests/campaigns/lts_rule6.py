#!/usr/bin/env python3
"""Spec rule 6 under stress: does the two-stage LTS search (peak POSITIONS from the 8-bit integer correlation, VALUES
in float32 for those candidates only -- what the HIP kernels run) ever give another frame start / fine CFO than
sync_long's exhaustive search over all 320 float32 magnitudes (`gnu_radio/IRS_AP.py:269,282`)?

Frames at SNR 0..8 dB -- where peak ranking is fragile --, flat and Saleh-Valenzuela multipath (tests/golden/sv_taps.npy),
BPSK / QPSK / 16-QAM / 64-QAM, CFO +-0.04 rad/sample, random lead-in.  Three oracle evaluations of every frame:

  spec        SPEC arithmetic, rule 6 as the kernels run it (lts_search = 0)
  exhaustive  SPEC arithmetic, the float32 values of all 320 lags, their four largest (lts_search = 1)
  libm        the upstream-literal evaluation (double sums, hypotf, atan2f)

spec vs exhaustive isolates the candidate stage: everything else is bit-identical, so ANY differing record is the
candidate stage losing a peak that mattered.  spec vs libm adds the arithmetic distance (a 1-ulp change of a magnitude
can reorder two near-equal peaks in any implementation, upstream's own included).

    python tests/campaigns/lts_rule6.py [frames_per_group=1400] [seed=5] [threads=0: all] [threshold=0.56]

A lower sync_short threshold (0.35) lets the frames at 0..3 dB through to the LTS search, which the reference's 0.56
mostly rejects before it (|A|/P ~ S/(S+N)).

72 groups (9 SNRs x 4 constellations x 2 channels): 1400 per group = 100 800 frames.  Prints one JSON line; every
differing frame is listed with the four lags handed to the pair search (and their magnitudes) in both modes.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
sys.path.insert(0, ROOT)
from wifirx import txgen  # noqa: E402

SLOT = 2560
PSDU_LEN = 60
SNRS = (0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0)
ENCS = (0, 2, 4, 6)


def groups():
    for snr in SNRS:
        for enc in ENCS:
            for chan in ("flat", "sv"):
                yield snr, enc, chan


def make_group(n, snr, enc, chan, seed):
    """n slots of SLOT samples: one frame each, random CFO and lead-in, flat or multipath."""
    rng = np.random.default_rng(seed)
    psdu = txgen.make_psdus(n, PSDU_LEN, seed=seed)
    tx = txgen.encode_psdus(psdu, enc, seeds=rng.integers(1, 128, n))
    taps = None
    if chan == "sv":
        tset = np.load(os.path.join(ROOT, "tests", "golden", "sv_taps.npy"))
        taps = tset[rng.integers(0, tset.shape[0], n)]
    cfo = rng.uniform(-0.04, 0.04, n)
    # one lead-in per block of frames (impair places a whole batch at one offset)
    out = np.empty((n, SLOT), np.complex64)
    blk = 50
    for k in range(0, n, blk):
        e = min(n, k + blk)
        out[k:e] = txgen.impair(tx.samples[k:e], snr, cfo=cfo[k:e], lead=int(rng.integers(40, 300)), total=SLOT,
                                seed=int(rng.integers(1 << 30)), taps=None if taps is None else taps[k:e])
    return out.reshape(-1)


def lts_debug(orc, iq_slot, mode, lts_search, threshold=0.56):
    """The four lags (and magnitudes) the pair search of one slot sees."""
    import ctypes as C
    top = np.full(4, -2, np.int32)
    mag = np.zeros(4, np.float32)
    prm = orc.make_params(max_sym=8, math_mode=mode, lts_search=lts_search, threshold=threshold)
    prm.dbg_top4 = top.ctypes.data_as(C.c_void_p).value
    prm.dbg_mag4 = mag.ctypes.data_as(C.c_void_p).value
    orc.demod_batch(iq_slot, SLOT, prm, n_threads=1)
    return [int(v) for v in top], [float(v) for v in mag]


def run(per_group=1400, seed=5, threads=None, list_max=200, threshold=0.56):
    from oracle import oracle as orc
    threads = threads or os.cpu_count() or 1
    t0 = time.perf_counter()
    rows, listed = [], []
    tot = {"frames": 0, "detected": 0, "sync_spec": 0, "sync_exhaustive": 0, "exhaustive_differs": 0,
           "libm_comparable": 0, "libm_frame_start_differs": 0, "libm_sync_flag_differs": 0}
    worst_cfo = 0.0
    for gi, (snr, enc, chan) in enumerate(groups()):
        iq = make_group(per_group, snr, enc, chan, seed * 1000 + gi)
        res = {}
        for name, mode, ls in (("spec", orc.MATH_SPEC, 0), ("exhaustive", orc.MATH_SPEC, 1), ("libm", orc.MATH_LIBM, 0)):
            prm = orc.make_params(max_sym=8, math_mode=mode, lts_search=ls, threshold=threshold)
            res[name] = orc.demod_batch(iq, SLOT, prm, n_threads=threads)["frames"]
        a, b, c = res["spec"], res["exhaustive"], res["libm"]
        det = (a["flags"] & orc.F_DETECTED) != 0
        # spec vs exhaustive: the whole record must be equal (same arithmetic everywhere else)
        dif = det & (a != b)
        # spec vs libm: compare where both triggered at the same sample (detection arithmetic differs too)
        comp = det & ((c["flags"] & orc.F_DETECTED) != 0) & (a["trigger"] == c["trigger"])
        sa, sc = (a["flags"] & orc.F_SYNC) != 0, (c["flags"] & orc.F_SYNC) != 0
        fdif = comp & sa & sc & (a["frame_start"] != c["frame_start"])
        sdif = comp & (sa != sc)
        both = comp & sa & sc & ~fdif
        if both.any():
            worst_cfo = max(worst_cfo, float(np.abs(a["cfo_fine"][both] - c["cfo_fine"][both]).max()))
        row = {"snr_db": snr, "encoding": enc, "channel": chan, "frames": per_group, "detected": int(det.sum()),
               "sync_spec": int((det & sa).sum()), "sync_exhaustive": int((det & ((b["flags"] & orc.F_SYNC) != 0)).sum()),
               "exhaustive_differs": int(dif.sum()), "libm_comparable": int(comp.sum()),
               "libm_frame_start_differs": int(fdif.sum()), "libm_sync_flag_differs": int(sdif.sum())}
        rows.append(row)
        for k in ("frames", "detected", "sync_spec", "sync_exhaustive", "exhaustive_differs", "libm_comparable",
                  "libm_frame_start_differs", "libm_sync_flag_differs"):
            tot[k] += row[k]
        for kind, sel in (("exhaustive", dif), ("libm", fdif | sdif)):
            for k in np.nonzero(sel)[0]:
                if len(listed) >= list_max:
                    break
                slot = iq[k * SLOT:(k + 1) * SLOT]
                t_s, m_s = lts_debug(orc, slot, orc.MATH_SPEC, 0, threshold)
                t_o, m_o = lts_debug(orc, slot, orc.MATH_SPEC if kind == "exhaustive" else orc.MATH_LIBM,
                                     1 if kind == "exhaustive" else 0, threshold)
                o = b if kind == "exhaustive" else c
                listed.append({"vs": kind, "snr_db": snr, "encoding": enc, "channel": chan, "group_seed": seed * 1000 + gi,
                               "frame": int(k), "trigger": int(a["trigger"][k]),
                               "spec": {"flags": int(a["flags"][k]), "frame_start": int(a["frame_start"][k]),
                                        "cfo_fine": float(a["cfo_fine"][k]), "top4": t_s, "mag4": m_s},
                               "other": {"flags": int(o["flags"][k]), "frame_start": int(o["frame_start"][k]),
                                         "cfo_fine": float(o["cfo_fine"][k]), "top4": t_o, "mag4": m_o}})
    tot["exhaustive_differs_fraction"] = tot["exhaustive_differs"] / max(1, tot["detected"])
    tot["libm_differs_fraction"] = (tot["libm_frame_start_differs"] + tot["libm_sync_flag_differs"]) / max(1, tot["libm_comparable"])
    tot["largest_cfo_fine_distance_to_libm"] = worst_cfo
    return {"per_group": per_group, "seed": seed, "threshold": threshold, "threads": threads, "slot_len": SLOT, "psdu_len": PSDU_LEN,
            "totals": tot, "groups": rows, "differing_frames": listed, "seconds": time.perf_counter() - t0}


if __name__ == "__main__":
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 1400
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    th = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else None
    thr = float(sys.argv[4]) if len(sys.argv) > 4 else 0.56
    print(json.dumps(run(per, sd, th, threshold=thr)))
