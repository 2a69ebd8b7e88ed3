#!/usr/bin/env python3
"""The one place where this build deviates from upstream's sync_long ON PURPOSE, measured (VERDICT r03 "missing" 5 / item 6b).

When none of the pairs among the four strongest LTS correlation lags are 64 / 63 / 65 apart, upstream's sync_long
(`gnu_radio/IRS_AP.py:269,309`) keeps `d_frame_start` at its default -- `sync_length` = 320 -- and
`d_freq_offset` at the PREVIOUS frame's value, and copies the frame out anyway.  This build drops such a frame (DESIGN.md
section 3: a start that was not found and another frame's frequency offset are stream artefacts, not a decode).  How many
frames take that branch, and does upstream's guess ever produce a PDU?

Frames at 0..8 dB as in tests/campaigns/lts_rule6.py (BPSK .. 64-QAM, flat and Saleh-Valenzuela multipath, CFO +-0.04
rad/sample, random lead-in), the oracle's LIBM mode (the upstream-literal arithmetic), every batch twice:
  drop      no_pair_fallback = 0: the build's semantics
  upstream  no_pair_fallback = 1: such frames run again, in slot order, at offset 320 with the fine CFO of the last frame
            before them that found a pair
and decode_mac on both.  Reported per 100 000 detected frames: how many find no pair, and of those under upstream's rule
how many decode a SIGNAL field with good parity, run to their last symbol, and pass the FCS.

    python tests/campaigns/lts_no_pair.py [frames_per_group=1400] [seed=5] [threshold=0.56]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lts_rule6  # noqa: E402  (frame generator shared with the rule-6 campaign)


def run(per_group=1400, seed=5, threshold=0.56, threads=None):
    from oracle import oracle as orc
    threads = threads or os.cpu_count() or 1
    t0 = time.perf_counter()
    tot = {"frames": 0, "detected": 0, "lts_search_ran": 0, "no_pair": 0, "upstream_signal_ok": 0, "upstream_complete": 0,
           "upstream_crc_ok": 0, "upstream_signal_matches_tx": 0, "drop_crc_ok": 0, "upstream_total_crc_ok": 0}
    rows = []
    for gi, (snr, enc, chan) in enumerate(lts_rule6.groups()):
        iq = lts_rule6.make_group(per_group, snr, enc, chan, seed * 1000 + gi)
        res = {}
        for name, fb in (("drop", 0), ("upstream", 1)):
            prm = orc.make_params(max_sym=24, math_mode=orc.MATH_LIBM, threshold=threshold, no_pair_fallback=fb)
            o = orc.demod_batch(iq, lts_rule6.SLOT, prm, n_threads=threads)
            orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=64, n_threads=threads)
            res[name] = o["frames"]
        a, b = res["drop"], res["upstream"]
        det = (a["flags"] & orc.F_DETECTED) != 0
        ran = det & ((a["flags"] & 0x80) == 0)                      # not TRUNCATED before the search
        nop = ran & ((a["flags"] & orc.F_SYNC) == 0)
        row = {"snr_db": snr, "encoding": enc, "channel": chan, "frames": per_group, "detected": int(det.sum()),
               "lts_search_ran": int(ran.sum()), "no_pair": int(nop.sum()),
               "upstream_signal_ok": int((nop & ((b["flags"] & orc.F_SIGNAL) != 0)).sum()),
               "upstream_signal_matches_tx": int((nop & ((b["flags"] & orc.F_SIGNAL) != 0) & (b["encoding"] == enc) &
                                                  (b["psdu_len"] == lts_rule6.PSDU_LEN)).sum()),
               "upstream_complete": int((nop & ((b["flags"] & orc.F_COMPLETE) != 0)).sum()),
               "upstream_crc_ok": int((nop & ((b["flags"] & orc.F_CRC_OK) != 0)).sum()),
               "drop_crc_ok": int(((a["flags"] & orc.F_CRC_OK) != 0).sum()),
               "upstream_total_crc_ok": int(((b["flags"] & orc.F_CRC_OK) != 0).sum())}
        # frames that found a pair are untouched by the switch
        assert np.array_equal(a[~nop], b[~nop])
        rows.append(row)
        for k in tot:
            tot[k] += row[k]
    d = max(1, tot["detected"])
    tot["no_pair_per_100k_detected"] = 1e5 * tot["no_pair"] / d
    tot["upstream_crc_ok_per_100k_detected"] = 1e5 * tot["upstream_crc_ok"] / d
    return {"per_group": per_group, "seed": seed, "threshold": threshold, "slot_len": lts_rule6.SLOT, "psdu_len": lts_rule6.PSDU_LEN,
            "mode": "oracle LIBM (upstream-literal arithmetic)", "totals": tot, "groups": rows, "seconds": time.perf_counter() - t0}


if __name__ == "__main__":
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 1400
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.56
    print(json.dumps(run(per, sd, thr)))
