#!/usr/bin/env python3
"""How far is the numerics spec (what the HIP kernels equal bit for bit) from the upstream-literal evaluation of
the same chain (libm, C99 complex arithmetic, double running sums: the oracle's LIBM mode)?  Both oracle modes on
the random frames of tests/campaigns/parity_campaign.py; CPU only.

    python tests/campaigns/spec_vs_libm.py [n_frames=3000] [seed=11]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_campaign as pc  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    slot = 8192
    iq = pc.make_batch(n, slot, seed)
    res = {}
    for mode in (orc.MATH_SPEC, orc.MATH_LIBM):
        prm = orc.make_params(max_sym=96, llr_bits=6, math_mode=mode)
        o = orc.demod_batch(iq, slot, prm, want_eq=True, n_threads=os.cpu_count() or 1)
        orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=2048, n_threads=os.cpu_count() or 1)
        res[mode] = o
    a, b = res[orc.MATH_SPEC], res[orc.MATH_LIBM]
    same = ((a["frames"]["trigger"] == b["frames"]["trigger"]) & (a["frames"]["frame_start"] == b["frames"]["frame_start"]) &
            (a["frames"]["flags"] == b["frames"]["flags"]))
    both = same & ((a["frames"]["flags"] & orc.F_COMPLETE) != 0)
    tot = diff = 0
    worst = 0.0
    for k in np.nonzero(both)[0]:
        m = int(a["frames"]["n_sym_out"][k])
        tot += m * 48
        diff += int((a["idx"][k, :m] != b["idx"][k, :m]).sum())
        worst = max(worst, float(np.abs(a["eq"][k, :m] - b["eq"][k, :m]).max()))
    print(json.dumps({"frames": n, "seed": seed, "records_equal": int(same.sum()), "decisions_compared": tot,
                      "decisions_differing": diff, "largest_equalised_point_difference": worst,
                      "crc_ok_spec": int(((a["frames"]["flags"] & orc.F_CRC_OK) != 0).sum()),
                      "crc_ok_libm": int(((b["frames"]["flags"] & orc.F_CRC_OK) != 0).sum())}))


if __name__ == "__main__":
    main()
