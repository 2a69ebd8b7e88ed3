"""The deviation "no LTS pair => frame dropped"  against upstream's rule, which decodes such a frame at
offset 320 with the PREVIOUS frame's fine CFO (`gnu_radio/IRS_AP.py:269,309`): the oracle's measurement
switch `no_pair_fallback`, and a reduced run of tests/campaigns/lts_no_pair.py (the full runs: profiles/r04_lts_no_pair_*.json)."""
import os
import sys

import numpy as np

from wifirx import txgen

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "campaigns"))


def test_fallback_switch_decodes_at_320_with_the_previous_fine_cfo(orc):
    """slot 0: a good frame with a carrier offset; slot 1: a pure tone -- sync_short triggers on it (|A|/P = 1), its LTS
    correlation has the same magnitude at every lag, so the four 'strongest' lags are neighbours and no pair is 63..65 apart"""
    tx = txgen.encode_psdus(txgen.make_psdus(1, 60, seed=3), 2)
    S = 2560
    good = txgen.impair(tx.samples, 25.0, cfo=np.array([0.021]), lead=160, total=S, seed=9)[0]
    n = np.arange(S)
    tone = (3.0 * np.exp(1j * 0.3 * n)).astype(np.complex64)
    tone[:100] = 0
    iq = np.concatenate([good, tone]).astype(np.complex64)
    for mode in (orc.MATH_LIBM, orc.MATH_SPEC):
        a = orc.demod_batch(iq, S, orc.make_params(max_sym=tx.n_sym, math_mode=mode), n_threads=1)["frames"]
        b = orc.demod_batch(iq, S, orc.make_params(max_sym=tx.n_sym, math_mode=mode, no_pair_fallback=1), n_threads=1)["frames"]
        assert (a["flags"][0] & orc.F_COMPLETE) and np.array_equal(a[0], b[0])          # a frame with a pair is untouched
        assert (a["flags"][1] & orc.F_DETECTED) and not (a["flags"][1] & orc.F_SYNC)     # the build: dropped
        assert (b["flags"][1] & orc.F_SYNC) and b["frame_start"][1] == 320               # upstream: default start ...
        assert b["cfo_fine"][1] == a["cfo_fine"][0] != 0.0                               # ... and the previous frame's offset
        assert b["trigger"][1] == a["trigger"][1] and b["cfo_coarse"][1] == a["cfo_coarse"][1]


def test_reduced_no_pair_campaign():
    import lts_no_pair
    r = lts_no_pair.run(per_group=24, seed=11, threshold=0.35, threads=os.cpu_count() or 1)
    t = r["totals"]
    assert t["frames"] == 72 * 24 and t["detected"] > 0.8 * t["frames"]
    assert t["upstream_total_crc_ok"] >= t["drop_crc_ok"] > 0          # upstream's rule can only add PDUs ...
    assert t["upstream_crc_ok"] == t["upstream_total_crc_ok"] - t["drop_crc_ok"]
    assert t["upstream_complete"] <= t["upstream_signal_ok"] <= t["no_pair"] <= t["lts_search_ran"]
