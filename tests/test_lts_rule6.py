"""Spec rule 6 (the two-stage LTS search the HIP kernels run) against sync_long's exhaustive search over all 320 float32
magnitudes (`gnu_radio/IRS_AP.py:269,282`), where peak ranking is fragile: 0..8 dB, flat and
Saleh-Valenzuela multipath, all four constellations.  CPU only (oracle vs oracle: the GPU-vs-oracle leg of the same
frames is tests/test_gpu_campaign.py::test_rule6_low_snr_gpu_equals_exhaustive_search).

Round 2's rule (top two integer peaks whenever 64 apart) lost the exhaustive search's pair on 4 frames of 154 000 here;
the frames are pinned below.  The present rule (top two only when the third integer magnitude is below 7/8 of the
second, else the top eight) gives the exhaustive search's records on every frame of the campaign
(`profiles/r03_lts_rule6_*.json`: 2 x 100 800 frames).
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "campaigns"))
import lts_rule6 as camp  # noqa: E402


@pytest.mark.parametrize("threshold,per_group,seed", [(0.56, 700, 15), (0.35, 700, 16)])
def test_rule6_equals_exhaustive_float_search_at_low_snr(threshold, per_group, seed):
    """72 groups x 700 frames per run (50 400; the two runs together 100 800): every frame record of the two-stage
    search equals the exhaustive float32 search's, and where the upstream-literal arithmetic triggers at the same sample
    it finds the same frame start."""
    res = camp.run(per_group, seed, threshold=threshold)
    tot = res["totals"]
    assert tot["frames"] == 72 * per_group
    assert tot["detected"] > (0.4 if threshold > 0.5 else 0.9) * tot["frames"]      # the sweep does reach the LTS search
    assert tot["exhaustive_differs"] == 0, res["differing_frames"]
    assert tot["sync_spec"] == tot["sync_exhaustive"]
    # arithmetic distance to the upstream-literal mode: no frame-start change expected either; a 1-ulp reordering of two
    # near-equal peaks is possible in principle (any two implementations of upstream differ so), hence a bound, not zero
    assert tot["libm_frame_start_differs"] + tot["libm_sync_flag_differs"] <= 2, res["differing_frames"]
    assert tot["largest_cfo_fine_distance_to_libm"] < 1e-6


# (group seed, snr, encoding, channel, frame, threshold, frame_start of the exhaustive search): the frames round 2's
# rule got wrong -- two paths of nearly equal strength, the third integer magnitude within 0.3 % of the second
ROUND2_MISSES = [
    (5049, 6.0, 0, "sv", 638, 0.56, 151),
    (6001, 0.0, 0, "sv", 1205, 0.35, 162),
    (6035, 4.0, 2, "sv", 22, 0.35, 61),
    (6039, 4.0, 6, "sv", 162, 0.35, 162),
]


@pytest.mark.parametrize("gseed,snr,enc,chan,frame,threshold,fs_exhaustive", ROUND2_MISSES)
def test_frames_round2_rule_missed(orc, gseed, snr, enc, chan, frame, threshold, fs_exhaustive):
    iq = camp.make_group(1400, snr, enc, chan, gseed)
    slot = iq[frame * camp.SLOT:(frame + 1) * camp.SLOT]
    out = {}
    for name, ls in (("spec", 0), ("exhaustive", 1)):
        prm = orc.make_params(max_sym=8, lts_search=ls, threshold=threshold)
        out[name] = orc.demod_batch(slot, camp.SLOT, prm)["frames"][0]
    assert out["exhaustive"]["frame_start"] == fs_exhaustive
    assert out["spec"] == out["exhaustive"]
    top_s, _ = camp.lts_debug(orc, slot, orc.MATH_SPEC, 0, threshold)
    top_x, _ = camp.lts_debug(orc, slot, orc.MATH_SPEC, 1, threshold)
    assert top_s == top_x and min(top_s) >= 0          # the wide path ran: four peaks, the exhaustive search's four
