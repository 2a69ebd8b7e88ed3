"""The PRODUCT against the structurally independent receiver, without the oracle in between (round 4).

`tests/independent_rx.py` (float64 NumPy, written to SURVEY App. A, shares no code or table with product or oracle) and the
HIP chain through the C ABI on the same random frames: same trigger, frame start, SIGNAL field; coarse / fine CFO to float32
accuracy; hard decisions equal but for points within float32 noise of a decision boundary (counted, bounded, recorded in
gpurun_out/r04_gpu_vs_independent.json).  This is not bit parity -- the kernels run the float32 SPEC arithmetic, the
independent receiver float64 -- it is the check that what the kernels compute IS the chain of the reference's flowgraph
(`gnu_radio/IRS_AP.py:268-285`) as the survey's appendix describes it, made by code that was not derived from the oracle."""
import json
import os

import numpy as np
import pytest

from independent_rx import IndependentRx
from wifirx import txgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpu_chain_agrees_with_the_independent_receiver():
    from wifirx import capi
    rx_i = IndependentRx(bandwidth=20e6, frequency=5.89e9)
    taps_all = np.load(os.path.join(ROOT, "tests", "golden", "sv_taps.npy"))
    rng = np.random.default_rng(4242)
    tot = dict(frames=0, detected_both=0, trigger_equal=0, frame_start_equal=0, signal_equal=0, complete_both=0,
               decisions=0, decisions_differ=0)
    max_cfo_c = max_cfo_f = 0.0
    n_per = 600
    for rep in range(3):
        for enc in range(8):
            plen = int(rng.integers(28, 160))
            snr = float(rng.uniform(8.0 + 2.2 * enc, 30.0))
            tx = txgen.encode_psdus(txgen.make_psdus(n_per, plen, seed=int(rng.integers(1 << 30))), enc)
            S = ((160 + tx.samples.shape[1] + 200 + 63) // 64) * 64
            taps = taps_all[rng.integers(0, taps_all.shape[0], n_per)] if rep == 2 else None
            iq = txgen.impair(tx.samples, snr, cfo=rng.uniform(-0.037, 0.037, n_per), lead=int(rng.integers(100, 200)), total=S,
                              seed=int(rng.integers(1 << 30)), taps=taps)
            g = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0)
            r = g.demod_batch(iq.reshape(-1), S)
            g.close()
            i = rx_i.receive(iq, max_sym=tx.n_sym)
            fr = r["frames"]
            g_det, g_sync = (fr["flags"] & capi.F_DETECTED) != 0, (fr["flags"] & capi.F_SYNC) != 0
            g_sig, g_cmp = (fr["flags"] & capi.F_SIGNAL) != 0, (fr["flags"] & capi.F_COMPLETE) != 0
            tot["frames"] += n_per
            both = g_det & i["detected"]
            tot["detected_both"] += int(both.sum())
            same_t = both & (fr["trigger"] == i["trigger"])
            tot["trigger_equal"] += int(same_t.sum())
            same_fs = same_t & g_sync & i["sync"] & (fr["frame_start"] == i["frame_start"])
            tot["frame_start_equal"] += int(same_fs.sum())
            if same_fs.any():
                max_cfo_c = max(max_cfo_c, float(np.abs(fr["cfo_coarse"][same_fs] - i["cfo_coarse"][same_fs]).max()))
                max_cfo_f = max(max_cfo_f, float(np.abs(fr["cfo_fine"][same_fs] - i["cfo_fine"][same_fs]).max()))
            same_sig = same_fs & g_sig & i["signal_ok"] & (fr["encoding"] == i["encoding"]) & (fr["psdu_len"] == i["psdu_len"])
            tot["signal_equal"] += int(same_sig.sum())
            cb = same_sig & g_cmp & i["complete"]
            tot["complete_both"] += int(cb.sum())
            tot["decisions"] += int((fr["n_sym"][cb].astype(np.int64) * 48).sum())
            tot["decisions_differ"] += int((r["idx"][cb] != i["idx"][cb]).sum())
    tot.update(max_abs_cfo_coarse_diff=max_cfo_c, max_abs_cfo_fine_diff=max_cfo_f)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r04_gpu_vs_independent.json"), "w") as f:
            json.dump(tot, f, indent=1)
    except OSError:
        pass
    assert tot["frames"] == 14400
    assert tot["detected_both"] >= 0.99 * tot["frames"], tot
    assert tot["trigger_equal"] >= 0.999 * tot["detected_both"], tot            # float32 block sums vs float64 sliding sums at the threshold
    assert tot["frame_start_equal"] >= 0.999 * tot["trigger_equal"], tot
    assert max_cfo_c < 2e-6 and max_cfo_f < 2e-6, tot                             # sp_atan2: |err| < 4e-7 rad, / 16 resp. / 64
    assert tot["signal_equal"] >= 0.999 * tot["frame_start_equal"], tot
    assert tot["complete_both"] >= 0.9 * tot["frames"], tot
    assert tot["decisions_differ"] <= 2e-5 * tot["decisions"], tot                # points within float32 noise of a boundary
