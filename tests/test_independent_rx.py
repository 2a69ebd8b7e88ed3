"""the oracle against a structurally independent receiver.

`tests/independent_rx.py` is a float64 NumPy receiver written to SURVEY.md Appendix A literally (np.fft, np.angle, complex
division; constants from the reference's flowgraph via tests/golden/grc_constants.json and from IEEE 802.11) that shares no
code and no table with the product or the oracle.  The oracle's LIBM mode -- the upstream-literal arithmetic in float32 --
must agree with it on WHAT is computed: trigger, frame start, both frequency offsets, the SIGNAL field and the hard
decisions, over >= 20 000 random frames of all rates at 6..30 dB with and without multipath.  What remains between a
float32 and a float64 evaluation of the same formulas is counted and bounded; a common-mode slip of the oracle's two
modes (a sign, an index, a cadence, a constant) would show as a gross disagreement.

Runs on the CPU (`-m "not gpu"`); the GPU is tied to the oracle's SPEC mode bit for bit by the `-m gpu` suite, and SPEC to
LIBM by tests/test_spec_math.py and the measured distances of DESIGN.md section 2."""
import json
import os

import numpy as np

from independent_rx import IndependentRx
from wifirx import txgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_independent_receiver_shares_nothing_with_product_or_oracle():
    src = open(os.path.join(ROOT, "tests", "independent_rx.py")).read()
    code = "\n".join(l for l in src.split('"""')[2].splitlines() if not l.strip().startswith("#"))     # everything behind the module docstring
    for banned in ("wifirx", "oracle", "wifirx_tables", "ctypes"):
        assert banned not in code, banned


def test_oracle_libm_mode_agrees_with_the_independent_receiver(orc):
    rx = IndependentRx(bandwidth=20e6, frequency=5.89e9)
    taps_all = np.load(os.path.join(GOLD, "sv_taps.npy"))
    rng = np.random.default_rng(20261005)
    cores = os.cpu_count() or 1
    tot = dict(frames=0, detected_both=0, trigger_equal=0, sync_both=0, frame_start_equal=0, signal_both=0, signal_equal=0,
               complete_both=0, decisions=0, decisions_differ=0, frames_with_differing_decisions=0,
               detected_only_oracle=0, detected_only_independent=0, sync_only_oracle=0, sync_only_independent=0)
    max_cfo_c = max_cfo_f = max_snr = 0.0
    n_per = 420
    for rep in range(6):
        for enc in range(8):
            plen = int(rng.integers(28, 140))
            snr = float(rng.uniform(6.0 + 2.2 * enc, 30.0))
            tx = txgen.encode_psdus(txgen.make_psdus(n_per, plen, seed=int(rng.integers(1 << 30))), enc)
            S = ((160 + tx.samples.shape[1] + 200 + 63) // 64) * 64
            taps = None
            if rep % 3 == 2:                                  # a third of the batches under the config-3 multipath draws
                taps = taps_all[rng.integers(0, taps_all.shape[0], n_per)]
            iq = txgen.impair(tx.samples, snr, cfo=rng.uniform(-0.037, 0.037, n_per), lead=int(rng.integers(100, 200)), total=S,
                              seed=int(rng.integers(1 << 30)), taps=taps)
            r = rx.receive(iq, max_sym=tx.n_sym)
            o = orc.demod_batch(iq.reshape(-1), S, orc.make_params(max_sym=tx.n_sym, math_mode=orc.MATH_LIBM), n_threads=cores)
            fr = o["frames"]
            o_det, o_sync = (fr["flags"] & orc.F_DETECTED) != 0, (fr["flags"] & orc.F_SYNC) != 0
            o_sig, o_cmp = (fr["flags"] & orc.F_SIGNAL) != 0, (fr["flags"] & orc.F_COMPLETE) != 0
            tot["frames"] += n_per
            both = o_det & r["detected"]
            tot["detected_both"] += int(both.sum())
            tot["detected_only_oracle"] += int((o_det & ~r["detected"]).sum())
            tot["detected_only_independent"] += int((~o_det & r["detected"]).sum())
            same_t = both & (fr["trigger"] == r["trigger"])
            tot["trigger_equal"] += int(same_t.sum())
            if same_t.any():
                max_cfo_c = max(max_cfo_c, float(np.abs(fr["cfo_coarse"][same_t] - r["cfo_coarse"][same_t]).max()))
            sb = same_t & o_sync & r["sync"]
            tot["sync_both"] += int(sb.sum())
            tot["sync_only_oracle"] += int((same_t & o_sync & ~r["sync"]).sum())
            tot["sync_only_independent"] += int((same_t & ~o_sync & r["sync"]).sum())
            same_fs = sb & (fr["frame_start"] == r["frame_start"])
            tot["frame_start_equal"] += int(same_fs.sum())
            if same_fs.any():
                max_cfo_f = max(max_cfo_f, float(np.abs(fr["cfo_fine"][same_fs] - r["cfo_fine"][same_fs]).max()))
                max_snr = max(max_snr, float(np.abs(fr["snr_db"][same_fs] - r["snr_db"][same_fs]).max()))
            sg = same_fs & o_sig & r["signal_ok"]
            tot["signal_both"] += int((same_fs & (o_sig | r["signal_ok"])).sum())
            same_sig = sg & (fr["encoding"] == r["encoding"]) & (fr["psdu_len"] == r["psdu_len"])
            tot["signal_equal"] += int(same_sig.sum())
            cb = same_sig & o_cmp & r["complete"]
            tot["complete_both"] += int(cb.sum())
            d = (o["idx"][cb] != r["idx"][cb])
            # decisions of the symbols a frame has (rows behind n_sym are zero on both sides)
            tot["decisions"] += int((fr["n_sym"][cb].astype(np.int64) * 48).sum())
            tot["decisions_differ"] += int(d.sum())
            tot["frames_with_differing_decisions"] += int(d.reshape(d.shape[0], -1).any(axis=1).sum())
    tot.update(max_abs_cfo_coarse_diff=max_cfo_c, max_abs_cfo_fine_diff=max_cfo_f, max_abs_snr_db_diff=max_snr)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r04_independent_rx.json"), "w") as f:
            json.dump(tot, f, indent=1)
    except OSError:
        pass
    assert tot["frames"] >= 20000
    # the same samples trigger; where float32 running sums and float64 window sums put |A|/P on different sides of the
    # threshold the trigger moves by a sample: rare
    assert tot["detected_both"] >= 0.995 * tot["frames"], tot
    assert tot["trigger_equal"] >= 0.999 * tot["detected_both"], tot
    assert max_cfo_c < 1e-6, tot
    # the same LTS pair (top-4 / 64-63-65 rule); the fine offset to float32 accuracy
    assert tot["sync_only_oracle"] + tot["sync_only_independent"] <= 0.001 * tot["trigger_equal"], tot
    assert tot["frame_start_equal"] >= 0.999 * tot["sync_both"], tot
    assert max_cfo_f < 1e-6, tot
    assert max_snr < 1e-3, tot
    # the same SIGNAL field (a tie in the 24-step trellis may be broken differently: only on frames with SIGNAL errors)
    assert tot["signal_equal"] >= 0.999 * tot["signal_both"], tot
    # the same decisions but for points within float32 noise of a decision boundary
    assert tot["complete_both"] >= 0.9 * tot["frames"], tot
    assert tot["decisions_differ"] <= 2e-5 * tot["decisions"], tot
