"""The bulk parity evidence, in the driver-run suite: randomised campaigns of GPU (through the C ABI)
against the oracle with EXACT equality on frame records, decisions, LLRs, equalised points, CSI, flags after decode_mac and
PSDU bytes -- the scripts of tests/campaigns/ run as tests.  Budget: about 200 s of the suite's 900 s on the GPU box.

Every test writes its result dictionary to gpurun_out/r03_campaign_*.json (copied to profiles/ by hand after a run).
"""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "campaigns"))

pytestmark = pytest.mark.gpu


def _record(name, res):
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "r03_campaign_%s.json" % name), "w") as f:
            json.dump(res, f)
    except OSError:
        pass


@pytest.mark.parametrize("tag,seed,snrs", [
    ("seed301", 301, (6.0, 12.0, 20.0, 28.0, 35.0)),
    ("seed302_low_snr", 302, (0.0, 2.0, 4.0, 6.0, 8.0, 10.0)),
])
def test_random_frames_four_equalisers(tag, seed, snrs):
    """20 000 frames per run, random rate / length / SNR / CFO / lead-in / channel (flat or SV multipath), each through LS,
    LMS, COMB and STA: 160 000 frame-equaliser pairs over the two runs."""
    import parity_campaign as pc
    res = pc.run(20000, seed, snrs=snrs)
    _record("frames_" + tag, res)
    for name, v in res["equalisers"].items():
        assert v["total_mismatches"] == 0, (name, v["mismatches"])
        assert v["detected"] > 0.5 * res["frames"]
    assert res["equalisers"]["LS"]["crc_ok"] > (0.5 if min(snrs) >= 6 else 0.1) * res["frames"]


def test_random_frames_plain_outputs():
    """The output set bench.py times (decisions + LLRs alone): the generator's frames come in groups of one rate, so most
    waves run the kernel's constellation loops with whole-line stores -- with random lead-ins, lengths, SNRs and channels,
    and ragged waves where two groups meet."""
    import parity_campaign as pc
    res = pc.run(20000, 308, plain=True)
    _record("frames_seed308_plain_outputs", res)
    for name, v in res["equalisers"].items():
        assert v["total_mismatches"] == 0, (name, v["mismatches"])
        assert v["detected"] > 0.5 * res["frames"]


def test_random_frames_throughput_decoder():
    """The same kind of batch with decode_mac forced onto the throughput kernel (128 frames per wave, lane = two frames):
    mixed-rate waves, every wave position filled."""
    import parity_campaign as pc
    res = pc.run(20000, 303, equalisers=(0, 3), decode_small_max=0)
    _record("frames_seed303_throughput_decoder", res)
    for name, v in res["equalisers"].items():
        assert v["total_mismatches"] == 0, (name, v["mismatches"])


def test_random_frames_four_frames_per_lane_decoder(monkeypatch):
    """... and onto decode_q_kernel (four frames per lane), which batches of a million frames take by themselves: random
    rates (grouped by rate on the device), random lengths inside a wave (frames ending at different steps), bit errors."""
    import parity_campaign as pc
    monkeypatch.setenv("WIFIRX_DECODE_Q", "1")
    monkeypatch.setenv("WIFIRX_DECODE_FPW", "256")
    monkeypatch.setenv("WIFIRX_DECODE_OVL", "1")          # trace-back overlapped with the next task (uniform tasks are deferred, others not)
    res = pc.run(20000, 306, equalisers=(0, 1), decode_small_max=0)
    _record("frames_seed306_four_frames_per_lane_decoder", res)
    for name, v in res["equalisers"].items():
        assert v["total_mismatches"] == 0, (name, v["mismatches"])
    monkeypatch.setenv("WIFIRX_DECODE_OVL", "2")          # speculative walks under the task's own add-compare-select (round 5)
    res = pc.run(20000, 309, equalisers=(0, 1), decode_small_max=0)
    _record("frames_seed309_four_frames_per_lane_decoder_speculative", res)
    for name, v in res["equalisers"].items():
        assert v["total_mismatches"] == 0, (name, v["mismatches"])
    monkeypatch.setenv("WIFIRX_DECODE_FPW", "0")
    res = pc.run(1200, 307, long_frames=True, equalisers=(0,), decode_small_max=0)
    _record("long_frames_seed307_four_frames_per_lane_decoder", res)
    assert res["equalisers"]["LS"]["total_mismatches"] == 0, res["equalisers"]["LS"]["mismatches"]


def test_long_frames_four_equalisers():
    """640 frames of up to 511 OFDM symbols (PSDUs up to 1530 bytes at every rate: the longest decode_mac accepts; the
    renormalisation of its path metrics over 12 000+ trellis steps; the carried derotation phasor over 40 000 samples)."""
    import parity_campaign as pc
    res = pc.run(640, 304, long_frames=True)
    _record("long_frames_seed304", res)
    for name, v in res["equalisers"].items():
        assert v["total_mismatches"] == 0, (name, v["mismatches"])
    assert res["equalisers"]["LS"]["complete"] > 0.5 * res["frames"]


def test_random_streams():
    """120 continuous streams x 30 frames: gaps down to zero samples, frames cut off by the next one, random push sizes
    and batch thresholds, a random equaliser per stream -- same frames, records, decisions and PSDUs as the oracle's
    stream driver."""
    import stream_campaign as sc
    res = sc.run(120, 30, 305)
    _record("streams_seed305", res)
    assert res["all_equal"], res
    assert res["frames_found"] > 0.7 * res["frames_sent"]


@pytest.mark.parametrize("threshold,seed", [(0.56, 25), (0.35, 26)])
def test_rule6_low_snr_gpu_equals_exhaustive_search(orc, threshold, seed):
    """Spec rule 6 on the GPU against sync_long's exhaustive float32 search (the oracle with lts_search = 1) at 0..8 dB,
    flat and multipath, four constellations: 72 groups x 250 frames per run.  Frame records -- trigger, frame start, fine
    CFO, flags, SNR, SIGNAL contents -- must be EQUAL: the GPU's candidate stage never loses a peak that matters."""
    import lts_rule6 as camp
    from wifirx import capi
    per = 250
    n_det = n_sync = 0
    rx = capi.WifiRx(max_sym=8, llr_bits=0, sensitivity=threshold)
    bad = []
    for gi, (snr, enc, chan) in enumerate(camp.groups()):
        iq = camp.make_group(per, snr, enc, chan, seed * 1000 + gi)
        g = rx.demod_batch(iq, camp.SLOT)["frames"]
        o = orc.demod_batch(iq, camp.SLOT, orc.make_params(max_sym=8, lts_search=1, threshold=threshold),
                            n_threads=os.cpu_count() or 1)["frames"]
        n_det += int(((o["flags"] & orc.F_DETECTED) != 0).sum())
        n_sync += int(((o["flags"] & orc.F_SYNC) != 0).sum())
        for k in np.nonzero(g != o)[0]:
            bad.append((snr, enc, chan, int(k), g[k].tolist(), o[k].tolist()))
    rx.close()
    _record("rule6_gpu_thr%02d" % int(threshold * 100), {"threshold": threshold, "frames": 72 * per, "detected": n_det,
                                                         "sync": n_sync, "records_differing_from_exhaustive_search": len(bad),
                                                         "differing": bad[:50]})
    assert n_det > (0.4 if threshold > 0.5 else 0.9) * 72 * per
    assert not bad, bad[:5]
