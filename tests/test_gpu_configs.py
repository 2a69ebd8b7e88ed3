"""BASELINE.json configs 2 (CFO = 0 variant and the distance to the upstream-literal arithmetic), 3 (reduced BER
sweep) and 5 (all six Kodak images end to end) on the MI355X, through the C ABI / the drop-in block.

"Oracle" below is oracle/ in SPEC mode unless LIBM is named.  PARITY UNPINNED: the reference holds no vectors for this
path; "bit-exact" means equal to that restatement."""
import json
import os

import numpy as np
import pytest

from wifirx import txgen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CFO_20PPM = 2 * np.pi * 20e-6 * 5.89e9 / 20e6


# ------------------------------------------------------------------------------------------------------------------
# config 3: 64-QAM 3/4 under the SV-derived multipath, LS equaliser, SNR sweep
@pytest.mark.timeout(900)
def test_config3_reduced_ber_sweep(orc):
    """SNR 5, 10, ..., 30 dB x 2048 frames: the GPU's frame records and hard decisions equal the oracle's at every
    point; coded-bit BER falls monotonically, and BER, FER and detection rate sit within a confidence interval of
    tests/golden/config3_ber_table.json -- made by the ORACLE on the CPU from host-generated frames
    (tests/golden/make_config3_ber_table.py, 30 000 frames per point; no GPU output in it), so these asserts are oracle parity
    too, on independent noise."""
    from wifirx import capi
    table = {p["snr_db"]: p for p in json.load(open(os.path.join(GOLD, "config3_ber_table.json")))["points"]}
    taps = np.load(os.path.join(GOLD, "sv_taps.npy"))
    n_t = taps.shape[0]
    n_frames, slot_len, lead = 2 * n_t, 1472, 160
    psdu = txgen.make_psdus(n_t, 294, seed=31)
    tx = txgen.encode_psdus(psdu, 7)
    faded = txgen.impair(tx.samples, None, cfo=0.0, lead=0, total=tx.samples.shape[1] + 8, taps=taps)
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=0, chan_est=capi.EQ_LS)
    slots = rx.alloc(n_frames * slot_len * 8)
    bits_tx = ((tx.data_idx[..., None] >> np.arange(6)) & 1).astype(np.uint8)
    prm = orc.make_params(max_sym=tx.n_sym)
    tmpl = np.arange(n_frames) % n_t
    bers, report = [], []
    for snr in (5, 10, 15, 20, 25, 30):
        rx.synth_slots(faded, slots.ptr, slot_len, n_frames, lead, float(snr), float(CFO_20PPM), 1000 + snr)
        dev = rx.alloc_out(n_frames, psdu_stride=320)
        rx.demod_batch_dev(slots.ptr, slot_len, n_frames, dev)
        rx.decode_batch_dev(n_frames, dev)
        rx.sync()
        r = rx.download_out(dev, n_frames)
        rx.free_out(dev)
        x = slots.download(np.complex64, n_frames * slot_len)
        o = orc.demod_batch(x, slot_len, prm, n_threads=os.cpu_count() or 1)
        opsdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=320, n_threads=os.cpu_count() or 1)
        assert np.array_equal(r["frames"], o["frames"]), "frame records differ from the oracle at %d dB" % snr
        assert np.array_equal(r["idx"], o["idx"]), "hard decisions differ from the oracle at %d dB" % snr
        dec = (o["frames"]["flags"] & orc.F_DECODED) != 0
        assert np.array_equal(r["psdu"][dec][:, :294], opsdu[dec][:, :294])
        good = ((r["frames"]["flags"] & capi.F_COMPLETE) != 0) & (r["frames"]["encoding"] == 7) & (r["frames"]["psdu_len"] == 294)
        bits_rx = ((r["idx"][good][..., None] >> np.arange(6)) & 1).astype(np.uint8)
        per_frame = (bits_rx != bits_tx[tmpl[good]]).reshape(int(good.sum()), -1).mean(axis=1)
        ber, se = float(per_frame.mean()), float(per_frame.std() / np.sqrt(len(per_frame)))
        ref = table[snr]
        # 5 standard errors of the difference (this sample's and the table's) + 1 % of the table value
        se_d = float(np.hypot(se, ref.get("coded_ber_se", 0.0)))
        assert abs(ber - ref["coded_ber"]) < 5 * se_d + 0.01 * ref["coded_ber"], (snr, ber, ref["coded_ber"], se_d)
        assert abs(good.mean() - ref["detected_and_signal_ok"]) < 0.05
        ok = ((r["frames"]["flags"] & capi.F_CRC_OK) != 0) & (r["psdu"][:, :294] == psdu[tmpl]).all(axis=1)
        fer = 1.0 - ok.mean()
        assert abs(fer - ref["fer"]) < 0.05, (snr, fer, ref["fer"])
        bers.append(ber)
        report.append({"snr_db": snr, "frames": n_frames, "coded_ber": ber, "coded_ber_se": se, "fer": float(fer),
                       "detected_and_signal_ok": float(good.mean()), "oracle_mismatches": 0})
    assert all(a > b for a, b in zip(bers, bers[1:])), bers          # BER falls with SNR
    slots.free()
    rx.close()
    _record("config3_reduced_sweep", report)


# ------------------------------------------------------------------------------------------------------------------
# config 5: six images, IRS_user TX (CPU) -> IQ -> wifi_phy_rx -> Extract Pics -> pixels
@pytest.mark.timeout(1200)
def test_config5_six_kodak_images_pixel_exact():
    from wifirx import app, block, grshim
    imgs = np.load(os.path.join(GOLD, "kodim_300.npz"))
    assert len(imgs.files) == 6
    report = {}
    n_ok = 0
    for name in sorted(imgs.files):
        rx = block.wifi_phy_rx(bandwidth=20e6, frequency=5.89e9, publish_carrier=False)
        got = []
        pics = app.extract_pics(sink=got.append)
        grshim.msg_connect(rx, "mac_out", pics, "MAC")
        img = imgs[name]
        pieces = app.detach_image_sorted(img)
        assert len(pieces) == 2700
        payloads = [app.pack_piece(p) for p in pieces]
        streams = [None] * len(payloads)
        by_len = {}
        for k, p in enumerate(payloads):
            by_len.setdefault(len(p), []).append(k)
        for L, ks in by_len.items():
            psdus = np.stack([np.frombuffer(txgen.mac_frame(payloads[k], seq=k), dtype=np.uint8) for k in ks])
            tx = txgen.encode_psdus(psdus, 0, seeds=[(k % 127) + 1 for k in ks])
            for row, k in enumerate(ks):
                streams[k] = tx.samples[row] * np.float32(10.0)                  # 20 dB over the unit noise below
        x = np.concatenate([np.concatenate([np.zeros(100, np.complex64), s, np.zeros(1000, np.complex64)]) for s in streams])
        rng = np.random.default_rng(sum(name.encode()))
        x += ((rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size)) * np.sqrt(0.5)).astype(np.complex64)
        grshim.run_stream(rx, x, chunk=8192)          # one image = one stream; stop() settles the last frames
        out = np.zeros_like(img)
        for g in got:
            app.redraw_image(app.load_piece(g), out)
        assert len(got) == 2700, (name, len(got))
        assert np.array_equal(out, img), name
        report[name] = {"pieces": 2700, "pdus": len(got), "pixel_exact": True, "samples": int(x.size)}
        n_ok += rx.stats()["frames_crc_ok"]
        rx.close()
    assert n_ok == 6 * 2700
    _record("config5_six_images", report)


# ------------------------------------------------------------------------------------------------------------------
# config 2 and the upstream-literal arithmetic (oracle LIBM mode)
def _geometry(name):
    if name == "config1":      # BPSK 1/2, 294 B: 99 symbols
        return dict(enc=0, slot=8576, n=256, taps=None, snr=20.0)
    if name == "config2":      # QPSK 1/2, 294 B: 50 symbols
        return dict(enc=2, slot=4608, n=512, taps=None, snr=20.0)
    return dict(enc=7, slot=1472, n=1024, taps=np.load(os.path.join(GOLD, "sv_taps.npy")), snr=30.0)       # config 3


def _gpu_vs_libm(orc, geo, cfo_max, seed):
    from wifirx import capi
    nb = txgen.RATE_TABLE[geo["enc"]][0]
    n_t = 64
    tx = txgen.encode_psdus(txgen.make_psdus(n_t, 294, seed=seed), geo["enc"])
    tmpl = tx.samples
    if geo["taps"] is not None:
        tmpl = txgen.impair(tx.samples, None, cfo=0.0, lead=0, total=tx.samples.shape[1] + 8, taps=geo["taps"][:n_t])
    rx = capi.WifiRx(max_sym=tx.n_sym, llr_bits=nb, want_carrier=True)
    n, slot = geo["n"], geo["slot"]
    slots = rx.alloc(n * slot * 8)
    rx.synth_slots(tmpl, slots.ptr, slot, n, 160, geo["snr"], float(cfo_max), seed)
    x = slots.download(np.complex64, n * slot)
    slots.free()
    r = rx.demod_batch(x, slot)
    rx.close()
    spec = orc.demod_batch(x, slot, orc.make_params(max_sym=tx.n_sym, llr_bits=nb), want_eq=True, n_threads=os.cpu_count() or 1)
    libm = orc.demod_batch(x, slot, orc.make_params(max_sym=tx.n_sym, llr_bits=nb, math_mode=orc.MATH_LIBM), want_eq=True,
                           n_threads=os.cpu_count() or 1)
    assert np.array_equal(r["frames"], spec["frames"]) and np.array_equal(r["idx"], spec["idx"])
    assert np.array_equal(r["llr"], spec["llr"]) and np.array_equal(r["carrier"], spec["eq"])
    same_frames = bool(np.array_equal(r["frames"]["trigger"], libm["frames"]["trigger"]) and
                       np.array_equal(r["frames"]["frame_start"], libm["frames"]["frame_start"]) and
                       np.array_equal(r["frames"]["flags"], libm["frames"]["flags"]) and
                       np.array_equal(r["frames"]["n_sym_out"], libm["frames"]["n_sym_out"]))
    complete = int(((r["frames"]["flags"] & capi.F_COMPLETE) != 0).sum())
    n_dec = int(r["idx"].size)
    diff = int((r["idx"] != libm["idx"]).sum())

    def dist(a, ref):
        """max |a - ref| over the batch, relative to the largest |ref| (the reading the earlier rounds reported), and element by
        element relative to |ref| with a floor of 1e-3 of the largest (the stricter reading of "relative")"""
        d = np.abs(a - ref)
        big = float(np.abs(ref).max())
        return float(d.max() / big), float((d / np.maximum(np.abs(ref), 1e-3 * big)).max())

    llr_rel, llr_elem = dist(r["llr"], libm["llr"])
    eq_rel, eq_elem = dist(r["carrier"], libm["eq"])
    # the same against LIBM with the derotation angles formed in double (oracle switch libm_exact_phase): everything upstream's
    # arithmetic does EXCEPT rounding -cfo_c * m and cfo_f * m to float32 before the sine / cosine
    ex = orc.demod_batch(x, slot, orc.make_params(max_sym=tx.n_sym, llr_bits=nb, math_mode=orc.MATH_LIBM, libm_exact_phase=1),
                         want_eq=True, n_threads=os.cpu_count() or 1)
    same_ex = bool(np.array_equal(r["frames"]["trigger"], ex["frames"]["trigger"]) and
                   np.array_equal(r["frames"]["frame_start"], ex["frames"]["frame_start"]) and
                   np.array_equal(r["frames"]["flags"], ex["frames"]["flags"]))
    llr_rel_x, llr_elem_x = dist(r["llr"], ex["llr"])
    eq_rel_x, eq_elem_x = dist(r["carrier"], ex["eq"])
    # how far upstream's literal arithmetic is from ITSELF with exact angles: the size of the angle rounding alone
    lit_rel, lit_elem = dist(libm["llr"], ex["llr"])
    return dict(frames=n, frames_complete=complete, same_trigger_start_flags=same_frames, decisions=n_dec,
                decisions_differing=diff, llr_max_rel=llr_rel, eq_max_rel=eq_rel, llr_elem_rel=llr_elem, eq_elem_rel=eq_elem,
                exact_phase=dict(same_trigger_start_flags=same_ex, decisions_differing=int((r["idx"] != ex["idx"]).sum()),
                                 llr_max_rel=llr_rel_x, eq_max_rel=eq_rel_x, llr_elem_rel=llr_elem_x, eq_elem_rel=eq_elem_x),
                upstream_literal_vs_its_own_exact_phase=dict(llr_max_rel=lit_rel, llr_elem_rel=lit_elem,
                                                             decisions_differing=int((libm["idx"] != ex["idx"]).sum())))


@pytest.mark.timeout(600)
def test_config2_cfo0_matches_upstream_literal_arithmetic(orc):
    """BASELINE config 2, CFO = 0 variant, full frame length (50 symbols): the GPU's hard decisions equal those of the
    oracle's LIBM mode (upstream's own operation order: libm sincos/atan2, C99 complex division, float32 derotation
    angle) and its LLRs are within north_star's 1e-5 relative."""
    d = _gpu_vs_libm(orc, _geometry("config2"), 0.0, 2025)
    assert d["frames_complete"] == d["frames"] and d["same_trigger_start_flags"]
    assert d["decisions_differing"] == 0
    assert d["llr_max_rel"] <= 1e-5 and d["eq_max_rel"] <= 1e-5, d
    _record("config2_cfo0_vs_libm", d)


@pytest.mark.timeout(900)
def test_distance_to_upstream_literal_arithmetic_at_20ppm(orc):
    """The same distance with the full +-20 ppm CFO, per BASELINE config geometry -- RECORDED (gpurun_out/ and, once
    copied, profiles/), and bounded by the values DESIGN.md section 2 states.  North_star's 1e-5 on the soft values is NOT met
    at +-20 ppm: config 1 (BPSK, 99 symbols: the derotation angle reaches 300 rad, where upstream's float32 `cfo*n` has
    an ulp of 3e-5 rad) and config 3 (deep fades divide by a small |H|) measure 1-3e-5, config 2 about 1e-5 -- stated-
    tolerance misses that are upstream's own quantisation noise, reported as such (`meets_north_star_1e-5`)."""
    bound = {"config1": 6e-5, "config2": 2e-5, "config3": 3e-5}
    rec = {}
    for name in ("config1", "config2", "config3"):
        d = _gpu_vs_libm(orc, _geometry(name), CFO_20PPM, 7 + len(name))
        d["llr_bound_asserted"] = bound[name]
        d["meets_north_star_1e-5"] = bool(d["llr_max_rel"] <= 1e-5)
        rec[name] = d
        assert d["same_trigger_start_flags"], name
        assert d["decisions_differing"] <= max(2, d["decisions"] // 500_000), (name, d)    # a sample on a slicer boundary
        assert d["llr_max_rel"] <= bound[name], (name, d)
        # with upstream's angle rounding taken out (libm_exact_phase) the distance is north_star's 1e-5
        assert d["exact_phase"]["same_trigger_start_flags"], name
        assert d["exact_phase"]["llr_max_rel"] <= 1e-5 and d["exact_phase"]["eq_max_rel"] <= 1e-5, (name, d)
    _record("distance_to_libm_20ppm", rec)


def _record(key, value):
    """keeps what the tests measured: gpurun_out/r05_gpu_configs.json (copied to profiles/ by the builder)"""
    path = os.path.join(ROOT, "gpurun_out", "r05_gpu_configs.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        cur = {}
        if os.path.exists(path):
            with open(path) as f:
                cur = json.load(f)
        cur[key] = value
        with open(path, "w") as f:
            json.dump(cur, f, indent=1)
    except OSError:
        pass
