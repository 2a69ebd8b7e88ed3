"""Stream mode (wifirx_push / wifirx_poll and the wifi_phy_rx block) against the oracle's stream driver."""
import numpy as np
import pytest

from wifirx import txgen

pytestmark = pytest.mark.gpu


def build_stream(seed=3, noise=True):
    """A stream like the one IRS_user sends (frames with 100 front / 1000 tail padding,
    gnu_radio/IRS_user.py:193), mixed rates and lengths, per-frame CFO, unit-variance noise."""
    rng = np.random.default_rng(seed)
    parts, psdus = [], []
    specs = [(0, 60), (2, 294), (4, 500), (7, 1000), (2, 294), (5, 120), (3, 294), (6, 64), (1, 200), (2, 294)]
    for k, (enc, plen) in enumerate(specs):
        psdu = txgen.make_psdus(1, plen, seed=seed * 100 + k, seq0=k)
        tx = txgen.encode_psdus(psdu, enc, seeds=[(k % 127) + 1])
        n = tx.samples.shape[1]
        cfo = rng.uniform(-0.03, 0.03)
        sig = tx.samples[0] * np.exp(1j * cfo * np.arange(n)) * np.sqrt(10 ** (24 / 10))
        gap_front, gap_tail = 100, (1000 if k != 4 else 200)
        parts += [np.zeros(gap_front, np.complex64), sig.astype(np.complex64), np.zeros(gap_tail, np.complex64)]
        psdus.append(psdu[0])
    x = np.concatenate(parts)
    if noise:
        x = x + ((rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size)) * np.sqrt(0.5)).astype(np.complex64)
    return x.astype(np.complex64), psdus


def oracle_stream(orc, x, max_sym=511):
    prm = orc.make_params(max_sym=max_sym)
    o = orc.demod_stream(x, prm, want_eq=True, cap=256)
    psdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=2048)
    return o, psdu


@pytest.mark.parametrize("chunk,batch", [(777, 0), (4096, 0), (100000, 0), (10**7, 0), (777, 20000), (4096, 1 << 20)])
@pytest.mark.parametrize("noise", [True, False])
def test_push_poll_matches_oracle_stream(orc, chunk, batch, noise, decode_path):
    """chunk = samples per push; batch = WIFIRX_P_STREAM_BATCH (0: process every push; otherwise pushes only collect
    until that many samples wait): the frames that come out do not depend on either"""
    from wifirx import capi
    x, psdus = build_stream(noise=noise)
    o, opsdu = oracle_stream(orc, x)
    rx = capi.WifiRx(max_sym=511, want_carrier=True)
    rx.set_param(capi.P_STREAM_BATCH, batch)
    got = []
    for p in range(0, x.size, chunk):
        rx.push(x[p:p + chunk])
        got.append(rx.poll(cap=64, want_idx=True))
        if batch and p + chunk < batch:
            assert len(got[-1]["frames"]) == 0          # still collecting
    rx._check(capi.lib().wifirx_push(rx._h, None, 0, 0))       # flush
    got.append(rx.poll(cap=64, want_idx=True))
    frames = np.concatenate([g["frames"] for g in got])
    psdu = np.concatenate([g["psdu"] for g in got])
    idx = np.concatenate([g["idx"] for g in got])
    car = np.concatenate([g["carrier"] for g in got])
    assert len(frames) == len(o["frames"])
    assert np.array_equal(frames, o["frames"]), (frames, o["frames"])
    for k in range(len(frames)):
        n = int(frames[k]["n_sym_out"])
        assert np.array_equal(idx[k, :n], o["idx"][k, :n])
        assert np.array_equal(car[k, :n], o["eq"][k, :n])
        if frames[k]["flags"] & capi.F_CRC_OK:
            L = int(frames[k]["psdu_len"])
            assert np.array_equal(psdu[k, :L], opsdu[k, :L])
    ok = frames[(frames["flags"] & capi.F_CRC_OK) != 0]
    assert len(ok) == len(psdus)          # every transmitted frame is recovered
    rx.close()


def test_block_publishes_reference_pdus(orc):
    """The GNU-Radio-shaped block: PDU shape and metadata the reference's consumers rely on."""
    from wifirx import block, grshim, app
    x, psdus = build_stream(seed=5)
    got_mac, got_car, got_pics = [], [], []

    class Sink(grshim.basic_block):
        def __init__(self):
            grshim.basic_block.__init__(self, name="sink")
            self.message_port_register_in("in")
            self.message_port_register_in("car")
            self.set_msg_handler("in", got_mac.append)
            self.set_msg_handler("car", got_car.append)

    rx = block.wifi_phy_rx(bandwidth=20e6, frequency=5.89e9, sensitivity=0.56, chan_est=block.LS)
    sink = Sink()
    pics = app.extract_pics(sink=got_pics.append)
    grshim.msg_connect(rx, "mac_out", sink, "in")
    grshim.msg_connect(rx, "carrier", sink, "car")
    grshim.msg_connect(rx, "mac_out", pics, "MAC")
    n = grshim.run_stream(rx, x, chunk=8192)
    assert n == x.size
    assert len(got_mac) == len(psdus)
    for (meta, blob), ref in zip(got_mac, psdus):
        assert blob.dtype == np.uint8 and np.array_equal(blob, ref[:-4])     # MAC frame without FCS
        assert set(meta) >= {"frame_bytes", "encoding", "snr", "freq", "freq_offset", "dlt"}
        assert meta["frame_bytes"] == len(ref) and 15 < meta["snr"] < 35
    assert [len(p) for p in got_pics] == [len(r) - 32 for r in psdus]        # [24:][4:] of the blob
    n_sym_total = sum(txgen.n_sym_for(len(r), e) for r, e in zip(psdus, [0, 2, 4, 7, 2, 5, 3, 6, 1, 2]))
    assert len(got_car) >= n_sym_total and got_car[0][0] == {} and got_car[0][1].shape == (48,)
    rx.set_frequency(2.6e9); rx.set_bandwidth(10e6); rx.set_sensitivity(0.6)
    for eq in (block.LMS, block.COMB, block.STA, block.LS):
        rx.set_chan_est(eq)
    with pytest.raises(Exception):
        rx.set_chan_est(7)


def test_stream_csi_equals_batch_csi_and_reaches_the_pdu(orc):
    """wifirx_poll_csi: the channel state of a frame found in the stream is the LS estimate batch mode (and the
    oracle) compute for the same samples; the block puts it into the mac_out dictionary as upstream's
    frame_equalizer does (`csi`, 52 complex values)"""
    from wifirx import block, capi, grshim
    taps = np.array([[1.0, 0.3 - 0.2j, 0.1j]], dtype=np.complex64)
    psdu = txgen.make_psdus(1, 200, seed=77)
    tx = txgen.encode_psdus(psdu, 4)
    total = ((300 + tx.samples.shape[1] + 700 + 63) // 64) * 64
    x = txgen.impair(tx.samples, 27.0, cfo=0.012, lead=300, total=total, seed=6, taps=taps).reshape(-1)
    o = orc.demod_batch(x, total, orc.make_params(max_sym=tx.n_sym), want_csi=True)
    assert o["frames"]["flags"][0] & orc.F_COMPLETE
    rx = capi.WifiRx(max_sym=511)
    for p in range(0, x.size, 1000):
        rx.push(x[p:p + 1000])
    rx._check(capi.lib().wifirx_push(rx._h, None, 0, 0))
    got = rx.poll(cap=8, want_csi=True)
    rx.close()
    assert len(got["frames"]) == 1 and got["frames"]["flags"][0] & capi.F_CRC_OK
    assert np.array_equal(got["csi"][0].view(np.uint64), o["csi"][0].view(np.uint64))
    pdus = []
    blk = block.wifi_phy_rx(bandwidth=20e6, publish_carrier=False, publish_csi=True)

    class sink(grshim.basic_block):
        def __init__(self):
            grshim.basic_block.__init__(self, name="sink", in_sig=None, out_sig=None)
            self.message_port_register_in(grshim.intern("in"))
            self.set_msg_handler(grshim.intern("in"), lambda m: pdus.append(grshim.to_python(m)))
    s = sink()
    grshim.msg_connect(blk, "mac_out", s, "in")
    grshim.run_stream(blk, x, chunk=4096)
    assert len(pdus) == 1
    meta, blob = pdus[0]
    assert meta["csi"].shape == (52,) and np.array_equal(meta["csi"].view(np.uint64), o["csi"][0].view(np.uint64))
    assert np.array_equal(blob, psdu[0][:-4])


def test_push_after_a_failed_allocation_recovers(orc, monkeypatch):
    """A hipMalloc of the stream path fails once (WIFIRX_TEST_FAIL_ALLOC = which one): that push returns ENOMEM, leaves
    the handle consistent -- no stale capacity over freed buffers -- and the stream, pushed again, delivers exactly
    the frames of an undisturbed run."""
    from wifirx import capi
    x, _ = build_stream(seed=5)
    chunk = 30000
    rx = capi.WifiRx(max_sym=511)
    ref = []
    for p in range(0, x.size, chunk):
        rx.push(x[p:p + chunk])
        ref.append(rx.poll(cap=64, want_idx=True))
    rx._check(capi.lib().wifirx_push(rx._h, None, 0, 0))
    ref.append(rx.poll(cap=64, want_idx=True))
    rx.close()
    ref_frames = np.concatenate([g["frames"] for g in ref])
    ref_psdu = np.concatenate([g["psdu"] for g in ref])
    assert len(ref_frames) == 10
    failures = 0
    for k in range(1, 14):               # 3 sample-side buffers + 5 output buffers, and their first regrowth
        monkeypatch.setenv("WIFIRX_TEST_FAIL_ALLOC", str(k))
        rx = capi.WifiRx(max_sym=511)
        got = []
        for p in range(0, x.size, chunk):
            piece = x[p:p + chunk]
            try:
                rx.push(piece)
            except capi.WifiRxError as e:
                assert e.code == -3 or "hipMalloc" in str(e)          # WIFIRX_ENOMEM
                failures += 1
                # a failed pass is undone as a whole: nothing was consumed, whichever allocation it was -- hand the
                # piece in again (include/wifirx.h, WIFIRX_P_STREAM_BATCH: ERRORS)
                assert rx.push_consumed() == 0
                rx.push(piece)
                assert rx.push_consumed() == piece.size
            got.append(rx.poll(cap=64, want_idx=True))
        rx._check(capi.lib().wifirx_push(rx._h, None, 0, 0))
        got.append(rx.poll(cap=64, want_idx=True))
        rx.close()
        frames = np.concatenate([g["frames"] for g in got])
        psdu = np.concatenate([g["psdu"] for g in got])
        assert np.array_equal(frames, ref_frames), k
        assert np.array_equal(psdu, ref_psdu), k
    assert failures >= 8


def _ref_frames(x, chunk=30000):
    from wifirx import capi
    rx = capi.WifiRx(max_sym=511)
    ref = []
    for p in range(0, x.size, chunk):
        rx.push(x[p:p + chunk])
        ref.append(rx.poll(cap=64, want_idx=True))
    rx.flush()
    ref.append(rx.poll(cap=64, want_idx=True))
    rx.close()
    return np.concatenate([g["frames"] for g in ref]), np.concatenate([g["psdu"] for g in ref])


@pytest.mark.parametrize("batch,chunk", [(20000, 777), (50000, 8192), (20000, 70000)])
def test_worker_batch_failure_loses_nothing(monkeypatch, batch, chunk):
    """The same allocation failures with a batch size set, i.e. on the handle's WORKER thread (the block's default mode):
    the failure is reported by a later push, which has consumed what wifirx_push_consumed says (nothing, unless the call
    spanned several batches); the failed batch stays staged in the library and runs again.  A caller that repeats the
    reported call from there gets exactly the frames of an undisturbed run -- no gap, no doubled sample."""
    from wifirx import capi
    x, _ = build_stream(seed=5)
    ref_frames, ref_psdu = _ref_frames(x)
    assert len(ref_frames) == 10
    failures = 0
    for k in range(1, 14):
        monkeypatch.setenv("WIFIRX_TEST_FAIL_ALLOC", str(k))
        rx = capi.WifiRx(max_sym=511)
        rx.set_param(capi.P_STREAM_BATCH, batch)
        got = []
        p = 0
        while p < x.size:
            piece = x[p:p + chunk]
            try:
                rx.push(piece)
                p += piece.size
            except capi.WifiRxError as e:
                assert e.code == -3, str(e)
                failures += 1
                p += rx.push_consumed()               # repeat from where the library stopped taking samples
            got.append(rx.poll(cap=64, want_idx=True))
        for attempt in range(3):                      # a flush can report a failed batch too: flush again
            try:
                rx.flush()
                break
            except capi.WifiRxError as e:
                assert e.code == -3, str(e)
                failures += 1
        got.append(rx.poll(cap=64, want_idx=True))
        st = rx.stats()
        rx.close()
        frames = np.concatenate([g["frames"] for g in got])
        psdu = np.concatenate([g["psdu"] for g in got])
        assert np.array_equal(frames, ref_frames), k
        assert np.array_equal(psdu, ref_psdu), k
        assert st["samples_in"] == x.size, (k, st)
    assert failures >= 8


def test_block_survives_a_failed_batch(monkeypatch):
    """... and through wifi_phy_rx: with raise_on_error off work() reports what was consumed, the scheduler hands the
    rest in again, every PDU arrives once; with it on (default) the failure is an exception of the block."""
    from wifirx import block, capi, grshim
    x, psdus = build_stream(seed=5)
    for k in (1, 2, 4, 6, 9):
        monkeypatch.setenv("WIFIRX_TEST_FAIL_ALLOC", str(k))
        blk = block.wifi_phy_rx(bandwidth=20e6, publish_carrier=False, batch_samples=20000)
        blk.raise_on_error = False
        got = []
        grshim.msg_connect(blk, "mac_out", grshim.sink_block(got.append), "in")
        pos = grshim.run_stream(blk, x, chunk=4096, finish=False)
        assert pos == x.size
        for attempt in range(3):
            try:
                blk.stop()
                break
            except capi.WifiRxError:
                pass
        assert blk.push_errors >= 1 and "hipMalloc" in blk.last_error, (k, blk.push_errors, blk.last_error)
        assert len(got) == len(psdus), (k, len(got))
        for (meta, blob), want in zip(got, psdus):
            assert np.array_equal(np.asarray(blob), want[:-4])
        blk.close()
    monkeypatch.setenv("WIFIRX_TEST_FAIL_ALLOC", "1")
    blk = block.wifi_phy_rx(bandwidth=20e6, publish_carrier=False, batch_samples=20000)
    with pytest.raises(capi.WifiRxError):
        grshim.run_stream(blk, x, chunk=4096)
    blk.close()


def test_batch_size_change_with_samples_staged_is_no_dead_end(orc):
    """WIFIRX_P_STREAM_BATCH changed while samples sit in the staging buffer (round 2: every later push, the flush
    included, returned EINVAL): the staged samples run as a short batch, the stream goes on, same frames as the oracle."""
    from wifirx import capi
    x, _ = build_stream(seed=7)
    o, opsdu = oracle_stream(orc, x)
    rx = capi.WifiRx(max_sym=511)
    got = []
    sizes = [50000, 20000, 0, 120000, 30000]
    p, i = 0, 0
    while p < x.size:
        rx.set_param(capi.P_STREAM_BATCH, sizes[i % len(sizes)])
        i += 1
        for _ in range(3):
            rx.push(x[p:p + 7000])
            p += 7000
        got.append(rx.poll(cap=64, want_idx=True))
    rx.set_param(capi.P_STREAM_BATCH, 4096)
    rx.flush()
    got.append(rx.poll(cap=64, want_idx=True))
    rx.close()
    frames = np.concatenate([g["frames"] for g in got])
    assert np.array_equal(frames, o["frames"])
    with pytest.raises(capi.WifiRxError):
        rx2 = capi.WifiRx(max_sym=64)
        try:
            rx2.set_param(capi.P_STREAM_BATCH, capi.STREAM_BATCH_MAX + 1)
        finally:
            rx2.close()


def test_batch_calls_interleaved_with_a_stream_batch_in_flight(orc):
    """Batch-mode entry points on a handle whose stream worker has a batch in flight: they wait for it; both
    the stream's frames and the batch results equal the oracle's.  Repeated so that the calls do meet a busy worker."""
    from wifirx import capi
    from helpers import make_slots
    x, _ = build_stream(seed=9)
    o, opsdu = oracle_stream(orc, x)
    iq, slot_len, tx = make_slots(96, 5, psdu_len=200, snr_db=24.0, seed=3)
    prm = orc.make_params(max_sym=tx.n_sym, llr_bits=0)
    ob = orc.demod_batch(iq, slot_len, prm)
    obp = orc.decode_batch(ob["frames"], ob["idx"], prm, psdu_stride=512)
    rx = capi.WifiRx(max_sym=511)
    rx.set_param(capi.P_STREAM_BATCH, 16384)
    for rep in range(4):
        got = []
        for p in range(0, x.size, 16384):
            rx.push(x[p:p + 16384])                   # hands a full batch to the worker and returns
            rb = rx.demod_batch(iq, slot_len, decode=True, psdu_stride=512)
            assert np.array_equal(rb["frames"], ob["frames"]), (rep, p)
            assert np.array_equal(rb["psdu"][:, :200], obp[:, :200])
            got.append(rx.poll(cap=64, want_idx=True))
        rx.flush()
        got.append(rx.poll(cap=64, want_idx=True))
        frames = np.concatenate([g["frames"] for g in got])
        psdu = np.concatenate([g["psdu"] for g in got])
        assert len(frames) == len(o["frames"])
        want = o["frames"]
        if rep == 0:
            assert np.array_equal(frames, want)
        # a further pass over the same samples continues the stream (trigger positions move on, the 16-sample blocks of
        # the window sums sit elsewhere): same frames, same bytes
        assert np.array_equal(frames["flags"], want["flags"]) and np.array_equal(frames["psdu_len"], want["psdu_len"])
        for i in range(len(frames)):
            L = int(frames["psdu_len"][i])
            if frames["flags"][i] & capi.F_CRC_OK:
                assert np.array_equal(psdu[i, :L], opsdu[i, :L])
    rx.close()


@pytest.mark.parametrize("batch", [0, 20000])
def test_failure_behind_the_commit_of_a_pass_doubles_nothing(monkeypatch, batch):
    """the carry step at the end of a stream pass runs AFTER the pass is committed (frames queued, fill and
    frontier advanced).  A device failure there must not come back as a failed push -- every caller answers that by
    handing the same samples in again, which would double them.  WIFIRX_TEST_FAIL_CARRY=k fails the k-th carry before it
    has touched the sample buffer: no error, and exactly the frames of an undisturbed run, in the caller's thread
    (batch = 0) and on the worker thread.  =-k fails it after the move began: the delivered frames stay, the stream is
    dead, every later push says so with push_consumed() = 0 (include/wifirx.h, WIFIRX_P_STREAM_BATCH: ERRORS)."""
    from wifirx import capi
    x, _ = build_stream(seed=5)
    ref_frames, ref_psdu = _ref_frames(x)
    chunk = 4096                             # eleven passes (batch = 0) / three batches: several carries
    for k in (1, 2, 3, 5):
        monkeypatch.setenv("WIFIRX_TEST_FAIL_CARRY", str(k))
        rx = capi.WifiRx(max_sym=511)
        if batch:
            rx.set_param(capi.P_STREAM_BATCH, batch)
        got = []
        for p in range(0, x.size, chunk):
            rx.push(x[p:p + chunk])                     # never raises
            assert rx.push_consumed() == min(chunk, x.size - p)
            got.append(rx.poll(cap=64, want_idx=True))
        rx.flush()
        got.append(rx.poll(cap=64, want_idx=True))
        rx.close()
        assert np.array_equal(np.concatenate([g["frames"] for g in got]), ref_frames), k
        assert np.array_equal(np.concatenate([g["psdu"] for g in got]), ref_psdu), k
    monkeypatch.setenv("WIFIRX_TEST_FAIL_CARRY", "-2")
    rx = capi.WifiRx(max_sym=511)
    if batch:
        rx.set_param(capi.P_STREAM_BATCH, batch)
    got, dead_at = [], None
    for p in list(range(0, x.size, chunk)) + [None]:        # the pushes, then the flush (a worker's failure is reported by the NEXT call)
        try:
            if p is None:
                rx.flush()
            else:
                rx.push(x[p:p + chunk])
        except capi.WifiRxError as e:
            assert e.code == capi.EDEAD and "lost" in str(e), str(e)
            assert rx.push_consumed() == 0
            dead_at = p if p is not None else x.size
            break
        got.append(rx.poll(cap=64, want_idx=True))
    assert dead_at is not None
    with pytest.raises(capi.WifiRxError):               # and it stays dead
        rx.push(x[:chunk])
    got.append(rx.poll(cap=64, want_idx=True))          # what was delivered before stays polled / pollable
    rx.close()
    frames = np.concatenate([g["frames"] for g in got])
    assert len(frames) >= 1 and np.array_equal(frames, ref_frames[:len(frames)])


def test_block_ends_on_a_dead_stream(monkeypatch):
    """a dead stream answers every push with WIFIRX_EDEAD and 0 consumed samples.  work() used to count the error and
    return 0 items with raise_on_error off -- the scheduler hands the same items in again, for ever.  Now it publishes what is
    finished and returns WORK_DONE (-1) -- or raises, with raise_on_error --, and stop() makes no flush that cannot succeed."""
    from wifirx import block, capi, grshim
    x, psdus = build_stream(seed=5)
    for raise_on_error in (False, True):
        monkeypatch.setenv("WIFIRX_TEST_FAIL_CARRY", "-2")
        blk = block.wifi_phy_rx(bandwidth=20e6, publish_carrier=False, batch_samples=0)
        blk.raise_on_error = raise_on_error
        got = []
        grshim.msg_connect(blk, "mac_out", grshim.sink_block(got.append), "in")
        calls = []
        work = blk.work
        blk.work = lambda i, o: (calls.append(1), work(i, o))[1]
        if raise_on_error:
            with pytest.raises(capi.WifiRxError) as ei:
                grshim.run_stream(blk, x, chunk=4096, finish=False)
            assert ei.value.code == capi.EDEAD
        else:
            pos = grshim.run_stream(blk, x, chunk=4096, finish=False)
            assert pos < x.size
        assert blk.stream_dead and "lost" in blk.last_error
        n_calls, n_err = len(calls), blk.push_errors
        assert n_err == 1                            # one failed push, not a spin of them
        assert blk.work([x[:4096]], []) == -1 if not raise_on_error else True
        blk.stop()                                   # no flush attempts on a dead stream
        assert blk.push_errors <= n_err + 1
        assert 1 <= len(got) <= len(psdus)           # what was finished before the failure was published
        for (meta, blob), want in zip(got, psdus):
            assert np.array_equal(np.asarray(blob), want[:-4])
        blk.close()


def test_block_stop_flushes_and_publishes_through_a_pending_worker_error(monkeypatch):
    """stop() used to raise on the worker's not-yet-reported error before flushing or publishing -- the frames
    in the queue and the batch to be run again never reached mac_out, and raise_on_error = False was ignored.  One stop()
    now settles everything; with raise_on_error the exception comes AFTER the PDUs are out."""
    from wifirx import block, capi, grshim
    x, psdus = build_stream(seed=5)
    for raise_on_error in (False, True):
        monkeypatch.setenv("WIFIRX_TEST_FAIL_ALLOC", "6")
        blk = block.wifi_phy_rx(bandwidth=20e6, publish_carrier=False, batch_samples=1 << 22)     # one batch: it fails inside stop()'s flush
        blk.raise_on_error = raise_on_error
        got = []
        grshim.msg_connect(blk, "mac_out", grshim.sink_block(got.append), "in")
        assert grshim.run_stream(blk, x, chunk=4096, finish=False) == x.size
        blk.stop()                                   # never raises here: the second flush runs the failed batch again
        assert len(got) == len(psdus), (raise_on_error, len(got), blk.push_errors, blk.last_error)
        assert blk.push_errors >= 1 and "hipMalloc" in blk.last_error
        for (meta, blob), want in zip(got, psdus):
            assert np.array_equal(np.asarray(blob), want[:-4])
        blk.close()
