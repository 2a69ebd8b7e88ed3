"""A continuous recording cut across ranks (the reference's receive side is one stream with sync_short
state across it, gnu_radio/IRS_AP.py:163-177,268): ownership ranges, pre-roll, halo, duplicate removal by absolute trigger
index, the seam check -- on the CPU, with the ORACLE's stream driver standing in for the GPU engine (test infrastructure:
the product's engine is wifirx.dist.gpu_stream_engine; tests/test_gpu_recording.py runs that one on the MI355X)."""
import os
import socket

import numpy as np
import pytest

from wifirx import dist as wdist, txgen


def test_stream_shards_tile_the_recording():
    for n in (0, 1, 63, 64, 65, 100000, 10_000_019):
        for w in (1, 2, 3, 8):
            sh = wdist.stream_shards(n, w)
            assert sh[0][0] == 0 and sh[-1][1] == n
            for (lo, hi, rlo, rhi), nxt in zip(sh, sh[1:] + [None]):
                assert (lo % 64 == 0 or lo == n) and rlo % 64 == 0 and rlo <= lo <= hi <= rhi <= max(n, 0)
                assert rhi == min(n, hi + wdist.STREAM_HALO)
                assert lo - rlo <= wdist.STREAM_PRE_ROLL + 63
                if nxt:
                    assert hi == nxt[0]
    assert wdist.STREAM_HALO >= 43200 + 320            # one maximal frame + sync_long's look-ahead


def test_unwrap_and_ownership():
    t = np.array([5, 100, (1 << 31) - 3, 7, 900], np.int32)            # wrapped once after the third
    a = wdist.unwrap_triggers(t, 1000)
    assert a.tolist() == [1005, 1100, (1 << 31) + 997, (1 << 31) + 1007, (1 << 31) + 1900]
    assert wdist.owned(a, 1100, (1 << 31) + 1007).tolist() == [False, True, True, False, False]


def _state_machine(cands, min_gap=wdist.WIFIRX_MIN_GAP, start=None):
    """sync_short's trigger selection over plateau candidates: accepted iff more than min_gap after the last accepted"""
    out, last = [], (-(1 << 40) if start is None else start)
    for c in cands:
        if c - last > min_gap:
            out.append(c)
            last = c
    return np.array(out, np.int64)


@pytest.mark.parametrize("seed", range(6))
def test_seam_check_on_a_model_of_the_state_machine(seed):
    """Random plateau-candidate sequences incl. long unbroken re-trigger chains: a rank that starts mid-stream either
    reproduces the uncut run's triggers in its range, or the seam check says so and a longer pre-roll mends it."""
    rng = np.random.default_rng(seed)
    n = 3_000_000
    cands = []
    p = 0
    while p < n:
        kind = rng.integers(0, 3)
        if kind == 0:                                   # isolated frames
            p += int(rng.integers(600, 30000)); cands.append(p)
        elif kind == 1:                                 # a burst of plateau hits a few samples apart (one preamble)
            p += int(rng.integers(600, 5000))
            for _ in range(int(rng.integers(2, 40))):
                p += int(rng.integers(1, 6)); cands.append(p)
        else:                                           # a jammer: candidates every few samples for up to 400 000 samples
            end = p + int(rng.integers(1000, 400000))
            while p < end:
                p += int(rng.integers(1, 300)); cands.append(p)
    cands = np.array([c for c in cands if c < n], np.int64)
    truth = _state_machine(cands)
    longer = 0
    for world in (2, 3, 5):
        prev = None
        for r in range(world):
            pre = wdist.STREAM_PRE_ROLL
            while True:
                lo, hi, rlo, rhi = wdist.stream_shards(n, world, pre)[r]
                mine = _state_machine(cands[(cands >= (rlo + 80 if rlo else 0)) & (cands < rhi)])   # after a cut the detector needs 80 samples to settle
                ok = r == 0 or wdist.seam_consistent(prev, mine, rlo, lo)
                if ok or rlo == 0:
                    break
                pre *= 2
            got = mine[wdist.owned(mine, lo, hi)]
            want = truth[wdist.owned(truth, lo, hi)]
            assert np.array_equal(got, want), (seed, world, r, pre)
            longer += pre > wdist.STREAM_PRE_ROLL
            prev = mine
    if seed == 0:
        assert longer > 0          # this seed has a seam inside a re-trigger chain longer than the default pre-roll


def _recording(seed=3, n_frames=60):
    """frames back to back with packet_pad2-like gaps, some gaps shorter than MIN_GAP, mixed rates / lengths"""
    rng = np.random.default_rng(seed)
    parts = [np.zeros(int(rng.integers(50, 400)), np.complex64)]
    for k in range(n_frames):
        enc, plen = int(rng.integers(0, 8)), int(rng.integers(40, 400))
        tx = txgen.encode_psdus(txgen.make_psdus(1, plen, seed=seed * 1000 + k, seq0=k), enc, seeds=[(k % 127) + 1])
        nsmp = tx.samples.shape[1]
        sig = tx.samples[0] * np.exp(1j * rng.uniform(-0.03, 0.03) * np.arange(nsmp)) * np.sqrt(10 ** 2.2)
        parts += [sig.astype(np.complex64), np.zeros(int(rng.choice([0, 120, 300, 1000, 2500])), np.complex64)]
    x = np.concatenate(parts)
    return (x + ((rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size)) * np.sqrt(0.5))).astype(np.complex64)


def _oracle_engine(orc):
    def run(x):
        prm = orc.make_params(max_sym=200)
        o = orc.demod_stream(x, prm, cap=4096)
        psdu = orc.decode_batch(o["frames"], o["idx"], prm, psdu_stride=512)
        return o["frames"], psdu
    return run


@pytest.mark.parametrize("world", [2, 3, 4])
def test_recording_cut_across_ranks_equals_one_rank(orc, world):
    """frames, records (trigger index included) and PSDUs of the cut recording == the uncut run's, in stream order"""
    x = _recording()
    eng = _oracle_engine(orc)
    one_f, one_p = eng(x)
    assert len(one_f) >= 55
    # a pre-roll far shorter than the default, so that the few-hundred-thousand-sample test recording is really cut
    # inside frames and re-trigger chains (the default spans two maximal frames)
    for pre in (4096, wdist.STREAM_PRE_ROLL):
        res = wdist.demod_recording_sharded(x, eng, world, pre_roll=pre)
        assert [p["own"] for p in res["parts"]] == [s[:2] for s in wdist.stream_shards(x.size, world, pre)]
        assert np.array_equal(res["frames"], one_f), (world, pre)
        assert np.array_equal(res["psdu"][:, :512], one_p), (world, pre)
        assert sum(len(p["frames"]) for p in res["parts"]) == len(one_f)              # nothing twice, nothing lost


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = _recording(seed=4, n_frames=40)
    res = wdist.gather_recording(x, _oracle_engine(orc), stride=512, pre_roll=4096)
    q.put((rank, res["frames"].tobytes(), res["psdu"].tobytes(), res["counts"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gather_recording_world2_gloo(orc):
    """two processes, gloo: every rank ends with the whole recording's frames and PSDUs in stream order"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, f, ps, cnt = q.get(timeout=150)
        res[r] = (f, ps, cnt)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    x = _recording(seed=4, n_frames=40)
    one_f, one_p = _oracle_engine(orc)(x)
    assert res[0][0] == res[1][0] == one_f.tobytes()
    assert res[0][1] == res[1][1] == np.ascontiguousarray(one_p[:, :512]).tobytes()
    assert sum(res[0][2]) == len(one_f) and min(res[0][2]) > 0


def _worker_uneven(rank, world, port, n_frames, q):
    import torch
    import torch.distributed as dist
    from wifirx import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = wdist.shard_range(n_frames, rank, world)
    ref = txgen.make_psdus(n_frames, 60, seed=5)
    psdu = np.zeros((hi - lo, 64), np.uint8)
    psdu[:, :60] = ref[lo:hi]
    fr = np.zeros(hi - lo, capi.FRAME_DTYPE)
    fr["flags"] = capi.F_DETECTED | capi.F_SYNC | capi.F_SIGNAL | capi.F_COMPLETE | capi.F_DECODED | capi.F_CRC_OK
    fr["psdu_len"] = 60
    g = wdist.ChunkedPduGather(hi - lo, 64, 3, "cpu")
    for c in range(g.n_chunks):
        g.gather_chunk(c, torch.from_numpy(psdu), torch.from_numpy(fr.view(np.uint8).reshape(-1, 32)), async_op=False)
    q.put((rank, g.pdus(n_frames)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_chunked_gather_with_unequal_shards_world2():
    """11 frames over 2 ranks = shards of 6 and 5 (mismatched collective shapes would hang): the layout
    follows the larger shard, the shorter one pads"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, 2, port, 11, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    ref = txgen.make_psdus(11, 60, seed=5)
    expect = [(k, bytes(ref[k, :56])) for k in range(11)]
    assert res[0] == expect and res[1] == expect
