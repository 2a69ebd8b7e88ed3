"""N > 1 path on the CPU: world_size-2 gloo run of the sharding + PDU all-gather plumbing that
bench.py --gpus N uses (RCCL on the GPUs).  The compute itself needs a GPU and is covered by -m gpu."""
import os
import socket

import numpy as np
import pytest

from wifirx import dist as wdist


def test_shard_ranges_cover_and_are_contiguous():
    for n in (0, 1, 7, 8, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            spans = [wdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
            assert max(h - l for l, h in spans) <= wdist.padded_shard(n, w)


def _worker(rank, world, port, n_frames, q):
    import torch
    import torch.distributed as dist
    from wifirx import capi, txgen
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = wdist.shard_range(n_frames, rank, world)
    per = wdist.padded_shard(n_frames, world)
    # what a rank holds after wifirx_decode_batch: PSDU buffer + frame records of its shard
    psdu_all = txgen.make_psdus(n_frames, 60, seed=5)
    psdu = np.zeros((per, 64), np.uint8)
    fr = np.zeros(per, capi.FRAME_DTYPE)
    psdu[:hi - lo, :60] = psdu_all[lo:hi]
    fr["flags"][:hi - lo] = capi.F_DETECTED | capi.F_SYNC | capi.F_SIGNAL | capi.F_COMPLETE | capi.F_DECODED | capi.F_CRC_OK
    fr["psdu_len"][:hi - lo] = 60
    if rank == 1 and hi - lo > 2:
        fr["flags"][2] &= ~np.uint32(capi.F_CRC_OK)        # one frame of rank 1 failed its FCS
    p_t = torch.from_numpy(psdu)
    f_t = torch.from_numpy(fr.view(np.uint8).reshape(per, 32))
    pa, fa = wdist.all_gather_pdus(p_t, f_t)
    pdus = wdist.pdus_from_gathered(pa, fa, n_frames, world)
    q.put((rank, [(k, bytes(b)) for k, b in pdus]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_all_gather_of_pdus_world2():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_frames = 11
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    from wifirx import txgen
    ref = txgen.make_psdus(n_frames, 60, seed=5)
    lo1, _ = wdist.shard_range(n_frames, 1, 2)
    expect = [(k, bytes(ref[k, :56])) for k in range(n_frames) if k != lo1 + 2]
    assert res[0] == expect and res[1] == expect       # every rank ends with the whole PDU stream, in frame order


@pytest.mark.timeout(300)
def test_bench_launches_its_own_ranks_world2():
    """`python bench.py --gpus 2` with no torchrun around it: the parent starts two ranks itself (before touching a
    GPU), the ranks run the timed-loop bookkeeping and the chunked PDU all-gather over gloo with a stand-in compute
    step (WIFIRX_BENCH_STUB=1), rank 0 prints the one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WIFIRX_BENCH_STUB="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--frames", "1000", "--gather-chunks", "3"], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # one JSON line, from rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["data"] == "stub"
    assert j["gather_consistent"] is True and j["pdus_gathered"] == 2000 and j["gather_chunks"] == 3
    # the fields a later SCALE run is checked by: the collective library saw both ranks
    c = j["collective"]
    assert c["backend"] == "gloo" and c["world_size_seen"] == 2 and c["dist_world_size"] == 2
    assert sorted(h["rank"] for h in c["ranks_hosts"]) == [0, 1] and len({h["pid"] for h in c["ranks_hosts"]}) == 2


@pytest.mark.timeout(600)
def test_bench_launches_its_own_ranks_world8():
    """the same rehearsal at the rank count of BASELINE.json configs[3] (8 GPUs, 1 M frames each) -- eight gloo
    ranks on the CPU: one JSON line, the collective saw eight ranks in eight processes, every rank ends with the 8 x n_frames PDUs
    (320-byte rows + 32-byte records) in global frame order, and the ranks draw distinct payloads and distinct noise."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WIFIRX_BENCH_STUB="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    n = 600
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--frames", str(n), "--gather-chunks", "4"], env=env, capture_output=True, text=True, timeout=560)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["data"] == "stub" and j["scaling"] == "weak"
    assert j["gather_consistent"] is True and j["pdus_gathered"] == 8 * n and j["gather_chunks"] == 4
    assert j["gathered_bytes_per_rank"] == 8 * (320 + 32) * n
    c = j["collective"]
    assert c["backend"] == "gloo" and c["world_size_seen"] == 8 and c["dist_world_size"] == 8
    assert sorted(h["rank"] for h in c["ranks_hosts"]) == list(range(8)) and len({h["pid"] for h in c["ranks_hosts"]}) == 8
    pr = j["per_rank"]
    assert [r["rank"] for r in pr] == list(range(8))
    assert len({r["payload_seed"] for r in pr}) == 8 and len({r["synth_seed"] for r in pr}) == 8


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WIFIRX_BENCH_STUB="1", WORLD_SIZE="4", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr


def test_chunked_gather_row_order():
    """frame_order() maps (rank, local frame) to its row of the [chunk][rank][frame] buffers, for even and ragged chunks"""
    class G(wdist.ChunkedPduGather):
        def __init__(self, n_local, n_chunks, world):       # layout arithmetic only: no process group
            self.world = world
            self._layout(n_local, n_chunks)
    for n_local, n_chunks, world in ((12, 3, 2), (11, 3, 2), (1000, 7, 8), (5, 8, 3)):
        g = G(n_local, n_chunks, world)
        o = g.frame_order()
        assert o.shape == (world, n_local) and len(np.unique(o)) == world * n_local
        assert o.max() < g.n_chunks * world * g.cf
        for c in range(g.n_chunks):
            lo, hi = g.chunk_range(c)
            for r in range(world):               # chunk c of rank r is one contiguous block of the gathered buffer
                assert np.array_equal(o[r, lo:hi], (c * world + r) * g.cf + np.arange(hi - lo))
