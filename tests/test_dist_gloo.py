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
