"""Multi-GPU execution of the receive chain (SURVEY.md section 8 row e).

Frames (slots) are independent, so a batch shards into contiguous frame ranges, one process and one
HIP stream per GPU, with NO data-path collective.  The only exchange is the one the north star asks
for: after decode_mac every rank all-gathers the fixed-stride PSDU buffer and the 32-byte frame
records, so that each rank ends up with the PDU stream of the whole batch in frame order.
``torch.distributed`` is plumbing here (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU
tests); the tensors are plain byte buffers.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_frames: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous frame range [lo, hi) of `rank`; the first n_frames % world ranks get one extra."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def padded_shard(n_frames: int, world: int) -> int:
    """Frames per rank after padding to a common size (all_gather needs equal shapes)."""
    return (n_frames + world - 1) // world


def all_gather_pdus(psdu, frames, group=None):
    """psdu: uint8 tensor [n_local, stride]; frames: uint8 tensor [n_local, 32] (wifirx_frame records).
    Returns (psdu_all [world*n_local, stride], frames_all [world*n_local, 32]) in rank order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    psdu_all = torch.empty((world * psdu.shape[0],) + tuple(psdu.shape[1:]), dtype=psdu.dtype, device=psdu.device)
    frames_all = torch.empty((world * frames.shape[0],) + tuple(frames.shape[1:]), dtype=frames.dtype, device=frames.device)
    dist.all_gather_into_tensor(psdu_all, psdu.contiguous(), group=group)
    dist.all_gather_into_tensor(frames_all, frames.contiguous(), group=group)
    return psdu_all, frames_all


def pdus_from_gathered(psdu_all, frames_all, n_frames: int, world: int):
    """Host-side view of the gathered buffers: list of (global frame index, bytes without FCS) for the
    frames whose FCS was good, dropping the padding frames at the end of every shard."""
    from .capi import FRAME_DTYPE, F_CRC_OK
    fr = np.ascontiguousarray(frames_all.cpu().numpy()).view(FRAME_DTYPE).reshape(-1)
    ps = psdu_all.cpu().numpy()
    per = padded_shard(n_frames, world)
    out = []
    for r in range(world):
        lo, hi = shard_range(n_frames, r, world)
        for j in range(hi - lo):
            k = r * per + j
            if fr[k]["flags"] & F_CRC_OK:
                out.append((lo + j, bytes(ps[k, :int(fr[k]["psdu_len"]) - 4])))
    return out


class ChunkedPduGather:
    """The PDU all-gather of ``bench.py --gpus N`` and of any multi-GPU caller: buffers allocated ONCE, the batch cut
    into `n_chunks` frame ranges so that the collective of chunk c runs (on the process group's own stream) while
    decode_mac works on chunk c + 1.

    Layout of the gathered buffers: ``[chunk][rank][frame in chunk]`` (an all-gather writes one contiguous block per
    call); global frame k of rank r sits at chunk ``j // cf``, row ``j % cf`` with ``j`` its index inside the rank's
    shard and ``cf = frames_per_chunk``.  :meth:`frame_order` returns the permutation back to rank-major frame order.
    """

    def __init__(self, n_local: int, stride: int, n_chunks: int, device, group=None):
        import torch
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.n_local, self.stride = int(n_local), int(stride)
        self.n_chunks = max(1, min(int(n_chunks), self.n_local)) if self.n_local else 1
        self.cf = (self.n_local + self.n_chunks - 1) // self.n_chunks if self.n_local else 0
        self.n_chunks = (self.n_local + self.cf - 1) // self.cf if self.cf else 1
        rows = self.n_chunks * self.world * self.cf
        self.psdu_all = torch.zeros((rows, self.stride), dtype=torch.uint8, device=device)
        self.frames_all = torch.zeros((rows, 32), dtype=torch.uint8, device=device)
        # staging for a last chunk shorter than cf (all_gather needs equal shapes on every rank)
        self._tail_p = torch.zeros((self.cf, self.stride), dtype=torch.uint8, device=device)
        self._tail_f = torch.zeros((self.cf, 32), dtype=torch.uint8, device=device)
        self._works = []

    def chunk_range(self, c: int):
        lo = c * self.cf
        return lo, min(self.n_local, lo + self.cf)

    def gather_chunk(self, c: int, psdu, frames, async_op=True):
        """psdu [n_local, stride], frames [n_local, 32]: the rank's own buffers; chunk c of them is exchanged."""
        import torch.distributed as dist
        lo, hi = self.chunk_range(c)
        p, f = psdu[lo:hi], frames[lo:hi]
        if hi - lo < self.cf:
            self._tail_p.zero_(); self._tail_f.zero_()
            self._tail_p[:hi - lo].copy_(p); self._tail_f[:hi - lo].copy_(f)
            p, f = self._tail_p, self._tail_f
        r0 = c * self.world * self.cf
        r1 = r0 + self.world * self.cf
        w1 = dist.all_gather_into_tensor(self.psdu_all[r0:r1], p.contiguous(), group=self.group, async_op=async_op)
        w2 = dist.all_gather_into_tensor(self.frames_all[r0:r1], f.contiguous(), group=self.group, async_op=async_op)
        if async_op:
            self._works += [w1, w2]

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []

    def frame_order(self) -> np.ndarray:
        """row index in the gathered buffers of (rank r, local frame j), as an array [world, n_local]"""
        j = np.arange(self.n_local)
        c, i = j // max(self.cf, 1), j % max(self.cf, 1)
        r = np.arange(self.world)[:, None]
        return (c[None, :] * self.world + r) * self.cf + i[None, :]

    def pdus(self, n_frames_total=None):
        """list of (global frame index, bytes without FCS) of the frames with a good FCS, in frame order; ranks hold
        equal shards of n_local frames (bench.py) unless n_frames_total says the batch was split by shard_range()"""
        from .capi import FRAME_DTYPE, F_CRC_OK
        fr = np.ascontiguousarray(self.frames_all.cpu().numpy()).view(FRAME_DTYPE).reshape(-1)
        ps = self.psdu_all.cpu().numpy()
        order = self.frame_order()
        out = []
        for r in range(self.world):
            if n_frames_total is None:
                lo, cnt = r * self.n_local, self.n_local
            else:
                lo, hi = shard_range(n_frames_total, r, self.world)
                cnt = hi - lo
            for j in range(cnt):
                k = order[r, j]
                if fr[k]["flags"] & F_CRC_OK:
                    out.append((lo + j, bytes(ps[k, :int(fr[k]["psdu_len"]) - 4])))
        return out
