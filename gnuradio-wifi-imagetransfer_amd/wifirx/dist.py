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
