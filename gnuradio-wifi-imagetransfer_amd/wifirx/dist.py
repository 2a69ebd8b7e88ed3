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
        self.n_own, self.stride = int(n_local), int(stride)
        # Shards may differ by a frame (shard_range): every rank must issue the same number of collectives with the
        # same shapes, so the layout follows the LARGEST shard and shorter ones are padded with zero rows.
        t = torch.tensor([self.n_own], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        self._layout(int(t.item()), n_chunks)
        rows = self.n_chunks * self.world * self.cf
        self.psdu_all = torch.zeros((rows, self.stride), dtype=torch.uint8, device=device)
        self.frames_all = torch.zeros((rows, 32), dtype=torch.uint8, device=device)
        # staging for a chunk this rank cannot fill (the last one, or a shard shorter than the largest)
        self._tail_p = torch.zeros((self.cf, self.stride), dtype=torch.uint8, device=device)
        self._tail_f = torch.zeros((self.cf, 32), dtype=torch.uint8, device=device)
        self._works = []

    def _layout(self, n_local: int, n_chunks: int):
        """n_local = frames per rank of the layout (the largest shard)"""
        self.n_local = int(n_local)
        self.n_chunks = max(1, min(int(n_chunks), self.n_local)) if self.n_local else 1
        self.cf = (self.n_local + self.n_chunks - 1) // self.n_chunks if self.n_local else 0
        self.n_chunks = (self.n_local + self.cf - 1) // self.cf if self.cf else 1

    def chunk_range(self, c: int):
        lo = c * self.cf
        return lo, min(self.n_local, lo + self.cf)

    def gather_chunk(self, c: int, psdu, frames, async_op=True):
        """psdu [n_own, stride], frames [n_own, 32]: the rank's own buffers; chunk c of them is exchanged."""
        import torch.distributed as dist
        lo, hi = self.chunk_range(c)
        n_own = getattr(self, "n_own", self.n_local)
        lo_o, hi_o = min(lo, n_own), min(hi, n_own)
        p, f = psdu[lo_o:hi_o], frames[lo_o:hi_o]
        if hi_o - lo_o < self.cf:
            self._tail_p.zero_(); self._tail_f.zero_()
            if hi_o > lo_o:
                self._tail_p[:hi_o - lo_o].copy_(p); self._tail_f[:hi_o - lo_o].copy_(f)
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
        equal shards of n_local frames (bench.py) unless n_frames_total says the batch was split by shard_range() (the
        layout then follows the largest shard, rows behind a shorter shard are zero padding and skipped here)"""
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


# ----------------------------------------------------------------------------------------------------------------------
# A CONTINUOUS recording cut across ranks ("shard boundaries need a >= one-max-frame overlap halo").
#
# The reference's receive side is one continuous stream (gnu_radio/IRS_AP.py:163-177,314-316) and sync_short carries
# state across it (:268): a plateau re-triggers only more than MIN_GAP = 480 copied samples after the previous trigger,
# and a trigger owns the samples up to the next one (at most MAX_SAMPLES = 43 200).  So a recording does not cut into
# independent slots.  What does hold:
#   * the window sums of the detector look back 64 + 16 samples and have no running state (DESIGN.md rule 3: blocked
#     sums, no subtraction), so from 80 samples after a cut ON A 64-SAMPLE BOUNDARY every A[n], P[n] is bit-identical
#     to the uncut stream's;
#   * the state machine's whole memory is the position of the last accepted trigger;
#   * a frame needs the samples from 16 before its trigger to the next trigger, MAX_SAMPLES at most.
# Rank r OWNS the triggers at absolute positions [own_lo, own_hi) and runs the ordinary stream mode over
# [own_lo - pre_roll, own_hi + HALO): the pre-roll brings the state machine into the right state at own_lo, the halo
# holds every sample an owned frame can use and shows the trigger that may cut the last owned frame short.  Frames whose
# trigger lies outside [own_lo, own_hi) are dropped (their owner delivers them): no duplicates, by absolute trigger index.
# Whether the pre-roll was long enough is CHECKED, not assumed: ranks r - 1 and r must have accepted the same triggers in
# the part of the pre-roll both saw in steady state; if not (an unbroken chain of re-triggers longer than the pre-roll,
# e.g. a jammer repeating the short preamble), rank r doubles its pre-roll and runs again -- at worst from sample 0, which
# is the one-rank run.
WIFIRX_MAX_SAMPLES = 540 * 80
WIFIRX_MIN_GAP = 480
STREAM_HALO = WIFIRX_MAX_SAMPLES + 320 + 64          # the longest frame's samples + sync_long's look-ahead, rounded up
STREAM_PRE_ROLL = 2 * WIFIRX_MAX_SAMPLES             # default: two maximal frames
_SETTLE = 80 + 64                                      # samples after a cut before detector outputs are the uncut stream's


def stream_shards(n_samples: int, world: int, pre_roll: int = STREAM_PRE_ROLL, halo: int = STREAM_HALO):
    """[(own_lo, own_hi, read_lo, read_hi)] per rank: ownership ranges tile [0, n_samples) on 64-sample boundaries;
    the read range adds the pre-roll in front (cut on a 64-sample boundary, so that the detector's 16-sample blocks sit
    where they sit in the uncut stream) and the halo behind."""
    tiles = (n_samples + 63) // 64
    out = []
    for r in range(world):
        t_lo, t_hi = shard_range(tiles, r, world)
        own_lo, own_hi = min(n_samples, t_lo * 64), min(n_samples, t_hi * 64)
        if r == world - 1:
            own_hi = n_samples
        read_lo = max(0, (own_lo - pre_roll) // 64 * 64)
        read_hi = min(n_samples, own_hi + halo)
        out.append((own_lo, own_hi, read_lo, read_hi))
    return out


def unwrap_triggers(trig31: np.ndarray, read_lo: int) -> np.ndarray:
    """wifirx_frame.trigger of the stream mode is the position modulo 2^31, in stream order: absolute positions again."""
    t = np.asarray(trig31, dtype=np.int64)
    if t.size == 0:
        return t
    wraps = np.concatenate([[0], np.cumsum(np.diff(t) < 0)])
    return t + wraps * (1 << 31) + read_lo


def owned(trig_abs: np.ndarray, own_lo: int, own_hi: int) -> np.ndarray:
    return (trig_abs >= own_lo) & (trig_abs < own_hi)


def seam_consistent(prev_trig_abs: np.ndarray, my_trig_abs: np.ndarray, my_read_lo: int, my_own_lo: int) -> bool:
    """Did rank r enter its own range in the state the uncut stream has there?  Both ranks saw [my_read_lo, my_own_lo);
    from _SETTLE samples in (detector settled) plus MIN_GAP (any trigger the cut could have suppressed or invented has
    lost its influence unless it started a chain) they must have accepted exactly the same triggers.  The state at
    own_lo is the last accepted trigger: equal lists in that window => equal state, given one trigger-free stretch of
    more than MIN_GAP or one common trigger in it; an empty window shorter than that proves nothing."""
    if my_read_lo == 0:
        return True                                    # rank r started where the stream starts: nothing to check
    lo = my_read_lo + _SETTLE + WIFIRX_MIN_GAP
    if lo >= my_own_lo:
        return False
    a = prev_trig_abs[(prev_trig_abs >= lo) & (prev_trig_abs < my_own_lo)]
    b = my_trig_abs[(my_trig_abs >= lo) & (my_trig_abs < my_own_lo)]
    if a.size != b.size or not np.array_equal(a, b):
        return False
    if a.size:
        return True                                    # a common accepted trigger: from there on the chains coincide
    return (my_own_lo - lo) > WIFIRX_MIN_GAP          # no trigger for more than MIN_GAP: the state is "free" in both


def gpu_stream_engine(**rx_kwargs):
    """engine(samples) -> (frames, psdu rows) through the library's stream mode (one handle per call)."""
    def run(x):
        from . import capi
        rx = capi.WifiRx(**rx_kwargs)
        try:
            rx.set_param(capi.P_STREAM_BATCH, 1 << 22)
            rx.set_param(capi.P_STREAM_IDX, 0)
            got = []
            step = 1 << 22
            for p in range(0, x.size, step):
                rx.push(x[p:p + step])
                got.append(rx.poll(cap=4096))
            rx.flush()
            while True:
                g = rx.poll(cap=4096)
                got.append(g)
                if len(g["frames"]) == 0:
                    break
            frames = np.concatenate([g["frames"] for g in got])
            psdu = np.concatenate([g["psdu"] for g in got])
            return frames, psdu
        finally:
            rx.close()
    return run


def demod_recording_shard(x, rank: int, world: int, engine, prev_triggers=None, pre_roll: int = STREAM_PRE_ROLL):
    """Rank `rank`'s part of a recording `x` (complex64, the whole recording or a memory map of it -- only the read range
    is touched).  engine(samples) -> (frames [n] FRAME_DTYPE in stream order, psdu [n, stride]).
    Returns dict(frames, psdu, trig_abs) of the OWNED frames (trigger field made absolute, modulo 2^31 as in a one-rank
    run), plus all_trig_abs (every trigger the rank accepted, for the neighbour's seam check) and the ranges used.
    prev_triggers: the absolute triggers rank - 1 accepted (None: skip the seam check -- the caller does it, see
    demod_recording_sharded)."""
    n = int(x.shape[0])
    while True:
        own_lo, own_hi, read_lo, read_hi = stream_shards(n, world, pre_roll)[rank]
        frames, psdu = engine(np.ascontiguousarray(x[read_lo:read_hi]))
        trig = unwrap_triggers(frames["trigger"], read_lo)
        ok = prev_triggers is None or rank == 0 or seam_consistent(np.asarray(prev_triggers, np.int64), trig, read_lo, own_lo)
        if ok or read_lo == 0:
            break
        pre_roll *= 2                                  # the chain of re-triggers was longer than the pre-roll: look further back
    keep = owned(trig, own_lo, own_hi)
    fr = frames[keep].copy()
    fr["trigger"] = (trig[keep] & 0x7fffffff).astype(np.int32)
    return dict(frames=fr, psdu=psdu[keep], trig_abs=trig[keep], all_trig_abs=trig, own=(own_lo, own_hi),
                read=(read_lo, read_hi), pre_roll=pre_roll, seam_ok=bool(ok))


def demod_recording_sharded(x, engine, world: int, pre_roll: int = STREAM_PRE_ROLL):
    """All ranks in one process, one after the other (tests, and the single-GPU rehearsal of the arithmetic): the PDU
    stream of the whole recording in stream order.  Multi-process form: gather_recording()."""
    parts, prev = [], None
    for r in range(world):
        p = demod_recording_shard(x, r, world, engine, prev_triggers=prev, pre_roll=pre_roll)
        prev = p["all_trig_abs"]
        parts.append(p)
    frames = np.concatenate([p["frames"] for p in parts])
    w = max(p["psdu"].shape[1] for p in parts)
    psdu = np.concatenate([np.pad(p["psdu"], ((0, 0), (0, w - p["psdu"].shape[1]))) for p in parts])
    return dict(frames=frames, psdu=psdu, parts=parts)


def gather_recording(x, engine, stride: int = 2048, group=None, device="cpu", pre_roll: int = STREAM_PRE_ROLL):
    """One process per rank (torch.distributed initialised; "nccl" = RCCL on the GPUs, "gloo" in the rehearsals).
    Every rank demodulates its shard; neighbours exchange their accepted triggers for the seam check (a few integers:
    all_gather of a padded int64 list); a rank whose seam does not hold runs again with a longer pre-roll; then ONE
    all-gather of fixed-stride PSDU rows + frame records, padded to the largest per-rank count.  Every rank returns the
    whole recording's frames and PSDUs in stream order (rank order = stream order: ownership ranges are disjoint and
    ascending)."""
    import torch
    import torch.distributed as dist
    from .capi import FRAME_DTYPE
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def exchange_triggers(trig):
        cnt = torch.tensor([trig.size], dtype=torch.int64, device=device)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt, group=group)
        m = max(int(c.item()) for c in cnts)
        buf = torch.full((max(m, 1),), -1, dtype=torch.int64, device=device)
        if trig.size:
            buf[:trig.size] = torch.from_numpy(trig.astype(np.int64)).to(device)
        allb = torch.empty((world * max(m, 1),), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(allb, buf, group=group)
        allb = allb.cpu().numpy().reshape(world, -1)
        return [allb[r][:int(cnts[r].item())] for r in range(world)]

    part = demod_recording_shard(x, rank, world, engine, prev_triggers=None, pre_roll=pre_roll)
    n = int(x.shape[0])
    for _ in range(32):                                # pre-roll doubles: 32 rounds reach the start of any recording
        trigs = exchange_triggers(part["all_trig_abs"])
        ok = rank == 0 or seam_consistent(trigs[rank - 1], part["all_trig_abs"], part["read"][0], part["own"][0])
        flag = torch.tensor([0 if ok else 1], dtype=torch.int64, device=device)
        dist.all_reduce(flag, group=group)
        if int(flag.item()) == 0:
            break
        if not ok:
            part = demod_recording_shard(x, rank, world, engine, prev_triggers=None, pre_roll=part["pre_roll"] * 2)
    else:
        raise RuntimeError("seam check did not settle")
    # the one exchange of the path: fixed-stride PSDU rows + frame records, padded to the largest shard
    cnt = torch.tensor([len(part["frames"])], dtype=torch.int64, device=device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    m = max(max(counts), 1)
    p = np.zeros((m, stride), np.uint8)
    f = np.zeros(m, FRAME_DTYPE)
    w = min(stride, part["psdu"].shape[1]) if len(part["frames"]) else 0
    p[:counts[rank], :w] = part["psdu"][:, :w]
    f[:counts[rank]] = part["frames"]
    pa, fa = all_gather_pdus(torch.from_numpy(p).to(device), torch.from_numpy(f.view(np.uint8).reshape(m, 32)).to(device), group)
    pa = pa.cpu().numpy().reshape(world, m, stride)
    fa = np.ascontiguousarray(fa.cpu().numpy()).view(FRAME_DTYPE).reshape(world, m)
    frames = np.concatenate([fa[r, :counts[r]] for r in range(world)])
    psdu = np.concatenate([pa[r, :counts[r]] for r in range(world)])
    return dict(frames=frames, psdu=psdu, counts=counts, part=part)
