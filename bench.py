#!/usr/bin/env python3
"""bench.py -- headline benchmark of the wifirx receive chain (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the whole RX hot path (autocorrelation + sync_short + sync_long + FFT +
LS equalise + hard demap + LLR, one fused HIP kernel) over one device-resident batch of
synthetic 802.11a frames: BASELINE.json configs[1] = 1,000,000 QPSK-1/2 frames of 294-byte
PSDUs at 20 MHz, AWGN SNR 20 dB, per-frame CFO within +-20 ppm, one frame per 4608-sample slot.

N > 1: one process per GPU.  `python bench.py --gpus N` starts its own ranks (a child
`python -m torch.distributed.run ... bench.py --gpus N`, spawned before this process has touched
the GPU or imported torch; it never re-execs); launched under torch.distributed.run it is a rank.
Every rank owns its own 1M-frame shard (weak scaling); the hot path has no exchange step, so the
timed region is the same at every N.  What follows the hot path -- decode_mac on the device and, for
N > 1, the RCCL all-gather that reassembles the decoded PDU stream on every rank (chunked: the
collective of chunk c runs while decode_mac works on chunk c + 1; buffers allocated once) -- is run
and timed separately after the timed region and reported in "pdu_leg" (never part of `value`).

Rank 0 prints ONE JSON line (see the round prompt for the contract) with the extra objects
"roofline" (dominant kernel: algorithmic bytes / HIP-event kernel time vs the 8 TB/s HBM peak, and the
same on the bytes the PMC counters saw move), "cpu_baseline" (the oracle, timed on this host's cores on
a bounded sample of the same batch), "host_path" (the drop-in block's work() fed with 8192-item host
chunks), "samples_to_pdu" (demod + decode_mac), "ber_vs_tx" (hard decisions against the transmitted interleaved bits)
and "variants": config 2 with CFO = 0, the config-3 and config-1 geometries, the reference's own output set (`carrier`
on, IRS_AP.py:293) and the LMS / COMB / STA equalisers -- each with its kernel time, roofline fraction, channel BER and
the parity of its first 4096 frames with the oracle.

"roofline.box" is the same-run, same-process yardstick of THIS box (tools/box_probe.hip, libboxprobe.so -- measurement
tooling, not product): a float4 stream at the kernel's read : write mix and the kernel's own loads and stores without
arithmetic (combined, loads only, stores only), measured on the very buffers of the timed batch right after the parity
checks.  `frac_of_mem_floor` and `kernel_vs_box_floor` come from these figures only, never from a profiles/ file.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # bytes/s, MI355X_MICROARCH.md (spec)
HBM_ACHIEVABLE = 6.29e12   # bytes/s, MI355X_MICROARCH.md (measured float4 copy)
PSDU_LEN = 294
ENCODING = 2               # QPSK 1/2
SLOT_LEN = 4608
LEAD = 160
SNR_DB = 20.0
BANDWIDTH = 20e6
FREQUENCY = 5.89e9
CFO_MAX = 2 * np.pi * 20e-6 * FREQUENCY / BANDWIDTH    # +-20 ppm of the carrier, rad/sample
N_TEMPLATES = 1024
PSDU_STRIDE = 320


def algorithmic_bytes_per_frame(slot_len, n_sym, n_bpsc):
    """8*S_in + 48*N_sym*(1 + 4*N_BPSC) + 32."""
    return 8 * slot_len + 48 * n_sym * (1 + 4 * n_bpsc) + 32


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames per GPU (config 2: 1M)")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="target wall time of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline, host_path and variant legs")
    ap.add_argument("--decode", action="store_true", help="put decode_mac (and the all-gather) INSIDE every timed step")
    ap.add_argument("--pdu-steps", type=int, default=2, help="steps of the separate decode_mac + all-gather leg (0 = skip)")
    ap.add_argument("--gather-chunks", type=int, default=0,
                    help="frame ranges the PDU all-gather is cut into (N > 1); 0 = as many as keep every range on decode_mac's "
                         "fastest kernel (>= 600 000 frames per range: one range for 1 M frames)")
    ap.add_argument("--host-samples", type=int, default=96_000_000, help="samples pushed through work() in the host_path leg")
    ap.add_argument("--pipeline-batches", type=int, default=6, help="batches of the pipelined demod / decode_mac figure (0 = skip; N = 1 only)")
    ap.add_argument("--no-variants", action="store_true", help="skip the other geometries / output sets / equalisers")
    ap.add_argument("--variant-frames", type=int, default=0, help="frames per variant batch (0 = as --frames)")
    return ap.parse_args(argv)


def launch_ranks(n: int) -> int:
    """Parent of a multi-GPU run: start n ranks under torch.distributed.run and wait.  Nothing here touches the GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def latest_profile(suffix):
    """newest profiles/*<suffix> as a dict (None if absent) -- PMC passes and the memory-floor probe cannot run inside
    a bench run, their committed summaries are quoted instead"""
    try:
        d = os.path.join(ROOT, "profiles")
        cands = sorted(f for f in os.listdir(d) if f.endswith(suffix))
        with open(os.path.join(d, cands[-1])) as f:
            j = json.load(f)
        j["_file"] = "profiles/" + cands[-1]
        return j
    except Exception:
        return None


def device_clocks():
    """rocm-smi's view of the device clocks, read BEFORE this process touches the GPU (a child process; nothing here
    initialises HIP).  None when the tool is missing or refuses."""
    # under rocprofv3 the profiler's preloaded library has initialised the GPU before Python starts, and a child that execs
    # from such a process is refused on this pool: no child processes then
    if os.environ.get("LD_PRELOAD") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return {"skipped": "profiler loaded"}
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, timeout=20, text=True)
        j = json.loads(r.stdout)
        card = j.get("card0") or next(iter(j.values()))
        return {k: v for k, v in card.items() if "clock" in k.lower()}
    except Exception:
        return None


_BOXPROBE = None


def boxprobe_lib():
    """tools/libboxprobe.so (built by __graft_entry__.build()); None when absent.  The library carries the hash of the
    source it was built from: a binary that does not match tools/box_probe.hip is reported as stale and not used."""
    global _BOXPROBE
    if _BOXPROBE is None:
        import ctypes as C
        import hashlib
        path = os.path.join(ROOT, "tools", "libboxprobe.so")
        try:
            lib = C.CDLL(path)
            lib.boxprobe_src_sha.restype = C.c_char_p
            lib.boxprobe_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
            with open(os.path.join(ROOT, "tools", "box_probe.hip"), "rb") as f:
                want = hashlib.sha256(f.read()).hexdigest()[:16]
            have = lib.boxprobe_src_sha().decode()
            _BOXPROBE = (lib, have, have == want)
        except OSError:
            _BOXPROBE = (None, None, False)
    return _BOXPROBE


def box_yardstick(iq_ptr, idx_ptr, llr_ptr, car_ptr, n_frames, slot_len, lead, n_sym, n_bpsc, reps=3):
    """This box, this process, these buffers: the float4 stream and the demod kernel's own memory pattern (see the module
    docstring).  Overwrites the idx / LLR / carrier rows -- call after the parity checks."""
    import ctypes as C
    lib, sha, fresh = boxprobe_lib()
    if lib is None or not fresh:
        return {"available": False, "probe_src_sha": sha, "stale": lib is not None}
    out = (C.c_double * 12)()
    rc = lib.boxprobe_run(iq_ptr, idx_ptr, llr_ptr, car_ptr, n_frames, slot_len, lead, n_sym, n_bpsc, reps, out)
    if rc != 0:
        return {"available": False, "probe_src_sha": sha, "hip_error": rc}
    return {"available": True, "probe_src_sha": sha, "tool": "tools/box_probe.hip",
            "stream_tbps": out[1] / (out[0] * 1e-3) / 1e12, "stream_ms": out[0], "stream_gb": out[1] / 1e9,
            "mem_floor_ms": out[2], "symbols_only_ms": out[3], "loads_only_ms": out[4], "stores_only_ms": out[5],
            "pattern_gb": out[6] / 1e9,
            "plain_loads_stores": {"mem_floor_ms": out[7], "stores_only_ms": out[8]},
            "streaming_decisions_too": {"mem_floor_ms": out[9], "stores_only_ms": out[10]},
            "note": "same run, same process, the timed batch's own buffers: float4 stream at the kernel's read:write mix; "
                    "the demod kernel's global loads and stores without arithmetic (mem_floor = preamble reads + symbol "
                    "loads + whole-line stores; loads / stores alone; the kernel issues streaming (nt) sample loads and 16-byte pieces "
                    "and plain decision dwords: the same patterns all plain and all streaming beside it), best of %d launches each" % reps}


def ber_vs_tx(torch, idx_t, frames_np, tx, n_bpsc, capi):
    """Coded-bit BER of the hard decisions against the transmitted interleaved bits ("vs transmitted
    bits"): slot k carries template k mod n_templates (wr_synth.hip); frames that did not demodulate completely at the
    transmitted rate and length are counted apart (they have no decisions to compare)."""
    n_t, n_sym = tx.data_idx.shape[0], tx.data_idx.shape[1]
    n = (idx_t.shape[0] // n_t) * n_t
    good = ((frames_np["flags"][:n] & capi.F_COMPLETE) != 0) & (frames_np["encoding"][:n] == tx.encoding) & \
           (frames_np["n_sym"][:n] == n_sym)
    ref = torch.from_numpy(np.ascontiguousarray(tx.data_idx.reshape(n_t, n_sym * 48))).to(idx_t.device)
    d = idx_t[:n].view(n // n_t, n_t, n_sym * 48) ^ ref[None]
    err = torch.zeros((n // n_t, n_t), dtype=torch.int32, device=idx_t.device)
    for b in range(n_bpsc):
        err += ((d >> b) & 1).sum(dim=2, dtype=torch.int32)
    g = torch.from_numpy(good.reshape(n // n_t, n_t)).to(idx_t.device)
    bit_errors = int((err * g).sum(dtype=torch.int64).item())
    n_good = int(good.sum())
    bits = n_good * n_sym * 48 * n_bpsc
    return {"coded_ber": bit_errors / bits if bits else None, "bit_errors": bit_errors, "bits_compared": bits,
            "frames_compared": n_good, "frames_not_demodulated": int(n - n_good),
            "what": "hard decisions (wifirx_out.idx) vs the transmitter's interleaved coded bits, before decode_mac"}


class VariantArena:
    """Device buffers of the variants leg, allocated ONCE for the largest geometry and cut into views per variant (the
    first version allocated and freed 60-110 GB per variant: 2 s each); plus the TX templates and synthesised batches the
    variants share (the four config-2 variants run on the same batch)."""

    def __init__(self, torch, n_frames, geometries):
        self.torch, self.n = torch, n_frames
        z = lambda nbytes: torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
        self.iq = z(max(8 * sl for sl, _, _ in geometries) * n_frames)
        self.idx = z(max(48 * ns for _, ns, _ in geometries) * n_frames)
        self.llr = z(max(192 * ns * nb for _, ns, nb in geometries) * n_frames)
        self.car = z(max(384 * ns for _, ns, _ in geometries) * n_frames)
        self.frames = z(32 * n_frames)
        self.tx = {}
        self.synth_key = None

    def views(self, slot_len, n_sym, n_bpsc, want_carrier):
        t, n = self.torch, self.n
        iq = self.iq[:n * slot_len * 8].view(t.float32).view(n, slot_len, 2)
        idx = self.idx[:n * n_sym * 48].view(n, n_sym * 48)
        llr = self.llr[:n * n_sym * 192 * n_bpsc].view(t.float32).view(n, n_sym * 48 * n_bpsc)
        car = self.car[:n * n_sym * 384].view(t.float32).view(n, n_sym * 48, 2) if want_carrier else None
        fr = self.frames.view(n, 32)
        for b in (idx, llr, car, fr):
            if b is not None:
                b.zero_()                   # the kernels only write what a frame fills
        return iq, fr, idx, llr, car


def run_variant(torch, capi, txgen, orc, arena, name, enc, slot_len, n_frames, device, chan_est=0, want_carrier=False,
                taps=None, snr_db=SNR_DB, cfo_max=CFO_MAX, n_templates=256, seed=4321, cores=1, yardstick=True,
                parity_frames=4096, cite="", lead=LEAD):
    """One more geometry / output set / equaliser through the same timed kernel: device-resident batch made like the
    headline's (host templates -> Philox AWGN + CFO on the device), kernel time by HIP events (mean of 3 after a warm-up),
    roofline fraction on SURVEY.md 8(d)'s bytes (carrier rows included when on), channel BER, parity of the first frames
    with the oracle, and -- for a geometry or output set of its own -- this box's memory yardstick."""
    n_sym = txgen.n_sym_for(PSDU_LEN, enc)
    n_bpsc = txgen.RATE_TABLE[enc][0]
    key = (enc, n_templates, seed, taps is not None)
    if key not in arena.tx:
        tx = txgen.encode_psdus(txgen.make_psdus(n_templates, PSDU_LEN, seed=seed), enc)
        samples = tx.samples
        if taps is not None:         # multipath on the host templates (tests/golden/sv_taps.npy: one draw per template)
            samples = txgen.impair(tx.samples, None, cfo=0.0, lead=0, total=tx.samples.shape[1] + taps.shape[1], taps=taps[:n_templates])
        arena.tx[key] = (tx, samples)
    tx, samples = arena.tx[key]
    assert lead + samples.shape[1] <= slot_len
    rx = capi.WifiRx(bandwidth=BANDWIDTH, frequency=FREQUENCY, sensitivity=0.56, chan_est=chan_est, max_sym=n_sym,
                     llr_bits=n_bpsc, want_carrier=want_carrier, device=device)
    iq, frames_t, idx_t, llr_t, car_t = arena.views(slot_len, n_sym, n_bpsc, want_carrier)
    skey = key + (slot_len, snr_db, float(cfo_max), lead)
    if arena.synth_key != skey:      # (the box yardstick only reads the samples: a batch stays valid for the next variant on it)
        rx.synth_slots(samples, iq.data_ptr(), slot_len, n_frames, lead, snr_db, float(cfo_max), seed)
        arena.synth_key = skey
    torch.cuda.synchronize()
    out = capi.Out(frames_t.data_ptr(), idx_t.data_ptr(), llr_t.data_ptr(), car_t.data_ptr() if want_carrier else None,
                   None, 0, 1, None)
    ms = capi.C.c_float(0)

    def demod(iters):
        rx._check(capi.lib().wifirx_time_demod(rx._h, iq.data_ptr(), slot_len, n_frames, capi.C.byref(out), iters, capi.C.byref(ms)))
        return ms.value

    demod(1)
    runs = [demod(1) for _ in range(3)]
    kernel_ms = float(np.mean(runs))
    fr = frames_t.cpu().numpy().view(capi.FRAME_DTYPE).reshape(-1)
    bpf = algorithmic_bytes_per_frame(slot_len, n_sym, n_bpsc) + (384 * n_sym if want_carrier else 0)
    res = {"what": name, "cite": cite, "frames": n_frames, "slot_len": slot_len, "encoding": enc, "n_sym": n_sym,
           "chan_est": ["LS", "LMS", "COMB", "STA"][chan_est], "carrier": bool(want_carrier), "snr_db": snr_db,
           "kernel_ms": kernel_ms, "kernel_ms_min": float(min(runs)),
           "gsamples_per_s": float(n_frames) * slot_len / (kernel_ms * 1e-3) / 1e9,
           "algorithmic_bytes_per_frame": bpf, "frac": bpf * n_frames / (kernel_ms * 1e-3) / HBM_PEAK,
           "frames_complete": int(((fr["flags"] & capi.F_COMPLETE) != 0).sum())}
    sync = (fr["flags"] & capi.F_SYNC) != 0
    first_lts = (fr["trigger"].astype(np.int64) - 16 + fr["frame_start"])[sync]       # slot sample the long training symbols start at
    res["frames_starting_on_a_line"] = float((first_lts % 16 == 0).mean()) if sync.any() else None
    res["ber_vs_tx"] = ber_vs_tx(torch, idx_t, fr, tx, n_bpsc, capi)
    if orc is not None and parity_frames > 0:
        n_p = min(parity_frames, n_frames)
        prm = orc.make_params(bandwidth=BANDWIDTH, frequency=FREQUENCY, max_sym=n_sym, llr_bits=n_bpsc, chan_est=chan_est)
        x = iq[:n_p].cpu().numpy().view(np.complex64).reshape(-1)
        o = orc.demod_batch(x, slot_len, prm, want_eq=want_carrier, n_threads=cores)
        mism = int((idx_t[:n_p].cpu().numpy().reshape(n_p, n_sym, 48) != o["idx"]).sum()) + \
               int((llr_t[:n_p].cpu().numpy() != o["llr"]).sum()) + int((fr[:n_p] != o["frames"]).sum())
        if want_carrier:
            mism += int((car_t[:n_p].cpu().numpy().view(np.complex64).reshape(n_p, n_sym, 48) != o["eq"]).sum())
        res["parity"] = {"frames_checked": n_p, "mismatching_values": mism}
    if yardstick:
        box = box_yardstick(iq.data_ptr(), idx_t.data_ptr(), llr_t.data_ptr(), car_t.data_ptr() if want_carrier else None,
                            n_frames, slot_len, lead, n_sym, n_bpsc, reps=2)
        res["box"] = box
        if box.get("available"):
            res["kernel_vs_box_floor"] = kernel_ms / box["mem_floor_ms"]
    rx.close()
    return res


def collective_info(dist, torch, backend, rank, local_rank, world, coll_dev):
    """What a reader of the JSON line needs to check that the collective library really saw N ranks: the backend as
    torch.distributed reports it, the RCCL version (torch.cuda.nccl.version(): RCCL on ROCm), the rank count as a
    collective itself counts it (all-reduce of ones), and which host / device every rank ran on."""
    ones = torch.ones(1, dtype=torch.int64, device=coll_dev)
    dist.all_reduce(ones)
    me = {"rank": rank, "host": socket.gethostname(), "local_rank": local_rank, "pid": os.getpid()}
    if coll_dev == "cuda":
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            me["device"] = "%s [%s]" % (pr.name, getattr(pr, "pci_bus_id", "?"))
        except Exception:
            pass
    hosts = [None] * world
    dist.all_gather_object(hosts, me)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:
            ver = "unavailable: %s" % e
    return {"backend": dist.get_backend(), "rccl_version": ver, "world_size_seen": int(ones.item()),
            "dist_world_size": dist.get_world_size(), "ranks_hosts": hosts}


def main():
    t_main = time.perf_counter()
    args = parse_args()
    stub = os.environ.get("WIFIRX_BENCH_STUB") == "1"        # CPU rehearsal of the launcher + all-gather plumbing (tests/)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    # stdout carries ONE JSON line and nothing else: whatever the libraries print there while this runs (RCCL's version banner,
    # gloo's connection notes) is sent to stderr, at the file-descriptor level; the descriptor comes back for the JSON line
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    clocks = device_clocks() if rank == 0 and not stub else None       # before anything here touches the GPU
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    import torch
    # "nccl" is RCCL on ROCm.  WIFIRX_BENCH_BACKEND=gloo is for rehearsing the multi-rank code path on a box with
    # fewer GPUs than ranks (the collectives then go through host memory; never used for a reported number).
    backend = "gloo" if stub else os.environ.get("WIFIRX_BENCH_BACKEND", "nccl")
    if not stub:
        n_dev = torch.cuda.device_count()
        if n_dev == 0 or not torch.cuda.is_available():
            print("bench.py: no GPU visible; the product has no CPU fallback", file=sys.stderr)
            sys.exit(2)
        if world > n_dev and backend == "nccl":
            print("bench.py: %d ranks but %d GPUs (WIFIRX_BENCH_BACKEND=gloo rehearses on fewer)" % (world, n_dev), file=sys.stderr)
            sys.exit(2)
        local_rank %= n_dev
        torch.cuda.set_device(local_rank)
    dist = None
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    # WIFIRX_BENCH_FORCE_DIST=1: run the multi-rank code path with ONE rank (process group, barriers, RCCL all-reduce /
    # all-gather on the library's stream) -- the only way RCCL itself can execute on the one-GPU build pool
    use_dist = world > 1 or (os.environ.get("WIFIRX_BENCH_FORCE_DIST") == "1" and not stub)
    if use_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if use_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from wifirx import dist as wdist

    n_frames = args.frames
    do_decode = args.decode                 # decode inside the timed step (off by default)
    want_pdus = do_decode or args.pdu_steps > 0

    if stub:
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        run_stub(args, rank, world, dist, wdist, torch)
        return

    from wifirx import capi, txgen
    n_sym = txgen.n_sym_for(PSDU_LEN, ENCODING)
    n_bpsc = txgen.RATE_TABLE[ENCODING][0]

    # ---- synthetic input: host templates -> device slots (Philox AWGN + CFO on the GPU) ----
    psdu = txgen.make_psdus(N_TEMPLATES, PSDU_LEN, seed=rank_seeds(rank)["payload_seed"])
    tx = txgen.encode_psdus(psdu, ENCODING)
    frame_len = tx.samples.shape[1]
    assert LEAD + frame_len <= SLOT_LEN

    rx = capi.WifiRx(bandwidth=BANDWIDTH, frequency=FREQUENCY, sensitivity=0.56, max_sym=n_sym,
                     llr_bits=n_bpsc, want_carrier=False, device=local_rank)
    iq = torch.empty((n_frames, SLOT_LEN, 2), dtype=torch.float32, device="cuda")
    cfo = torch.empty(n_frames, dtype=torch.float32, device="cuda")
    synth_seed = rank_seeds(rank)["synth_seed"]
    rx.synth_slots(tx.samples, iq.data_ptr(), SLOT_LEN, n_frames, LEAD, SNR_DB, float(CFO_MAX), synth_seed, cfo.data_ptr())
    # outputs as torch tensors (device memory + RCCL plumbing only)
    frames_t = torch.zeros((n_frames, 32), dtype=torch.uint8, device="cuda")
    idx_t = torch.zeros((n_frames, n_sym * 48), dtype=torch.uint8, device="cuda")
    llr_t = torch.zeros((n_frames, n_sym * 48 * n_bpsc), dtype=torch.float32, device="cuda")
    psdu_t = torch.zeros((n_frames, PSDU_STRIDE), dtype=torch.uint8, device="cuda") if want_pdus else None
    # the decisions as bit planes (wifirx_out.hbits): what decode_mac reads; written by the demod kernel of the PDU leg
    hbits_t = torch.zeros((n_frames, n_sym * 12), dtype=torch.int32, device="cuda") if want_pdus else None
    out = capi.Out(frames_t.data_ptr(), idx_t.data_ptr(), llr_t.data_ptr(), None,
                   psdu_t.data_ptr() if want_pdus else None, PSDU_STRIDE if want_pdus else 0, 1, None)
    out_hb = capi.Out(frames_t.data_ptr(), idx_t.data_ptr(), llr_t.data_ptr(), None,
                      psdu_t.data_ptr(), PSDU_STRIDE, 1, None, None, hbits_t.data_ptr()) if want_pdus else None
    if os.environ.get("WIFIRX_BENCH_PLANES") and want_pdus:      # experiment switch: time the demod kernel that also writes the planes
        out = out_hb
    if os.environ.get("WIFIRX_BENCH_NOIDX"):                     # experiment switch: what the four byte stores per symbol cost
        out.idx = None
    gather = None
    if use_dist and want_pdus:
        # decode_mac's four-frames-per-lane kernel takes ranges of >= 600 000 frames (12.0 ms per 1 M frames; two ranges of
        # 500 000 on the two-frames-per-lane kernel: 14.1 ms): overlapping the exchange with the next range's decode pays only
        # when the ranges stay that long
        n_chunks = args.gather_chunks if args.gather_chunks > 0 else max(1, n_frames // 600000)
        gather = wdist.ChunkedPduGather(n_frames, PSDU_STRIDE, n_chunks, coll_dev)
    lib_stream = torch.cuda.ExternalStream(rx.stream_ptr())      # the handle's own HIP stream, as torch sees it
    # the zero fills above ran on torch's stream, the library launches on its own: order them once
    torch.cuda.synchronize()

    def demod(o):
        ms = capi.C.c_float(0)
        rx._check(capi.lib().wifirx_time_demod(rx._h, iq.data_ptr(), SLOT_LEN, n_frames, capi.C.byref(o), 1, capi.C.byref(ms)))
        return ms.value

    def step():
        """one pass of the hot path; returns the demod kernel's HIP-event time in ms"""
        ms = demod(out_hb if do_decode else out)
        if do_decode:
            pdu_step()
        return ms

    def decode_range(lo, hi, planes=True):
        o = capi.Out(frames_t.data_ptr() + lo * 32, idx_t.data_ptr() + lo * n_sym * 48, None, None,
                     psdu_t.data_ptr() + lo * PSDU_STRIDE, PSDU_STRIDE, 1, None, None,
                     hbits_t.data_ptr() + lo * n_sym * 48 if planes else None)
        rx._check(capi.lib().wifirx_decode_batch(rx._h, hi - lo, capi.C.byref(o)))

    def pdu_step():
        """decode_mac over the demodulated batch; for N > 1 chunk by chunk, every chunk's all-gather (PSDUs + frame
        records, RCCL's own stream) overlapping the next chunk's decode.  Returns (decode ms, exposed all-gather ms)."""
        t_a = time.perf_counter()
        if gather is None:
            decode_range(0, n_frames)
            rx.sync()
            return (time.perf_counter() - t_a) * 1e3, 0.0
        for c in range(gather.n_chunks):
            lo, hi = gather.chunk_range(c)
            decode_range(lo, hi)
            if coll_dev == "cuda":
                with torch.cuda.stream(lib_stream):      # the collective waits for this chunk's decode only
                    gather.gather_chunk(c, psdu_t, frames_t, async_op=True)
            else:                                        # gloo rehearsal: through host memory, no overlap
                rx.sync()
                gather.gather_chunk(c, psdu_t.cpu(), frames_t.cpu(), async_op=False)
        rx.sync()
        t_b = time.perf_counter()
        gather.wait()
        torch.cuda.synchronize()
        return (t_b - t_a) * 1e3, (time.perf_counter() - t_b) * 1e3

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(args.steps):
        kernel_ms += step()
    t_own = time.perf_counter() - t0          # this rank alone, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_samples = float(n_frames) * SLOT_LEN * world * args.steps
    value = total_samples / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    kernel_ms_avg = kernel_ms / args.steps
    per_rank = [dict({"rank": rank, "kernel_ms": kernel_ms_avg, "gsamples_per_s": float(n_frames) * SLOT_LEN * args.steps / t_own / 1e9},
                     **rank_seeds(rank))]
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered

    coll = collective_info(dist, torch, backend, rank, local_rank, world, coll_dev) if use_dist else None

    pipe_err = {}
    # ---- the leg behind the hot path: decode_mac (+ RCCL all-gather of the PDUs), timed on its own ----
    pdu_leg = None
    if args.pdu_steps > 0:
        # the demod kernel once more, now also writing the bit planes decode_mac reads (same frames, same decisions)
        demod(out_hb)
        demod_planes_ms = float(np.median([demod(out_hb) for _ in range(3)]))
        # what the reference's application needs -- PDUs only: the demod kernel writing nothing but the planes and the records
        out_pdu = capi.Out(frames_t.data_ptr(), None, None, None, psdu_t.data_ptr(), PSDU_STRIDE, 1, None, None, hbits_t.data_ptr())
        demod(out_pdu)
        demod_pdu_only_ms = float(np.median([demod(out_pdu) for _ in range(3)]))
        demod(out_hb)                             # the buffers as the timed configuration leaves them
        pdu_step()                                # untimed: first-call allocations (survivor scratch), communicator set-up
        barrier()
        dec_ms, ag_ms = [], []
        t1 = time.perf_counter()
        for _ in range(args.pdu_steps):
            d, a = pdu_step()
            dec_ms.append(d)
            ag_ms.append(a)
        barrier()
        leg = (time.perf_counter() - t1) / args.pdu_steps * 1e3
        ag_alone = None
        if gather is not None:          # the same exchange with nothing to hide behind
            barrier()
            t2 = time.perf_counter()
            for c in range(gather.n_chunks):
                if coll_dev == "cuda":
                    gather.gather_chunk(c, psdu_t, frames_t, async_op=True)
                else:
                    gather.gather_chunk(c, psdu_t.cpu(), frames_t.cpu(), async_op=False)
            gather.wait()
            barrier()
            ag_alone = (time.perf_counter() - t2) * 1e3
        # the same decode from `idx` alone (a caller without planes: a pre-pass packs them), once
        decode_range(0, n_frames, planes=False)
        rx.sync()
        t3 = time.perf_counter()
        decode_range(0, n_frames, planes=False)
        rx.sync()
        dec_from_idx_ms = (time.perf_counter() - t3) * 1e3
        # ---- the same two kernels pipelined: decode_mac of batch k on a second handle's stream while the demod of batch k + 1 runs
        #      on the first one's; two sets of records / planes / PSDUs; BOTH timed together, wall clock over K batches ----
        pipe = None
        if not use_dist and args.pipeline_batches > 0:
            try:
                import threading
                K = args.pipeline_batches
                rx2 = capi.WifiRx(bandwidth=BANDWIDTH, frequency=FREQUENCY, sensitivity=0.56, max_sym=n_sym, llr_bits=n_bpsc,
                                  want_carrier=False, device=local_rank)
                frames2_t = torch.zeros_like(frames_t)
                hbits2_t = torch.zeros_like(hbits_t)
                psdu2_t = torch.zeros_like(psdu_t)
                torch.cuda.synchronize()
                sets = [(frames_t, hbits_t, psdu_t), (frames2_t, hbits2_t, psdu2_t)]
                o_dem = [capi.Out(f.data_ptr(), idx_t.data_ptr(), llr_t.data_ptr(), None, p_.data_ptr(), PSDU_STRIDE, 1, None, None, hb.data_ptr())
                         for f, hb, p_ in sets]
                o_dec = [capi.Out(f.data_ptr(), None, None, None, p_.data_ptr(), PSDU_STRIDE, 1, None, None, hb.data_ptr()) for f, hb, p_ in sets]
                filled, free_ = threading.Semaphore(0), threading.Semaphore(2)
                dem_ms, err = [], []

                def producer(n):
                    try:
                        for k in range(n):
                            free_.acquire()
                            dem_ms.append(demod(o_dem[k % 2]))          # returns when the kernel has finished (HIP events)
                            filled.release()
                    except Exception as e:          # pragma: no cover
                        err.append(e)
                        filled.release()

                def consumer(n):
                    try:
                        for k in range(n):
                            filled.acquire()
                            rx2._check(capi.lib().wifirx_decode_batch(rx2._h, n_frames, capi.C.byref(o_dec[k % 2])))
                            rx2.sync()
                            free_.release()
                    except Exception as e:          # pragma: no cover
                        err.append(e)
                        free_.release()

                def run_pipe(n):
                    ta, tb = threading.Thread(target=producer, args=(n,)), threading.Thread(target=consumer, args=(n,))
                    t_ = time.perf_counter()
                    ta.start(); tb.start(); ta.join(); tb.join()
                    return time.perf_counter() - t_

                run_pipe(2)                           # first-call allocations of the second handle (its survivor scratch)
                del dem_ms[:]
                wall = run_pipe(K)
                fr2 = frames2_t.cpu().numpy().view(capi.FRAME_DTYPE).reshape(-1)
                pipe = {"batches": K, "ms_per_batch": wall / K * 1e3, "gsamples_per_s": float(n_frames) * SLOT_LEN * K / wall / 1e9,
                        "demod_kernel_ms_under_decode": float(np.mean(dem_ms[1:])) if len(dem_ms) > 1 else None,
                        "frames_crc_ok_second_set": int(((fr2["flags"] & capi.F_CRC_OK) != 0).sum()), "errors": [str(e) for e in err],
                        "note": "two handles = two HIP streams: demod (idx + LLRs + planes) of batch k + 1 on one, decode_mac of batch k on "
                                "the other, two sets of records / planes / PSDUs, host threads hand the sets over; wall clock over all "
                                "batches, both kernels timed together"}
                rx2.close()
                del frames2_t, hbits2_t, psdu2_t
                demod(out_hb)                         # the first set as the serial configuration leaves it
                decode_range(0, n_frames)
                rx.sync()
            except Exception as e:        # the pipelined figure is optional: a failure is recorded, the line survives
                pipe = None
                pipe_err["pipelined"] = "%s: %s" % (type(e).__name__, e)
        pdu_leg = {"decode_mac_ms": float(np.median(dec_ms)), "pipelined": pipe,
                   "demod_with_planes_ms": demod_planes_ms, "decode_mac_from_idx_ms": dec_from_idx_ms,
                   "demod_planes_only_ms": demod_pdu_only_ms,
                   "all_gather_ms": ag_alone, "all_gather_exposed_ms": float(np.median(ag_ms)) if use_dist else None,
                   "gather_chunks": gather.n_chunks if gather is not None else None,
                   "ms_per_step": leg, "psdu_stride": PSDU_STRIDE,
                   "gathered_bytes_per_rank": (PSDU_STRIDE + 32) * n_frames * world if use_dist else None,
                   "note": "decode_mac on the device; N > 1: RCCL all_gather_into_tensor of PSDUs and frame records, chunk c "
                           "overlapping decode_mac of chunk c+1 (all_gather_exposed_ms = what is left after the last decode); "
                           "not in `value`"}

    # ---- sanity of the timed work: every frame must have been demodulated completely ----
    fr = frames_t.cpu().numpy().view(capi.FRAME_DTYPE).reshape(-1)
    n_complete = int(((fr["flags"] & capi.F_COMPLETE) != 0).sum())
    n_crc = int(((fr["flags"] & capi.F_CRC_OK) != 0).sum()) if want_pdus else None
    gather_ok = None
    if gather is not None:
        # every rank must now hold every rank's records: compare my own rows and count the good FCS of all ranks
        fa = gather.frames_all.cpu().numpy().view(capi.FRAME_DTYPE).reshape(-1)
        mine = gather.frame_order()[rank]
        gather_ok = bool(np.array_equal(fa[mine], fr))
        n_crc_all = int(((fa["flags"] & capi.F_CRC_OK) != 0).sum())
        pdu_leg["frames_crc_ok_all_ranks"] = n_crc_all
        pdu_leg["gather_consistent"] = gather_ok

    result = None
    leg_errors = dict(pipe_err)       # optional legs that failed: {leg: "Exception: text"} (the line still carries everything else)
    if rank == 0:
        bpf = algorithmic_bytes_per_frame(SLOT_LEN, n_sym, n_bpsc)
        # HBM bytes per launch from the PMC passes of the same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate passes, FETCH_SIZE x2 per MI355X_MICROARCH.md and tools/calib_fetch.hip): collected by
        # tools/pmc.sh, committed as profiles/*_traffic.json -- a bench run cannot count PMCs itself
        # Both files carry the hash of the kernel sources they were measured on (tools/csrc_sha.py): a figure from other
        # kernels than the ones that just ran is reported as stale and left out of the derived fractions.
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from csrc_sha import csrc_sha
        sha_now = csrc_sha()
        tj = latest_profile("_traffic.json")
        traffic_stale = bool(tj) and tj.get("csrc_sha") != sha_now
        moved_per_frame = tj["hbm_bytes_per_frame"] if tj else None
        traffic = moved_per_frame * n_frames / 1e9 if tj else None
        achieved = bpf * n_frames / (kernel_ms_avg * 1e-3)
        moved = moved_per_frame * n_frames / (kernel_ms_avg * 1e-3) if tj and not traffic_stale else None
        result = {
            "metric": "OFDM demod throughput (complex samples/s), 802.11a RX chain sync->LLR",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[1]: %d frames/GPU, 20 MHz QPSK-1/2, PSDU %d B (N_sym %d), "
                            "slot %d samples, AWGN SNR %g dB, CFO +-20 ppm, device-resident; %d TX templates (distinct "
                            "payloads and scrambler seeds) x distinct noise and CFO per slot"
                            % (n_frames, PSDU_LEN, n_sym, SLOT_LEN, SNR_DB, N_TEMPLATES),
                "frames_per_gpu": n_frames, "slot_len": SLOT_LEN, "encoding": "QPSK_1_2", "tx_templates": N_TEMPLATES,
                "outputs": "48 u8 decisions + 96 f32 LLRs per data symbol, 32 B frame record"
                           + ("; decode_mac + PSDU" if do_decode else ""),
                "parallelism": ("frames sharded %d-way, no collective on the hot path" % world if world > 1 else "1 GPU")
                               + ("" if backend == "nccl" else " [REHEARSAL: %s backend]" % backend),
            },
            "gsamples_per_s": value / 1e9,
            "msymbols_per_s": float(n_frames) * (n_sym + 3) * world * args.steps / elapsed / 1e6,
            "frames_complete": n_complete,
            "frames_crc_ok": n_crc,
            "per_rank": per_rank,
            "collective": coll,
            "n1_equivalent_gsamples_per_s": per_rank[0]["gsamples_per_s"],
            "pdu_leg": pdu_leg,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                "traffic": traffic,
                "traffic_unit": "GB per launch (PMC, %s)" % (tj["_file"] if tj else "none"),
                "traffic_stale": traffic_stale,       # true: the PMC file was measured on other kernel sources than this tree's
                "csrc_sha": sha_now,
                "hw_floor_ms": (moved_per_frame * n_frames / HBM_ACHIEVABLE * 1e3) if tj and not traffic_stale else None,
                "kernel": "wr::demod_batch_kernel",
                "kernel_ms": kernel_ms_avg,
                "algorithmic_bytes_per_frame": bpf,
                "bytes_moved_per_frame": moved_per_frame,
                "bytes_moved_frac": moved / HBM_PEAK if moved else None,
                "hbm_achievable": HBM_ACHIEVABLE / 1e9,
                "frac_of_achievable": achieved / HBM_ACHIEVABLE,
                "bytes_moved_frac_of_achievable": moved / HBM_ACHIEVABLE if moved else None,
                "box": None,                     # filled below: this box's own yardstick, same run, same process
                "device_clocks_before_run": clocks,
            },
        }
        if pdu_leg is not None:
            t_pair = pdu_leg["demod_with_planes_ms"] + pdu_leg["decode_mac_ms"]
            result["samples_to_pdu"] = {"demod_ms": pdu_leg["demod_with_planes_ms"], "decode_mac_ms": pdu_leg["decode_mac_ms"],
                                        "gsamples_per_s": float(n_frames) * SLOT_LEN / (t_pair * 1e-3) / 1e9,
                                        "pipelined_gsamples_per_s": pdu_leg["pipelined"]["gsamples_per_s"] if pdu_leg.get("pipelined") else None,
                                        "pipelined_ms_per_batch": pdu_leg["pipelined"]["ms_per_batch"] if pdu_leg.get("pipelined") else None,
                                        "pdu_only_demod_ms": pdu_leg["demod_planes_only_ms"],
                                        "pdu_only_gsamples_per_s": float(n_frames) * SLOT_LEN /
                                                                   ((pdu_leg["demod_planes_only_ms"] + pdu_leg["decode_mac_ms"]) * 1e-3) / 1e9,
                                        "note": "per GPU: samples -> decoded PSDUs, device-resident: the demod kernel writing idx, LLRs "
                                                "and the bit planes of the decisions (wifirx_out.hbits), decode_mac reading the planes; "
                                                "pdu_only = the demod kernel writing the planes alone (no idx, no LLRs): what the "
                                                "reference's image transfer consumes"}

    # ---- cpu_baseline leg: the oracle on this host's cores, bounded sample of the same batch ----
    if rank == 0 and world == 1 and not args.no_cpu:      # contract: rank 0 at N=1 only
        try:
            from oracle import oracle as orc
            march = orc.use_native_build()                    # -march=native copy built on this host when gcc is there
            cores = os.cpu_count() or 1
            prm = orc.make_params(bandwidth=BANDWIDTH, frequency=FREQUENCY, max_sym=n_sym, llr_bits=n_bpsc)
            probe = min(n_frames, 32 * cores)
            x = iq[:probe].cpu().numpy().view(np.complex64).reshape(-1)
            t = time.perf_counter()
            orc.demod_batch(x, SLOT_LEN, prm, n_threads=cores)
            rate = probe / (time.perf_counter() - t)
            n_cpu = int(max(probe, min(n_frames, rate * args.cpu_seconds, 262144)))
            x = iq[:n_cpu].cpu().numpy().view(np.complex64).reshape(-1)
            t = time.perf_counter()
            o = orc.demod_batch(x, SLOT_LEN, prm, n_threads=cores)
            dt = time.perf_counter() - t
            # parity of the timed GPU outputs with the oracle on the same frames (bit for bit)
            g_idx = idx_t[:n_cpu].cpu().numpy().reshape(n_cpu, n_sym, 48)
            g_llr = llr_t[:n_cpu].cpu().numpy()
            g_fr = fr[:n_cpu].copy()
            g_fr["flags"] &= ~np.uint32(capi.F_DECODED | capi.F_CRC_OK)
            mism = int((g_idx != o["idx"]).sum()) + int((g_llr != o["llr"]).sum()) + int((g_fr != o["frames"]).sum())
            if want_pdus:       # decoded PSDUs of the same frames against the oracle's decode_mac
                n_dec = min(n_cpu, 8192)
                of = o["frames"][:n_dec].copy()
                opsdu = orc.decode_batch(of, o["idx"][:n_dec], prm, psdu_stride=PSDU_STRIDE, n_threads=cores)
                g_psdu = psdu_t[:n_dec].cpu().numpy()
                mism += int((g_psdu[:, :PSDU_LEN] != opsdu[:, :PSDU_LEN]).sum()) + int((of["flags"] != fr[:n_dec]["flags"]).sum())
            n1 = max(64, min(n_cpu, int(rate / cores * 1.5)))          # ~1.5 s on one thread
            t = time.perf_counter()
            orc.demod_batch(x[:n1 * SLOT_LEN], SLOT_LEN, prm, n_threads=1)
            dt1 = time.perf_counter() - t
            result["cpu_baseline"] = {
                "value": n_cpu * SLOT_LEN / dt,
                "value_1thread": n1 * SLOT_LEN / dt1,
                "unit": "samples/s",
                "cores": cores,
                "cpu_model": cpu_model(),
                "build": "gcc -O3 -march=%s -ffp-contract=off (the numerics spec forbids contraction)" % march,
                "kind": "port",
                "sample": "first %d frames of the GPU batch (%.1f s), oracle spec mode, OpenMP over frames" % (n_cpu, dt),
                "gpu_vs_cpu": value / (n_cpu * SLOT_LEN / dt),
            }
            result["parity"] = {"frames_checked": n_cpu, "mismatching_values": mism,
                                "oracle": "parity unpinned: the reference holds no vectors for this path (DESIGN.md section 2)"}

            # ---- host_path leg: the drop-in block fed the way a GNU Radio scheduler feeds it ----
            if args.host_samples > 0:
                from wifirx import block, grshim
                n_host = max(1, min(n_frames, args.host_samples // SLOT_LEN))
                xs = iq[:n_host].cpu().numpy().view(np.complex64).reshape(-1)
                blk = block.wifi_phy_rx(bandwidth=BANDWIDTH, frequency=FREQUENCY, max_sym=n_sym, publish_carrier=False,
                                        device=local_rank, batch_samples=1 << 22)
                got = []
                grshim.msg_connect(blk, "mac_out", grshim.sink_block(got.append), "in")
                grshim.run_stream(blk, xs[:SLOT_LEN * min(n_host, 256)], chunk=8192)     # warm-up (allocations); stop() settles its frames
                n0 = len(got)
                t = time.perf_counter()
                grshim.run_stream(blk, xs, chunk=8192)
                dth = time.perf_counter() - t
                n_pdu = len(got) - n0
                # the same calls with PDU publication switched off: what work() itself sustains (staging copy, PCIe, device
                # pipeline on the library's worker thread); the difference is Python building one PDU per frame
                blk._publish = lambda: None
                t = time.perf_counter()
                grshim.run_stream(blk, xs, chunk=8192, finish=False)
                dtw = time.perf_counter() - t
                blk.close()
                # the reference wires the `carrier` port too (IRS_AP.py:293,312-313: frame_equalizer.symbols -> the SNR
                # probe): one Python PDU per data SYMBOL, 50 per frame here -- on an eighth of the samples
                blk = block.wifi_phy_rx(bandwidth=BANDWIDTH, frequency=FREQUENCY, max_sym=n_sym, publish_carrier=True,
                                        device=local_rank, batch_samples=1 << 22)
                got_c, n_car = [], [0]
                grshim.msg_connect(blk, "mac_out", grshim.sink_block(got_c.append), "in")
                grshim.msg_connect(blk, "carrier", grshim.sink_block(lambda m: n_car.__setitem__(0, n_car[0] + 1)), "in")
                xc = xs[:SLOT_LEN * max(256, n_host // 8)]
                grshim.run_stream(blk, xc[:SLOT_LEN * 64], chunk=8192)
                n_car[0] = 0
                t = time.perf_counter()
                grshim.run_stream(blk, xc, chunk=8192)
                dtc = time.perf_counter() - t
                result["host_path"] = {"gsamples_per_s": xs.size / dth / 1e9, "work_chunk_items": 8192, "samples": int(xs.size),
                                       "pdus": n_pdu, "frames_in": n_host, "batch_samples": 1 << 22,
                                       "work_only_gsamples_per_s": xs.size / dtw / 1e9,
                                       "with_carrier_port_gsamples_per_s": xc.size / dtc / 1e9,
                                       "with_carrier_port": {"samples": int(xc.size), "carrier_pdus": n_car[0]},
                                       "note": "wifi_phy_rx.work() with pageable host chunks: staging copy, PCIe, detection, frame "
                                               "kernel, decode_mac, one Python PDU per frame (a frame every 4608 samples); "
                                               "work_only = the same calls without building PDUs; with_carrier_port = also one "
                                               "Python PDU per data symbol on `carrier`, as IRS_AP.py:293 wires it; never `value`"}
                blk.close()
        except Exception as e:          # a failing optional leg is recorded in the line, it does not lose the measurement
            leg_errors['cpu_baseline_parity_host_path'] = "%s: %s" % (type(e).__name__, e)

    # ---- channel BER of the timed batch and this box's own memory yardstick, on the timed buffers ----
    if rank == 0:
        try:
            result["ber_vs_tx"] = ber_vs_tx(torch, idx_t, fr, tx, n_bpsc, capi)
            rx.sync()
            torch.cuda.synchronize()
            box = box_yardstick(iq.data_ptr(), idx_t.data_ptr(), llr_t.data_ptr(), None, n_frames, SLOT_LEN, LEAD, n_sym, n_bpsc)
            rf = result["roofline"]
            rf["box"] = box
            if box.get("available"):
                rf["mem_floor_ms"] = box["mem_floor_ms"]
                rf["mem_floor_source"] = "roofline.box (same run)"
                rf["frac_of_mem_floor"] = box["mem_floor_ms"] / kernel_ms_avg
                rf["kernel_vs_box_floor"] = kernel_ms_avg / box["mem_floor_ms"]
                rf["kernel_vs_stream"] = kernel_ms_avg / (box["pattern_gb"] / box["stream_tbps"])      # ms the float4 stream needs for the pattern's bytes
        except Exception as e:          # a failing optional leg is recorded in the line, it does not lose the measurement
            leg_errors['ber_and_box'] = "%s: %s" % (type(e).__name__, e)

    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            # ---- config 2, CFO = 0 variant: same frames and noise law, no carrier offset ----
            rx.synth_slots(tx.samples, iq.data_ptr(), SLOT_LEN, n_frames, LEAD, SNR_DB, 0.0, synth_seed, cfo.data_ptr())
            step()
            k0 = max(2, min(args.steps, 5))
            ms0 = sum(step() for _ in range(k0)) / k0
            fr0 = frames_t.cpu().numpy().view(capi.FRAME_DTYPE).reshape(-1)
            result["variants"] = {"cfo0": {"kernel_ms": ms0, "gsamples_per_s": float(n_frames) * SLOT_LEN / (ms0 * 1e-3) / 1e9,
                                           "frac": bpf * n_frames / (ms0 * 1e-3) / HBM_PEAK,
                                           "frames_complete": int(((fr0["flags"] & capi.F_COMPLETE) != 0).sum()),
                                           "note": "config 2 with CFO = 0 (IRS_tranceiver.py:121 range centre), device kernel time"}}
            t_var = time.perf_counter()
            if not args.no_variants:
                # the other BASELINE.json geometries, the reference's own output set and the other equalisers through the same
                # kernel, each with channel BER and oracle parity of its first 4096 frames; the headline's buffers go first
                del iq, idx_t, llr_t, hbits_t, psdu_t, out, out_hb
                torch.cuda.empty_cache()
                n_var = args.variant_frames or n_frames
                taps = np.load(os.path.join(ROOT, "tests", "golden", "sv_taps.npy"))
                V = result["variants"]
                geo = [(sl, txgen.n_sym_for(PSDU_LEN, e), txgen.RATE_TABLE[e][0]) for sl, e in ((1472, 7), (8576, 0), (SLOT_LEN, ENCODING))]
                arena = VariantArena(torch, n_var, geo)
                kw = dict(n_frames=n_var, device=local_rank, cores=cores)
                V["config3_geometry"] = run_variant(torch, capi, txgen, orc, arena, "BASELINE.json configs[2] geometry: 64-QAM 3/4, slot 1472, "
                                                    "multipath taps tests/golden/sv_taps.npy (utils/SV_channel.py:81-86,128 draws), "
                                                    "LS, 20 dB", 7, 1472, taps=taps, cite="gnu_radio/IRS_AP.py:81,268-285", **kw)
                # ... and the same geometry without the taps.  What the taps cost is the LTS search (eight candidates instead of two when the
                # third largest lag is close to the second: +0.3 ms in the preamble phase alone), not the frames' alignment (under multipath
                # half of them no longer start on a 128-byte line; the config-3 geometry, bound by instruction issue in front of the data
                # symbols, loses 1-2 % to that: profiles/r05_preamble_analysis.txt).  The alignment is what `config2_off_line` prices.
                V["config3_geometry_awgn"] = run_variant(torch, capi, txgen, orc, arena, "config-3 geometry on AWGN 20 dB (no taps): every frame "
                                                         "starts on a 128-byte line", 7, 1472, yardstick=False, parity_frames=0,
                                                         cite="gnu_radio/IRS_AP.py:81,268-285", **kw)
                # config 2 with every frame three samples (24 bytes) off the 128-byte grid: a row's 512 bytes of a symbol straddle five
                # lines, and the line with the cyclic prefix -- skipped when frames start on a line -- is fetched too (+25 % sample traffic).
                # A recording has no alignment at all; BASELINE's synthetic slots (lead 160 samples = ten lines) are the best case.
                V["config2_off_line"] = run_variant(torch, capi, txgen, orc, arena, "config 2 with a lead of %d samples: no frame starts on a "
                                                    "128-byte line" % (LEAD + 3), ENCODING, SLOT_LEN, yardstick=False, parity_frames=1024,
                                                    cite="gnu_radio/IRS_AP.py:268-285", lead=LEAD + 3, **kw)
                V["config2_off_line"]["vs_headline_same_run"] = V["config2_off_line"]["kernel_ms"] / kernel_ms_avg
                V["config1_geometry"] = run_variant(torch, capi, txgen, orc, arena, "BASELINE.json configs[0] geometry: BPSK 1/2, slot 8576, "
                                                    "AWGN 20 dB", 0, 8576, cite="gnu_radio/IRS_AP.py:268-285", **kw)
                V["carrier_on"] = run_variant(torch, capi, txgen, orc, arena, "config 2 with the reference's own output set: equalised points "
                                              "on `carrier` (want_carrier = 1), LS", ENCODING, SLOT_LEN, want_carrier=True,
                                              cite="gnu_radio/IRS_AP.py:293,312-313", **kw)
                for ce, nm in ((1, "lms"), (2, "comb"), (3, "sta")):
                    V[nm] = run_variant(torch, capi, txgen, orc, arena, "config 2 with the %s equaliser" % nm.upper(), ENCODING, SLOT_LEN,
                                        chan_est=ce, yardstick=False, cite="gnu_radio/IRS_AP.py:139-141", **kw)
                    V[nm]["vs_LS_same_run"] = V[nm]["kernel_ms"] / kernel_ms_avg
            result["wall_s"] = {"whole_process": time.perf_counter() - t_main, "variants_leg": time.perf_counter() - t_var}
        except Exception as e:          # a failing optional leg is recorded in the line, it does not lose the measurement
            leg_errors['variants'] = "%s: %s" % (type(e).__name__, e)

    # The one JSON line leaves as soon as it is complete -- BEFORE the barrier, the process group's teardown and the handle's:
    # a hang or an exception there must not cost the measurement.
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    if rank == 0:
        if leg_errors:
            result["leg_errors"] = leg_errors
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    rx.close()


def rank_seeds(rank):
    """Every rank draws its own payloads and its own noise / CFO realisations (weak scaling: 1 M distinct frames per GPU)."""
    return {"payload_seed": 2025 + rank, "synth_seed": 1234 + 7919 * rank}


def run_stub(args, rank, world, dist, wdist, torch):
    """CPU rehearsal (WIFIRX_BENCH_STUB=1, gloo): the launcher, the rank bookkeeping, the barrier / max-over-ranks timing
    and the chunked PDU all-gather with a stand-in for the compute step.  The JSON line says so; nothing here is a
    measurement of the product."""
    from wifirx import capi
    n_frames = min(args.frames, 4096)
    frames = np.zeros(n_frames, capi.FRAME_DTYPE)
    psdu = np.zeros((n_frames, PSDU_STRIDE), np.uint8)

    def step():           # stand-in for demod + decode_mac: every frame "decodes" to a rank- and frame-dependent PSDU
        k = np.arange(n_frames)
        frames["flags"] = capi.F_DETECTED | capi.F_SYNC | capi.F_SIGNAL | capi.F_COMPLETE | capi.F_DECODED | capi.F_CRC_OK
        frames["psdu_len"] = 64
        frames["trigger"] = k
        psdu[:, 0] = rank
        psdu[:, 1:5] = (k[:, None] >> (8 * np.arange(4))) & 255
        return 0.0

    gather = wdist.ChunkedPduGather(n_frames, PSDU_STRIDE, args.gather_chunks if args.gather_chunks > 0 else 2, "cpu") if world > 1 else None
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ok = True
    n_pdus = n_frames
    coll = collective_info(dist, torch, "gloo", rank, int(os.environ.get("LOCAL_RANK", "0")), world, "cpu") if world > 1 else None
    if gather is not None:
        p_t = torch.from_numpy(psdu)
        f_t = torch.from_numpy(frames.view(np.uint8).reshape(n_frames, 32))
        for c in range(gather.n_chunks):
            gather.gather_chunk(c, p_t, f_t, async_op=False)
        pdus = gather.pdus()
        n_pdus = len(pdus)
        for g, b in pdus:                       # global frame g = rank r, local frame j
            r, j = divmod(g, n_frames)
            ok &= b[0] == r and int.from_bytes(b[1:5], "little") == j and len(b) == 60
        ok &= n_pdus == world * n_frames
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok))
        ok = all(flags)
    per_rank = [dict({"rank": rank, "pid": os.getpid()}, **rank_seeds(rank))]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    if rank == 0:
        print(json.dumps({"metric": "STUB (no GPU work): launcher + all-gather rehearsal", "value": 0.0, "unit": "samples/s",
                          "per_rank": per_rank, "gathered_bytes_per_rank": (PSDU_STRIDE + 32) * n_frames * world,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "stub",
                          "config": {"workload": "stub", "parallelism": "gloo, %d ranks" % world},
                          "pdus_gathered": n_pdus, "gather_consistent": bool(ok), "collective": coll,
                          "gather_chunks": gather.n_chunks if gather else None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
