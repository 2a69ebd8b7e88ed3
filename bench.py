#!/usr/bin/env python3
"""bench.py -- headline benchmark of the wifirx receive chain (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the whole RX hot path (autocorrelation + sync_short + sync_long + FFT +
LS equalise + hard demap + LLR, one fused HIP kernel) over one device-resident batch of
synthetic 802.11a frames: BASELINE.json configs[1] = 1,000,000 QPSK-1/2 frames of 294-byte
PSDUs at 20 MHz, AWGN SNR 20 dB, per-frame CFO within +-20 ppm, one frame per 4608-sample slot.
With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) every rank owns its own
1M-frame shard (weak scaling); the hot path has no exchange step, so the timed region is the same
at every N.  What follows the hot path -- decode_mac on the device and, for N > 1, the RCCL
all-gather that reassembles the decoded PDU stream on every rank -- is run and timed separately
after the timed region and reported in the extra object "pdu_leg" (never part of `value`).

Rank 0 prints ONE JSON line (see the round prompt for the contract) with the extra objects
"roofline" (dominant kernel, algorithmic bytes / HIP-event kernel time vs the 8 TB/s HBM peak)
and "cpu_baseline" (the oracle, timed on this host's cores on a bounded sample of the same batch).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # bytes/s, MI355X_MICROARCH.md
PSDU_LEN = 294
ENCODING = 2               # QPSK 1/2
SLOT_LEN = 4608
LEAD = 160
SNR_DB = 20.0
BANDWIDTH = 20e6
FREQUENCY = 5.89e9
CFO_MAX = 2 * np.pi * 20e-6 * FREQUENCY / BANDWIDTH    # +-20 ppm of the carrier, rad/sample
N_TEMPLATES = 1024


def algorithmic_bytes_per_frame(slot_len, n_sym, n_bpsc):
    """8*S_in + 48*N_sym*(1 + 4*N_BPSC) + 32."""
    return 8 * slot_len + 48 * n_sym * (1 + 4 * n_bpsc) + 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames per GPU (config 2: 1M)")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="target wall time of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--decode", action="store_true", help="put decode_mac (and the all-gather) INSIDE every timed step")
    ap.add_argument("--pdu-steps", type=int, default=2, help="steps of the separate decode_mac + all-gather leg (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus), file=sys.stderr)
            sys.exit(2)
    import torch
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the product has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    local_rank %= max(1, torch.cuda.device_count())      # a launcher that shows every rank one device only
    torch.cuda.set_device(local_rank)
    dist = None
    # "nccl" is RCCL on ROCm.  WIFIRX_BENCH_BACKEND=gloo is for rehearsing the multi-rank code path on a box with
    # fewer GPUs than ranks (the collectives then go through host memory; never used for a reported number).
    backend = os.environ.get("WIFIRX_BENCH_BACKEND", "nccl")
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from wifirx import capi, txgen
    from wifirx import dist as wdist

    n_frames = args.frames
    n_sym = txgen.n_sym_for(PSDU_LEN, ENCODING)
    n_bpsc = txgen.RATE_TABLE[ENCODING][0]
    do_decode = args.decode                 # decode inside the timed step (off by default)
    want_pdus = do_decode or args.pdu_steps > 0

    # ---- synthetic input: host templates -> device slots (Philox AWGN + CFO on the GPU) ----
    psdu = txgen.make_psdus(N_TEMPLATES, PSDU_LEN, seed=2025 + rank)
    tx = txgen.encode_psdus(psdu, ENCODING)
    frame_len = tx.samples.shape[1]
    assert LEAD + frame_len <= SLOT_LEN

    rx = capi.WifiRx(bandwidth=BANDWIDTH, frequency=FREQUENCY, sensitivity=0.56, max_sym=n_sym,
                     llr_bits=n_bpsc, want_carrier=False, device=local_rank)
    iq = torch.empty((n_frames, SLOT_LEN, 2), dtype=torch.float32, device="cuda")
    cfo = torch.empty(n_frames, dtype=torch.float32, device="cuda")
    rx.synth_slots(tx.samples, iq.data_ptr(), SLOT_LEN, n_frames, LEAD, SNR_DB, float(CFO_MAX),
                   1234 + 7919 * rank, cfo.data_ptr())
    # outputs as torch tensors (device memory + RCCL plumbing only)
    frames_t = torch.zeros((n_frames, 32), dtype=torch.uint8, device="cuda")
    idx_t = torch.zeros((n_frames, n_sym * 48), dtype=torch.uint8, device="cuda")
    llr_t = torch.zeros((n_frames, n_sym * 48 * n_bpsc), dtype=torch.float32, device="cuda")
    psdu_stride = 320
    psdu_t = torch.zeros((n_frames, psdu_stride), dtype=torch.uint8, device="cuda") if want_pdus else None
    out = capi.Out(frames_t.data_ptr(), idx_t.data_ptr(), llr_t.data_ptr(), None,
                   psdu_t.data_ptr() if want_pdus else None, psdu_stride if want_pdus else 0, 1, None)
    def step():
        """one pass of the hot path; returns the demod kernel's HIP-event time in ms"""
        ms = capi.C.c_float(0)
        rx._check(capi.lib().wifirx_time_demod(rx._h, iq.data_ptr(), SLOT_LEN, n_frames, capi.C.byref(out), 1,
                                               capi.C.byref(ms)))
        if do_decode:
            pdu_step()
        return ms.value

    def pdu_step():
        """decode_mac over the demodulated batch (+ all-gather of PSDUs and frame records for N > 1)"""
        t_a = time.perf_counter()
        rx._check(capi.lib().wifirx_decode_batch(rx._h, n_frames, capi.C.byref(out)))
        rx.sync()
        t_b = time.perf_counter()
        if world > 1:
            wdist.all_gather_pdus(psdu_t.to(coll_dev), frames_t.to(coll_dev))
            torch.cuda.synchronize()
        return (t_b - t_a) * 1e3, (time.perf_counter() - t_b) * 1e3

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(args.steps):
        kernel_ms += step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_samples = float(n_frames) * SLOT_LEN * world * args.steps
    value = total_samples / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    kernel_ms_avg = kernel_ms / args.steps

    # ---- the leg behind the hot path: decode_mac (+ RCCL all-gather of the PDUs), timed on its own ----
    pdu_leg = None
    if args.pdu_steps > 0:
        barrier()
        dec_ms, ag_ms = [], []
        t1 = time.perf_counter()
        for _ in range(args.pdu_steps):
            d, a = pdu_step()
            dec_ms.append(d)
            ag_ms.append(a)
        barrier()
        leg = (time.perf_counter() - t1) / args.pdu_steps * 1e3
        pdu_leg = {"decode_mac_ms": float(np.median(dec_ms)), "all_gather_ms": float(np.median(ag_ms)) if world > 1 else None,
                   "ms_per_step": leg, "psdu_stride": psdu_stride,
                   "note": "decode_mac on the device + RCCL all_gather_into_tensor of PSDUs and frame records; not in `value`"}

    # ---- sanity of the timed work: every frame must have been demodulated completely ----
    fr = frames_t.cpu().numpy().view(capi.FRAME_DTYPE).reshape(-1)
    n_complete = int(((fr["flags"] & capi.F_COMPLETE) != 0).sum())
    n_crc = int(((fr["flags"] & capi.F_CRC_OK) != 0).sum()) if want_pdus else None

    result = None
    if rank == 0:
        bpf = algorithmic_bytes_per_frame(SLOT_LEN, n_sym, n_bpsc)
        # HBM bytes per launch from the PMC passes of the same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate passes, FETCH_SIZE x2 per MI355X_MICROARCH.md and tools/calib_fetch.hip): collected by
        # tools/pmc.sh, committed as profiles/*_traffic.json -- a bench run cannot count PMCs itself
        traffic = None
        try:
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
            with open(os.path.join(ROOT, "profiles", cands[-1])) as f:
                tj = json.load(f)
            traffic = tj["hbm_bytes_per_frame"] * n_frames / 1e9
        except Exception:
            traffic = None
        achieved = bpf * n_frames / (kernel_ms_avg * 1e-3)
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
                            "slot %d samples, AWGN SNR %g dB, CFO +-20 ppm, device-resident"
                            % (n_frames, PSDU_LEN, n_sym, SLOT_LEN, SNR_DB),
                "frames_per_gpu": n_frames, "slot_len": SLOT_LEN, "encoding": "QPSK_1_2",
                "outputs": "48 u8 decisions + 96 f32 LLRs per data symbol, 32 B frame record"
                           + ("; decode_mac + PSDU" if do_decode else ""),
                "parallelism": ("frames sharded %d-way, no collective on the hot path" % world if world > 1 else "1 GPU")
                               + ("" if backend == "nccl" else " [REHEARSAL: %s backend]" % backend),
            },
            "gsamples_per_s": value / 1e9,
            "msymbols_per_s": float(n_frames) * (n_sym + 3) * world * args.steps / elapsed / 1e6,
            "frames_complete": n_complete,
            "frames_crc_ok": n_crc,
            "pdu_leg": pdu_leg,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                "traffic": traffic,
                "traffic_unit": "GB per launch (PMC, profiles/*_traffic.json)",
                "kernel": "wr::demod_batch_kernel",
                "kernel_ms": kernel_ms_avg,
                "algorithmic_bytes_per_frame": bpf,
            },
        }

    # ---- cpu_baseline leg: the oracle on this host's cores, bounded sample of the same batch ----
    if rank == 0 and world == 1 and not args.no_cpu:      # contract: rank 0 at N=1 only
        from oracle import oracle as orc
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
            opsdu = orc.decode_batch(of, o["idx"][:n_dec], prm, psdu_stride=psdu_stride, n_threads=cores)
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
            "kind": "port",
            "sample": "first %d frames of the GPU batch (%.1f s), oracle spec mode, OpenMP over frames" % (n_cpu, dt),
            "gpu_vs_cpu": value / (n_cpu * SLOT_LEN / dt),
        }
        result["parity"] = {"frames_checked": n_cpu, "mismatching_values": mism}

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    rx.close()


if __name__ == "__main__":
    main()
