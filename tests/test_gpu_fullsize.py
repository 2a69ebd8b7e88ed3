"""BASELINE.json configs[1] at its FULL size as a -m gpu test: 1 000 000 frames, 20 MHz QPSK 1/2, PSDU 294 B
(50 symbols), slot 4608, AWGN 20 dB, CFO +-20 ppm, device-resident (36.9 GB of samples) -- until now that evidence existed only
inside bench.py.  Through the C ABI (`wifirx_demod_batch` on device buffers):
  * every frame COMPLETE at the transmitted rate and length;
  * channel BER 0: the hard decisions equal the transmitter's interleaved coded bits (4.8 G bits at 20 dB);
  * split invariance: the last 666 667 frames as a call of their own -- the cut at frame 333 333 regroups every wave's four
    slots -- give the same bytes;
  * equality with the oracle (records, decisions, LLRs, bit for bit) on a random subset of 20 000 frames."""
import ctypes as C
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_FRAMES, SLOT, LEAD, ENC, PSDU_LEN, N_T = 1_000_000, 4608, 160, 2, 294, 256
CUT, SUBSET = 333_333, 20_000


@pytest.mark.timeout(600)
def test_config2_at_full_size(orc):
    import torch
    from wifirx import capi, txgen
    t0 = time.perf_counter()
    tx = txgen.encode_psdus(txgen.make_psdus(N_T, PSDU_LEN, seed=2025), ENC)
    n_sym, nb = tx.n_sym, txgen.RATE_TABLE[ENC][0]
    assert n_sym == 50 and nb == 2
    dev = torch.device("cuda:0")
    rx = capi.WifiRx(max_sym=n_sym, llr_bits=nb, want_carrier=False, device=0)
    iq = torch.empty((N_FRAMES, SLOT, 2), dtype=torch.float32, device=dev)
    rx.synth_slots(tx.samples, iq.data_ptr(), SLOT, N_FRAMES, LEAD, 20.0, 0.037, 1234)       # 0.037 rad / sample = 20 ppm of 5.89 GHz at 20 MHz

    def demod(first, n):
        frames = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
        idx = torch.zeros((n, n_sym * 48), dtype=torch.uint8, device=dev)
        llr = torch.zeros((n, n_sym * 48 * nb), dtype=torch.float32, device=dev)
        out = capi.Out(frames.data_ptr(), idx.data_ptr(), llr.data_ptr(), None, None, 0, 1, None, None, None)
        torch.cuda.synchronize()      # torch's zero fills run on torch's stream, the library on its own: they must have landed first
        rx._check(capi.lib().wifirx_demod_batch(rx._h, iq.data_ptr() + first * SLOT * 8, 1, SLOT, n, C.byref(out)))
        rx.sync()
        return frames, idx, llr

    frames, idx, llr = demod(0, N_FRAMES)
    fr = frames.cpu().numpy().view(capi.FRAME_DTYPE).reshape(-1)
    # ---- every frame complete, at the transmitted rate and length ----
    assert ((fr["flags"] & capi.F_COMPLETE) != 0).all()
    assert (fr["encoding"] == ENC).all() and (fr["psdu_len"] == PSDU_LEN).all() and (fr["n_sym_out"] == n_sym).all()
    # ---- channel BER: slot k carries template k mod N_T (wr_synth.hip) ----
    n_grp = N_FRAMES // N_T
    ref = torch.from_numpy(np.ascontiguousarray(tx.data_idx.reshape(N_T, n_sym * 48))).to(dev)
    wrong = 0
    for g0 in range(0, n_grp, 512):                       # in pieces: the xor of 2.4 GB at once would double the footprint
        g1 = min(n_grp, g0 + 512)
        d = idx[g0 * N_T:g1 * N_T].view(g1 - g0, N_T, n_sym * 48) ^ ref[None]
        wrong += int((d != 0).sum().item())
    assert wrong == 0, "%d of %d decisions differ from the transmitted symbols at 20 dB" % (wrong, n_grp * N_T * n_sym * 48)
    # ---- split invariance ----
    f2, i2, l2 = demod(CUT, N_FRAMES - CUT)
    assert torch.equal(i2, idx[CUT:]) and torch.equal(l2.view(torch.int32), llr[CUT:].view(torch.int32))
    fr2 = f2.cpu().numpy().view(capi.FRAME_DTYPE).reshape(-1)
    assert np.array_equal(fr2, fr[CUT:])
    del f2, i2, l2
    # ---- the oracle on a random subset ----
    pick = np.sort(np.random.default_rng(5).choice(N_FRAMES, SUBSET, replace=False))
    pick_t = torch.from_numpy(pick).to(dev)
    x = iq[pick_t].cpu().numpy().view(np.complex64).reshape(-1)
    g_idx = idx[pick_t].cpu().numpy().reshape(SUBSET, n_sym, 48)
    g_llr = llr[pick_t].cpu().numpy()
    t1 = time.perf_counter()
    o = orc.demod_batch(x, SLOT, orc.make_params(max_sym=n_sym, llr_bits=nb), n_threads=min(os.cpu_count() or 1, 64))
    t_orc = time.perf_counter() - t1
    assert np.array_equal(fr[pick], o["frames"])
    assert np.array_equal(g_idx, o["idx"])
    assert np.array_equal(g_llr.view(np.int32).reshape(-1), o["llr"].view(np.int32).reshape(-1))
    rx.close()
    del iq, idx, llr, frames
    torch.cuda.empty_cache()
    print("config 2 at full size: %.1f s (oracle on %d frames: %.1f s)" % (time.perf_counter() - t0, SUBSET, t_orc))
