"""BASELINE.json configs[1] at its FULL size as a -m gpu test: 1 000 000 frames, 20 MHz QPSK 1/2, PSDU 294 B
(50 symbols), slot 4608, AWGN 20 dB, CFO +-20 ppm, device-resident (36.9 GB of samples) -- until now that evidence existed only
inside bench.py.  Through the C ABI (`wifirx_demod_batch` on device buffers; device memory through `wifirx_dev_alloc`: the pytest
process holds one HIP runtime, the library's):
  * every frame COMPLETE at the transmitted rate and length;
  * channel BER 0: the hard decisions equal the transmitter's interleaved coded bits (4.8 G bits at 20 dB);
  * split invariance: the last 666 667 frames as a call of their own -- the cut at frame 333 333 regroups every wave's four
    slots -- give the same bytes (records, decisions, LLRs);
  * equality with the oracle (records, decisions, LLRs, bit for bit) on a random subset of 20 000 frames."""
import ctypes as C
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_FRAMES, SLOT, LEAD, ENC, PSDU_LEN, N_T = 1_000_000, 4608, 160, 2, 294, 256
CUT, SUBSET, CHUNK = 333_333, 20_000, 50_000


@pytest.mark.timeout(900)
def test_config2_at_full_size(orc):
    from wifirx import capi, txgen
    lib = capi.lib()
    t0 = time.perf_counter()
    tx = txgen.encode_psdus(txgen.make_psdus(N_T, PSDU_LEN, seed=2025), ENC)
    n_sym, nb = tx.n_sym, txgen.RATE_TABLE[ENC][0]
    assert n_sym == 50 and nb == 2
    rx = capi.WifiRx(max_sym=n_sym, llr_bits=nb, want_carrier=False, device=0)
    iq = rx.alloc(N_FRAMES * SLOT * 8)
    rx.synth_slots(tx.samples, iq.ptr, SLOT, N_FRAMES, LEAD, 20.0, 0.037, 1234)       # 0.037 rad / sample = 20 ppm of 5.89 GHz at 20 MHz
    row_i, row_l = n_sym * 48, n_sym * 48 * nb * 4                                       # bytes of decisions / of LLRs per frame

    def demod(first, n):
        bufs = dict(frames=rx.alloc(n * 32), idx=rx.alloc(n * row_i), llr=rx.alloc(n * row_l))
        out = capi.Out(bufs["frames"].ptr, bufs["idx"].ptr, bufs["llr"].ptr, None, None, 0, 1, None, None, None)
        rx._check(lib.wifirx_demod_batch(rx._h, iq.ptr + first * SLOT * 8, 1, SLOT, n, C.byref(out)))
        rx.sync()
        return bufs

    def rows(buf, first, n, row_bytes, dtype):
        """frames [first, first + n) of a device array, on the host"""
        out = np.empty(n * row_bytes // np.dtype(dtype).itemsize, dtype=dtype)
        rx._check(lib.wifirx_memcpy_d2h(rx._h, out.ctypes.data_as(C.c_void_p), buf.ptr + first * row_bytes, out.nbytes))
        return out

    a = demod(0, N_FRAMES)
    fr = rows(a["frames"], 0, N_FRAMES, 32, np.uint8).view(capi.FRAME_DTYPE).reshape(-1)
    # ---- every frame complete, at the transmitted rate and length ----
    assert ((fr["flags"] & capi.F_COMPLETE) != 0).all()
    assert (fr["encoding"] == ENC).all() and (fr["psdu_len"] == PSDU_LEN).all() and (fr["n_sym_out"] == n_sym).all()
    b = demod(CUT, N_FRAMES - CUT)
    fr2 = rows(b["frames"], 0, N_FRAMES - CUT, 32, np.uint8).view(capi.FRAME_DTYPE).reshape(-1)
    assert np.array_equal(fr2, fr[CUT:])
    # ---- channel BER (slot k carries template k mod N_T, wr_synth.hip) and split invariance, chunk by chunk ----
    ref = np.ascontiguousarray(tx.data_idx.reshape(N_T, row_i))
    wrong = 0
    for f0 in range(0, N_FRAMES, CHUNK):
        n = min(CHUNK, N_FRAMES - f0)
        ia = rows(a["idx"], f0, n, row_i, np.uint8).reshape(n, row_i)
        wrong += int((ia != ref[(f0 + np.arange(n)) % N_T]).sum())
        g0 = max(f0, CUT)                        # the part of this chunk that the second call covers too
        if g0 < f0 + n:
            m = f0 + n - g0
            ib = rows(b["idx"], g0 - CUT, m, row_i, np.uint8).reshape(m, row_i)
            assert np.array_equal(ib, ia[g0 - f0:]), "decisions depend on how the batch is cut (frames %d..)" % g0
            la = rows(a["llr"], g0, m, row_l, np.uint32)
            lb = rows(b["llr"], g0 - CUT, m, row_l, np.uint32)
            assert np.array_equal(la, lb), "LLRs depend on how the batch is cut (frames %d..)" % g0
    assert wrong == 0, "%d of %d decisions differ from the transmitted symbols at 20 dB" % (wrong, N_FRAMES * row_i)
    for v in b.values():
        v.free()
    # ---- the oracle on a random subset ----
    pick = np.sort(np.random.default_rng(5).choice(N_FRAMES, SUBSET, replace=False))
    x = np.empty((SUBSET, SLOT), dtype=np.complex64)
    g_idx = np.empty((SUBSET, n_sym, 48), dtype=np.uint8)
    g_llr = np.empty((SUBSET, row_l // 4), dtype=np.uint32)
    for k, f in enumerate(pick):
        x[k] = rows(iq, int(f), 1, SLOT * 8, np.complex64)
        g_idx[k] = rows(a["idx"], int(f), 1, row_i, np.uint8).reshape(n_sym, 48)
        g_llr[k] = rows(a["llr"], int(f), 1, row_l, np.uint32)
    t1 = time.perf_counter()
    o = orc.demod_batch(x.reshape(-1), SLOT, orc.make_params(max_sym=n_sym, llr_bits=nb), n_threads=min(os.cpu_count() or 1, 64))
    t_orc = time.perf_counter() - t1
    assert np.array_equal(fr[pick], o["frames"])
    assert np.array_equal(g_idx, o["idx"])
    assert np.array_equal(g_llr.reshape(-1), o["llr"].view(np.uint32).reshape(-1))
    for v in a.values():
        v.free()
    iq.free()
    rx.close()
    print("config 2 at full size: %.1f s (oracle on %d frames: %.1f s)" % (time.perf_counter() - t0, SUBSET, t_orc))
