"""ctypes loader of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (see wifirx_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

MATH_SPEC = 0
MATH_LIBM = 1

FRAME_DTYPE = np.dtype([
    ("flags", "<u4"), ("trigger", "<i4"), ("frame_start", "<i4"),
    ("cfo_coarse", "<f4"), ("cfo_fine", "<f4"), ("snr_db", "<f4"),
    ("psdu_len", "<u2"), ("encoding", "u1"), ("n_bpsc", "u1"),
    ("n_sym", "<u2"), ("n_sym_out", "<u2"),
])
assert FRAME_DTYPE.itemsize == 32

F_DETECTED, F_SYNC, F_SIGNAL, F_COMPLETE, F_LLR, F_DECODED, F_CRC_OK = 1, 2, 4, 8, 16, 32, 64


class Params(C.Structure):
    _fields_ = [("bandwidth", C.c_double), ("frequency", C.c_double), ("threshold", C.c_float),
                ("min_plateau", C.c_int32), ("math_mode", C.c_int32), ("max_sym", C.c_int32),
                ("llr_bits", C.c_int32), ("chan_est", C.c_int32), ("llr_csi", C.c_int32),
                ("lts_search", C.c_int32), ("no_pair_fallback", C.c_int32), ("dbg_top4", C.c_void_p), ("dbg_mag4", C.c_void_p),
                ("fallback_cfo_f", C.c_float), ("libm_exact_phase", C.c_int32)]


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or \
            os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "wifirx_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B"])
    return LIB


_lib = None


def use_native_build() -> str:
    """bench.py's cpu_baseline leg: a copy of the oracle compiled with -march=native ON THIS HOST (gcc is on the GPU box
    too), so that the CPU baseline is not held back by the portable -march=x86-64-v3 of the shipped liboracle.so.
    Returns the -march the loaded library was built with.  Must be called before the first lib()."""
    global _lib
    import tempfile
    if _lib is not None:
        return "x86-64-v3"
    try:
        out = os.path.join(tempfile.mkdtemp(prefix="wifirx_oracle_"), "liboracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC",
                               "-I" + os.path.join(os.path.dirname(HERE), "include"), "-shared", "-o", out,
                               os.path.join(HERE, "wifirx_oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        _load(out)
        return "native"
    except Exception:
        _lib = None
        lib()
        return "x86-64-v3"


def _load(path):
    global _lib
    _lib = C.CDLL(path)
    _lib.orc_sync_short.restype = C.c_long
    _lib.orc_demod_stream.restype = C.c_long
    _lib.orc_crc32.restype = C.c_uint32
    return _lib


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _load(LIB)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_params(bandwidth=20e6, frequency=5.89e9, threshold=0.56, min_plateau=2,
                math_mode=MATH_SPEC, max_sym=64, llr_bits=0, chan_est=0, llr_csi=0, lts_search=0, no_pair_fallback=0,
                libm_exact_phase=0) -> Params:
    """lts_search (SPEC mode): 0 = rule 6 as the kernels run it, 1 = the exhaustive float32 search over all 320 lags.
    no_pair_fallback: 1 = frames whose LTS search finds no pair are decoded at offset 320 with the previous frame's fine CFO,
    as upstream's sync_long does (a measurement switch; the build drops them).
    libm_exact_phase (LIBM mode): 1 = the derotation angles formed in double instead of upstream's float32 products (a measurement
    switch: isolates the angle rounding in the distance between the GPU and the upstream-literal arithmetic)."""
    return Params(bandwidth, frequency, threshold, min_plateau, math_mode, max_sym, llr_bits, chan_est, llr_csi,
                  lts_search, no_pair_fallback, None, None, 0.0, libm_exact_phase)


def demod_batch(iq: np.ndarray, slot_len: int, prm: Params, want_eq=False, n_threads=1, want_csi=False):
    """iq: complex64 [n_slots*slot_len]. Returns dict(frames, idx, llr, eq)."""
    iq = np.ascontiguousarray(iq, dtype=np.complex64).reshape(-1)
    n_slots = iq.size // slot_len
    frames = np.zeros(n_slots, dtype=FRAME_DTYPE)
    idx = np.zeros((n_slots, prm.max_sym, 48), dtype=np.uint8)
    llr = np.zeros((n_slots, prm.max_sym * 48 * prm.llr_bits), dtype=np.float32) if prm.llr_bits else None
    eq = np.zeros((n_slots, prm.max_sym, 48), dtype=np.complex64) if want_eq else None
    csi = np.zeros((n_slots, 52), dtype=np.complex64) if want_csi else None
    rc = lib().orc_demod_batch(_p(iq), C.c_uint32(slot_len), C.c_uint32(n_slots), C.byref(prm),
                               _p(frames), _p(idx), _p(llr), _p(eq), _p(csi), C.c_int(n_threads))
    assert rc == 0
    return dict(frames=frames, idx=idx, llr=llr, eq=eq, csi=csi)


def decode_batch(frames: np.ndarray, idx: np.ndarray, prm: Params, psdu_stride=2048, n_threads=1):
    n = frames.shape[0]
    psdu = np.zeros((n, psdu_stride), dtype=np.uint8)
    rc = lib().orc_decode_batch(C.c_uint32(n), C.byref(prm), _p(frames), _p(idx), _p(psdu),
                                C.c_uint32(psdu_stride), C.c_int(n_threads))
    assert rc == 0
    return psdu


def demod_stream(x: np.ndarray, prm: Params, want_eq=False, cap=4096):
    x = np.ascontiguousarray(x, dtype=np.complex64).reshape(-1)
    frames = np.zeros(cap, dtype=FRAME_DTYPE)
    idx = np.zeros((cap, prm.max_sym, 48), dtype=np.uint8)
    llr = np.zeros((cap, prm.max_sym * 48 * prm.llr_bits), dtype=np.float32) if prm.llr_bits else None
    eq = np.zeros((cap, prm.max_sym, 48), dtype=np.complex64) if want_eq else None
    n = lib().orc_demod_stream(_p(x), C.c_long(x.size), C.byref(prm), _p(frames), _p(idx), _p(llr),
                               _p(eq), C.c_long(cap))
    return dict(frames=frames[:n], idx=idx[:n], llr=None if llr is None else llr[:n],
                eq=None if eq is None else eq[:n])


def sym_stats(eq: np.ndarray, n_sym_out: np.ndarray) -> np.ndarray:
    """eq: [n, max_sym, 48] equalised points, n_sym_out [n] -> [n, 4] float32 (sum |y|, sum |y|^2, sum |y|^4, 0)"""
    eq = np.ascontiguousarray(eq, dtype=np.complex64)
    out = np.zeros((eq.shape[0], 4), dtype=np.float32)
    for i in range(eq.shape[0]):
        row = np.ascontiguousarray(eq[i])
        lib().orc_sym_stats(_p(row), C.c_int(int(n_sym_out[i])), _p(out[i:i + 1]))
    return out


def sync_short(x: np.ndarray, threshold=0.56, min_plateau=2, math_mode=MATH_SPEC, first_only=False, cap=4096):
    x = np.ascontiguousarray(x, dtype=np.complex64).reshape(-1)
    trig = np.zeros(cap, dtype=np.int32)
    cfo = np.zeros(cap, dtype=np.float32)
    n = lib().orc_sync_short(_p(x), C.c_long(x.size), C.c_float(threshold), C.c_int(min_plateau),
                             C.c_int(math_mode), C.c_int(int(first_only)), _p(trig), _p(cfo), C.c_long(cap))
    return trig[:n].copy(), cfo[:n].copy()


def decode_mac(idx: np.ndarray, encoding: int, psdu_len: int):
    idx = np.ascontiguousarray(idx, dtype=np.uint8).reshape(-1)
    out = np.zeros(max(psdu_len, 1), dtype=np.uint8)
    rc = lib().orc_decode_mac(_p(idx), C.c_int(encoding), C.c_int(psdu_len), _p(out))
    return rc, out[:psdu_len]


def sincos(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    s = np.empty_like(x); c = np.empty_like(x)
    lib().orc_sincos(_p(x), _p(s), _p(c), C.c_long(x.size))
    return s, c


def atan2(y, x):
    y = np.ascontiguousarray(y, dtype=np.float32); x = np.ascontiguousarray(x, dtype=np.float32)
    r = np.empty_like(x)
    lib().orc_atan2(_p(y), _p(x), _p(r), C.c_long(x.size))
    return r


def log2(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    r = np.empty_like(x)
    lib().orc_log2(_p(x), _p(r), C.c_long(x.size))
    return r


def rsqrt(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    r = np.empty_like(x)
    lib().orc_rsqrt(_p(x), _p(r), C.c_long(x.size))
    return r


def recip(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    r = np.empty_like(x)
    lib().orc_recip(_p(x), _p(r), C.c_long(x.size))
    return r


def fft64(x, math_mode=MATH_SPEC):
    x = np.ascontiguousarray(x, dtype=np.complex64).reshape(-1, 64)
    out = np.empty_like(x)
    lib().orc_fft64(_p(x), _p(out), C.c_long(x.shape[0]), C.c_int(math_mode))
    return out


def viterbi(coded, n_bits):
    coded = np.ascontiguousarray(coded, dtype=np.uint8)
    out = np.zeros(n_bits, dtype=np.uint8)
    lib().orc_viterbi(_p(coded), C.c_int(n_bits), _p(out))
    return out


def crc32(data: bytes) -> int:
    a = np.frombuffer(data, dtype=np.uint8)
    return int(lib().orc_crc32(_p(a), C.c_long(a.size)))


def decode_signal(rx48):
    rx48 = np.ascontiguousarray(rx48, dtype=np.uint8)
    enc = C.c_int(0); ln = C.c_int(0)
    ok = lib().orc_decode_signal(_p(rx48), C.byref(enc), C.byref(ln))
    return bool(ok), enc.value, ln.value
