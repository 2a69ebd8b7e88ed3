"""ctypes binding of libwifirx.so (include/wifirx.h) -- the product's only compute path.

There is no CPU fallback: importing works everywhere (so that the ABI can be inspected on a
CPU-only box), but creating a :class:`WifiRx` raises :class:`WifiRxError` unless a gfx950 GPU
is usable, and a missing ``libwifirx.so`` raises at import of this module.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WIFIRX_LIB") or os.path.join(_HERE, "libwifirx.so")     # WIFIRX_LIB: A/B builds

ABI_VERSION = 4
DECODE_INPUT = "planes"      # what demod_batch(decode=True) hands decode_mac: "planes" (wifirx_out.hbits) or "idx"
EQ_LS, EQ_LMS, EQ_COMB, EQ_STA = 0, 1, 2, 3
OK, EINVAL, ENODEV, ENOMEM, EHIP, ERANGE, EDEAD = 0, -1, -2, -3, -4, -5, -6      # WIFIRX_E*; EDEAD: the stream is dead, do not retry
STREAM_BATCH_MAX = 1 << 27          # WIFIRX_STREAM_BATCH_MAX
P_BANDWIDTH, P_FREQUENCY, P_SENSITIVITY, P_CHAN_EST, P_STREAM_BATCH, P_DECODE_SMALL_MAX, P_LLR_CSI, P_STREAM_IDX = 1, 2, 3, 4, 5, 6, 7, 8
F_DETECTED, F_SYNC, F_SIGNAL, F_COMPLETE, F_LLR, F_DECODED, F_CRC_OK = 1, 2, 4, 8, 16, 32, 64

FRAME_DTYPE = np.dtype([
    ("flags", "<u4"), ("trigger", "<i4"), ("frame_start", "<i4"),
    ("cfo_coarse", "<f4"), ("cfo_fine", "<f4"), ("snr_db", "<f4"),
    ("psdu_len", "<u2"), ("encoding", "u1"), ("n_bpsc", "u1"),
    ("n_sym", "<u2"), ("n_sym_out", "<u2"),
])
assert FRAME_DTYPE.itemsize == 32

EXPORTS = [
    "wifirx_create", "wifirx_destroy", "wifirx_last_error", "wifirx_abi_version", "wifirx_set_param",
    "wifirx_get_stats", "wifirx_demod_batch", "wifirx_decode_batch", "wifirx_push", "wifirx_poll", "wifirx_poll_csi",
    "wifirx_sync", "wifirx_stream", "wifirx_synth_slots", "wifirx_dev_alloc", "wifirx_dev_free",
    "wifirx_memcpy_h2d", "wifirx_memcpy_d2h", "wifirx_time_demod", "wifirx_poll_ex", "wifirx_demod_batch_v",
    "wifirx_push_consumed", "wifirx_queued",
]


class WifiRxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("wifirx error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("bandwidth", C.c_double),
                ("frequency", C.c_double), ("sensitivity", C.c_float), ("min_plateau", C.c_int32),
                ("chan_est", C.c_int32), ("max_sym", C.c_uint32), ("llr_bits", C.c_uint32),
                ("want_carrier", C.c_uint32), ("max_batch", C.c_uint32), ("max_slot_len", C.c_uint32)]


class Out(C.Structure):
    _fields_ = [("frames", C.c_void_p), ("idx", C.c_void_p), ("llr", C.c_void_p), ("carrier", C.c_void_p),
                ("psdu", C.c_void_p), ("psdu_stride", C.c_uint32), ("on_device", C.c_uint32), ("csi", C.c_void_p),
                ("sym_stats", C.c_void_p), ("hbits", C.c_void_p)]


class PollOut(C.Structure):
    _fields_ = [("frames", C.c_void_p), ("psdu", C.c_void_p), ("psdu_stride", C.c_uint32), ("reserved", C.c_uint32),
                ("idx", C.c_void_p), ("carrier", C.c_void_p), ("csi", C.c_void_p), ("sym_stats", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("samples_in", C.c_uint64), ("frames_detected", C.c_uint64), ("frames_signal_ok", C.c_uint64),
                ("frames_complete", C.c_uint64), ("frames_crc_ok", C.c_uint64), ("frames_dropped", C.c_uint64)]


if not os.path.exists(LIB_PATH):
    raise ImportError("libwifirx.so is missing (%s): build it with `python __graft_entry__.py` or "
                      "`make -C gnuradio-wifi-imagetransfer_amd/csrc`; there is no CPU fallback" % LIB_PATH)

_lib = C.CDLL(LIB_PATH)
_lib.wifirx_last_error.restype = C.c_char_p
_lib.wifirx_last_error.argtypes = [C.c_void_p]
_lib.wifirx_stream.restype = C.c_void_p
_lib.wifirx_stream.argtypes = [C.c_void_p]
_lib.wifirx_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
_lib.wifirx_destroy.argtypes = [C.c_void_p]
_lib.wifirx_set_param.argtypes = [C.c_void_p, C.c_int, C.c_double]
_lib.wifirx_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
_lib.wifirx_demod_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(Out)]
_lib.wifirx_decode_batch.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Out)]
_lib.wifirx_demod_batch_v.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(Out)]
_lib.wifirx_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
_lib.wifirx_push_consumed.restype = C.c_size_t
_lib.wifirx_queued.restype = C.c_uint32
_lib.wifirx_queued.argtypes = [C.c_void_p]
_lib.wifirx_push_consumed.argtypes = [C.c_void_p]
_lib.wifirx_poll.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                             C.c_uint32, C.POINTER(C.c_uint32)]
_lib.wifirx_poll_csi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_uint32, C.POINTER(C.c_uint32)]
_lib.wifirx_poll_ex.argtypes = [C.c_void_p, C.POINTER(PollOut), C.c_uint32, C.POINTER(C.c_uint32)]
_lib.wifirx_sync.argtypes = [C.c_void_p]
_lib.wifirx_synth_slots.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p,
                                    C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_uint64,
                                    C.c_void_p]
_lib.wifirx_dev_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
_lib.wifirx_dev_free.argtypes = [C.c_void_p, C.c_void_p]
_lib.wifirx_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
_lib.wifirx_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
_lib.wifirx_time_demod.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Out), C.c_int,
                                   C.POINTER(C.c_float)]


def lib():
    return _lib


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DevBuf:
    """A device allocation owned through the C ABI (no torch needed)."""

    def __init__(self, rx: "WifiRx", nbytes: int):
        self.rx, self.nbytes = rx, int(nbytes)
        p = C.c_void_p()
        rx._check(_lib.wifirx_dev_alloc(rx._h, max(self.nbytes, 16), C.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.rx._check(_lib.wifirx_memcpy_h2d(self.rx._h, self.ptr, _np_ptr(arr), arr.nbytes))
        return self

    def download(self, dtype, count) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.rx._check(_lib.wifirx_memcpy_d2h(self.rx._h, _np_ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            _lib.wifirx_dev_free(self.rx._h, self.ptr)
            self.ptr = None


class WifiRx:
    """One receive chain = one handle = one HIP stream."""

    def __init__(self, bandwidth=20e6, frequency=5.89e9, sensitivity=0.56, min_plateau=2, chan_est=EQ_LS,
                 max_sym=64, llr_bits=0, want_carrier=False, device=0, max_batch=0, max_slot_len=0):
        self.cfg = Config(ABI_VERSION, device, bandwidth, frequency, sensitivity, min_plateau, chan_est,
                          max_sym, llr_bits, int(bool(want_carrier)), max_batch, max_slot_len)
        h = C.c_void_p()
        rc = _lib.wifirx_create(C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise WifiRxError(rc, _lib.wifirx_last_error(None).decode())
        self._h = h

    # -- plumbing --
    def _check(self, rc):
        if rc != 0:
            raise WifiRxError(rc, _lib.wifirx_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            _lib.wifirx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def max_sym(self):
        return self.cfg.max_sym

    @property
    def llr_bits(self):
        return self.cfg.llr_bits

    def set_param(self, pid, value):
        self._check(_lib.wifirx_set_param(self._h, pid, float(value)))

    def stats(self) -> dict:
        st = Stats()
        self._check(_lib.wifirx_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in Stats._fields_}

    def sync(self):
        self._check(_lib.wifirx_sync(self._h))

    def stream_ptr(self):
        return _lib.wifirx_stream(self._h)

    def alloc(self, nbytes) -> DevBuf:
        return DevBuf(self, nbytes)

    # -- batch mode, host buffers (PCIe-bound convenience path) --
    def demod_batch(self, iq: np.ndarray, slot_len: int, decode=False, psdu_stride=2048, want_csi=False,
                    want_stats=False, want_hbits=False) -> dict:
        iq = np.ascontiguousarray(iq, dtype=np.complex64).reshape(-1)
        n_slots = iq.size // slot_len
        assert n_slots * slot_len == iq.size
        ms = self.cfg.max_sym
        frames = np.zeros(n_slots, dtype=FRAME_DTYPE)
        idx = np.zeros((n_slots, ms, 48), dtype=np.uint8)
        llr = np.zeros((n_slots, ms * 48 * self.cfg.llr_bits), dtype=np.float32) if self.cfg.llr_bits else None
        car = np.zeros((n_slots, ms, 48), dtype=np.complex64) if self.cfg.want_carrier else None
        psdu = np.zeros((n_slots, psdu_stride), dtype=np.uint8) if decode else None
        if not decode:
            csi = np.zeros((n_slots, 52), dtype=np.complex64) if want_csi else None
            stats = np.zeros((n_slots, 4), dtype=np.float32) if want_stats else None
            hbits = np.zeros((n_slots, ms * 12), dtype=np.uint32) if want_hbits else None
            out = Out(_np_ptr(frames), _np_ptr(idx), _np_ptr(llr), _np_ptr(car), None, 0, 0, _np_ptr(csi), _np_ptr(stats),
                      _np_ptr(hbits))
            self._check(_lib.wifirx_demod_batch(self._h, _np_ptr(iq), 0, slot_len, n_slots, C.byref(out)))
            return dict(frames=frames, idx=idx, llr=llr, carrier=car, psdu=None, csi=csi, sym_stats=stats, hbits=hbits)
        # decode needs the decisions on the device: run on device buffers, then download
        d_iq = self.alloc(iq.nbytes).upload(iq)
        # decode_mac reads the decisions as bit planes written by the demod kernel (DECODE_INPUT = "planes"), or -- for
        # callers that only hold `idx` -- packs them itself ("idx")
        dev = self.alloc_out(n_slots, psdu_stride=psdu_stride, want_csi=want_csi, want_stats=want_stats,
                             want_hbits=want_hbits or DECODE_INPUT == "planes")
        try:
            self.demod_batch_dev(d_iq.ptr, slot_len, n_slots, dev)
            self.decode_batch_dev(n_slots, dev)
            self.sync()
            return self.download_out(dev, n_slots)
        finally:
            d_iq.free()
            self.free_out(dev)

    def demod_batch_var(self, iq: np.ndarray, slot_off) -> dict:
        """batch mode over slots of unequal length: slot k = iq[slot_off[k] : slot_off[k+1]] (host buffers)"""
        iq = np.ascontiguousarray(iq, dtype=np.complex64).reshape(-1)
        off = np.ascontiguousarray(slot_off, dtype=np.uint64)
        n_slots = off.size - 1
        assert n_slots >= 0 and int(off[-1]) <= iq.size
        ms = self.cfg.max_sym
        frames = np.zeros(n_slots, dtype=FRAME_DTYPE)
        idx = np.zeros((n_slots, ms, 48), dtype=np.uint8)
        llr = np.zeros((n_slots, ms * 48 * self.cfg.llr_bits), dtype=np.float32) if self.cfg.llr_bits else None
        car = np.zeros((n_slots, ms, 48), dtype=np.complex64) if self.cfg.want_carrier else None
        out = Out(_np_ptr(frames), _np_ptr(idx), _np_ptr(llr), _np_ptr(car), None, 0, 0, None, None)
        self._check(_lib.wifirx_demod_batch_v(self._h, _np_ptr(iq), 0, _np_ptr(off), n_slots, C.byref(out)))
        return dict(frames=frames, idx=idx, llr=llr, carrier=car)

    # -- batch mode, device buffers (the measured path) --
    def alloc_out(self, n_slots, psdu_stride=0, want_csi=False, want_stats=False, want_hbits=False, want_idx=True) -> dict:
        ms = self.cfg.max_sym
        d = dict(n_slots=n_slots, psdu_stride=psdu_stride)
        d["csi"] = self.alloc(n_slots * 52 * 8) if want_csi else None
        d["sym_stats"] = self.alloc(n_slots * 16) if want_stats else None
        d["frames"] = self.alloc(n_slots * 32)
        d["idx"] = self.alloc(n_slots * ms * 48) if want_idx else None
        d["hbits"] = self.alloc(n_slots * ms * 48) if want_hbits else None
        d["llr"] = self.alloc(n_slots * ms * 48 * self.cfg.llr_bits * 4) if self.cfg.llr_bits else None
        d["carrier"] = self.alloc(n_slots * ms * 48 * 8) if self.cfg.want_carrier else None
        d["psdu"] = self.alloc(n_slots * psdu_stride) if psdu_stride else None
        for k in ("frames", "idx", "llr", "carrier", "psdu", "csi", "sym_stats", "hbits"):      # the kernels only write what a frame fills
            if d[k] is not None and d[k].nbytes:
                d[k].upload(np.zeros(d[k].nbytes, dtype=np.uint8))
        return d

    def free_out(self, dev):
        for k in ("frames", "idx", "llr", "carrier", "psdu", "csi", "sym_stats", "hbits"):
            if dev.get(k) is not None:
                dev[k].free()

    def _out_struct(self, dev) -> Out:
        g = lambda k: dev[k].ptr if dev.get(k) is not None else None
        return Out(g("frames"), g("idx"), g("llr"), g("carrier"), g("psdu"), dev.get("psdu_stride", 0), 1, g("csi"), g("sym_stats"),
                   g("hbits"))

    def demod_batch_dev(self, iq_ptr, slot_len, n_slots, dev):
        out = self._out_struct(dev)
        self._check(_lib.wifirx_demod_batch(self._h, iq_ptr, 1, slot_len, n_slots, C.byref(out)))

    def decode_batch_dev(self, n_slots, dev):
        out = self._out_struct(dev)
        self._check(_lib.wifirx_decode_batch(self._h, n_slots, C.byref(out)))

    def time_demod(self, iq_ptr, slot_len, n_slots, dev, iters=1) -> float:
        out = self._out_struct(dev)
        ms = C.c_float(0)
        self._check(_lib.wifirx_time_demod(self._h, iq_ptr, slot_len, n_slots, C.byref(out), iters, C.byref(ms)))
        return ms.value

    def download_out(self, dev, n_slots) -> dict:
        ms = self.cfg.max_sym
        r = dict(frames=dev["frames"].download(FRAME_DTYPE, n_slots), idx=None,
                 llr=None, carrier=None, psdu=None, csi=None, sym_stats=None, hbits=None)
        if dev.get("idx") is not None:
            r["idx"] = dev["idx"].download(np.uint8, n_slots * ms * 48).reshape(n_slots, ms, 48)
        if dev.get("hbits") is not None:
            r["hbits"] = dev["hbits"].download(np.uint32, n_slots * ms * 12).reshape(n_slots, ms * 12)
        if dev.get("sym_stats") is not None:
            r["sym_stats"] = dev["sym_stats"].download(np.float32, n_slots * 4).reshape(n_slots, 4)
        if dev.get("csi") is not None:
            r["csi"] = dev["csi"].download(np.complex64, n_slots * 52).reshape(n_slots, 52)
        if dev.get("llr") is not None:
            r["llr"] = dev["llr"].download(np.float32, n_slots * ms * 48 * self.cfg.llr_bits).reshape(n_slots, -1)
        if dev.get("carrier") is not None:
            r["carrier"] = dev["carrier"].download(np.complex64, n_slots * ms * 48).reshape(n_slots, ms, 48)
        if dev.get("psdu") is not None:
            r["psdu"] = dev["psdu"].download(np.uint8, n_slots * dev["psdu_stride"]).reshape(n_slots, -1)
        return r

    def synth_slots(self, templates: np.ndarray, slots_ptr, slot_len, n_slots, lead, snr_db, cfo_max, seed,
                    cfo_out_ptr=None):
        templates = np.ascontiguousarray(templates, dtype=np.complex64)
        n_t, flen = templates.shape
        self._check(_lib.wifirx_synth_slots(self._h, _np_ptr(templates), 0, n_t, flen, slots_ptr, slot_len,
                                            n_slots, lead, snr_db, cfo_max, seed, cfo_out_ptr))

    # -- stream mode --
    def push(self, iq: np.ndarray):
        iq = np.ascontiguousarray(iq, dtype=np.complex64).reshape(-1)
        self._check(_lib.wifirx_push(self._h, _np_ptr(iq), iq.size, 0))

    def flush(self):
        self._check(_lib.wifirx_push(self._h, None, 0, 0))

    def queued(self) -> int:
        """finished frames waiting for poll()"""
        return int(_lib.wifirx_queued(self._h))

    def push_consumed(self) -> int:
        """Leading samples of the last push() the stream has taken over (after an error: where to repeat it from)."""
        return int(_lib.wifirx_push_consumed(self._h))

    def poll(self, cap=256, psdu_stride=2048, want_idx=False, want_csi=False, want_stats=False, trim_psdu=False):
        """Finished frames of the stream, oldest first (at most `cap`).  The landing buffers are kept between calls
        (a scheduler polls after every work()); what is returned are copies of the filled part."""
        ms = self.cfg.max_sym
        key = (cap, psdu_stride, bool(want_idx), bool(want_csi), bool(want_stats))
        if getattr(self, "_poll_key", None) != key:
            self._poll_key = key
            self._poll_buf = (np.zeros(cap, dtype=FRAME_DTYPE), np.zeros((cap, psdu_stride), dtype=np.uint8),
                              np.zeros((cap, ms, 48), dtype=np.uint8) if want_idx else None,
                              np.zeros((cap, ms, 48), dtype=np.complex64) if self.cfg.want_carrier else None,
                              np.zeros((cap, 52), dtype=np.complex64) if want_csi else None,
                              np.zeros((cap, 4), dtype=np.float32) if want_stats else None)
            frames, psdu, idx, car, csi, stats = self._poll_buf
            self._poll_out = PollOut(_np_ptr(frames), _np_ptr(psdu), psdu_stride, 0, _np_ptr(idx), _np_ptr(car),
                                     _np_ptr(csi), _np_ptr(stats))
        frames, psdu, idx, car, csi, stats = self._poll_buf
        n = C.c_uint32(0)
        self._check(_lib.wifirx_poll_ex(self._h, C.byref(self._poll_out), cap, C.byref(n)))
        n = n.value
        # trim_psdu: the rows are cut to the longest PSDU of the call before they are copied (a frame usually fills a fraction
        # of its 2048-byte row; the block polls this way)
        w = psdu_stride
        if trim_psdu and n:
            w = min(max(int(frames["psdu_len"][:n].max()), 1), psdu_stride)
        return dict(frames=frames[:n].copy(), psdu=psdu[:n, :w].copy(), idx=None if idx is None else idx[:n].copy(),
                    carrier=None if car is None else car[:n].copy(), csi=None if csi is None else csi[:n].copy(),
                    sym_stats=None if stats is None else stats[:n].copy())
