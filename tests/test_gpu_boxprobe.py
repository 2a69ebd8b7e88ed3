"""tools/libboxprobe.so (bench.py's same-run memory yardstick; measurement tooling, not product) on canary-fenced buffers.

VERDICT r04 item 3: the first bench.py run that carried the probe (round 4, 02:03, before commit 0cc8d4f) ended in
`Memory access fault by GPU ... on address 0x7bbf50a03000` (a page boundary).  The cause, from the code as committed 17 minutes
later and the figures of the runs either side (profiles/README.md): the probe's `stream` kernel reads piece i
of the sample buffer and WRITES piece i of the LLR buffer for i < n16 = (bytes the demod kernel reads) / 16.  On config 2 that is
32.0 GB of pieces against an LLR buffer of 19.2 GB (1 M x 50 x 48 x 2 x 4 B): the writes ran 12.8 GB past the end of `llr` and hit
the first unmapped page behind it.  The committed probe clamps the run to the shorter buffer (`cap16`, `x16` in boxprobe_run) -- the
next run's `stream_gb` = 32.1 = 19.2 GB x (1 + 43/64) is exactly that clamp at work -- and refuses a slot the pattern does not fit.
Only bench.py at full size ever called the probe; this test calls it the way bench.py does, on small buffers with fences."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

FENCE = 1 << 20          # bytes of 0xA5 on either side of every buffer
HIP_ERROR_INVALID_VALUE = 1


def _lib():
    path = os.path.join(ROOT, "tools", "libboxprobe.so")
    if not os.path.exists(path):
        pytest.skip("tools/libboxprobe.so not built (python __graft_entry__.py)")
    lib = C.CDLL(path)
    lib.boxprobe_run.restype = C.c_int
    lib.boxprobe_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.POINTER(C.c_double)]
    return lib


class Fenced:
    """`nbytes` of device memory with FENCE bytes of 0xA5 before and after it (allocated through the library's C ABI: the pytest
    process holds ONE HIP runtime, the one libwifirx.so and libboxprobe.so link; torch's bundled one stays out of it)"""

    def __init__(self, rx, nbytes, fill=0):
        self.n = int(nbytes)
        self.buf = rx.alloc(2 * FENCE + self.n)
        host = np.full(2 * FENCE + self.n, 0xA5, dtype=np.uint8)
        host[FENCE:FENCE + self.n] = fill
        self.buf.upload(host)
        self.ptr = self.buf.ptr + FENCE
        assert self.ptr % 256 == 0

    def _host(self):
        return self.buf.download(np.uint8, 2 * FENCE + self.n)

    def fences_intact(self):
        h = self._host()
        return bool((h[:FENCE] == 0xA5).all()) and bool((h[FENCE + self.n:] == 0xA5).all())

    def touched(self):
        return bool((self._host()[FENCE:FENCE + self.n] != 0).any())

    def free(self):
        self.buf.free()


@pytest.mark.parametrize("n_bpsc", [1, 2, 4, 6])
@pytest.mark.parametrize("with_car", [False, True])
def test_probe_stays_inside_its_buffers(n_bpsc, with_car):
    """n_slots not a multiple of 4 (the last wave holds one slot), a slot the pattern barely fits, every constellation, with and
    without the `carrier` rows; the sample buffer is the smallest of the three in some cases (n_bpsc 6) and the LLR buffer in
    others (n_bpsc 1): the stream leg must stop at the shorter one."""
    from wifirx import capi
    lib = _lib()
    rx = capi.WifiRx(max_sym=8)
    n_slots, lead, n_sym = 1001, 160, 7
    slot_len = lead + 192 + 128 + 80 * (n_sym + 1)             # the shortest slot boxprobe_run accepts
    x = Fenced(rx, n_slots * slot_len * 8, fill=0)
    idx = Fenced(rx, n_slots * n_sym * 48)
    llr = Fenced(rx, n_slots * n_sym * 48 * n_bpsc * 4)
    car = Fenced(rx, n_slots * n_sym * 48 * 8) if with_car else None
    out = (C.c_double * 12)()
    rx.sync()
    rc = lib.boxprobe_run(x.ptr, idx.ptr, llr.ptr, car.ptr if car else None, n_slots, slot_len, lead, n_sym, n_bpsc, 1, out)     # (synchronises the device before it returns)
    assert rc == 0
    for name, b in (("x", x), ("idx", idx), ("llr", llr), ("car", car)):
        if b is not None:
            assert b.fences_intact(), "%s: the probe wrote outside the buffer" % name
    assert not x.touched(), "the sample buffer is read-only for the probe"
    assert out[0] > 0 and out[2] > 0 and out[5] > 0
    # one sample shorter: the pattern's last symbol would leave the slot -- refused on the host, nothing launched
    rc = lib.boxprobe_run(x.ptr, idx.ptr, llr.ptr, car.ptr if car else None, n_slots, slot_len - 1, lead, n_sym, n_bpsc, 1, out)
    assert rc == HIP_ERROR_INVALID_VALUE
    assert lib.boxprobe_run(x.ptr, idx.ptr, llr.ptr, None, n_slots, slot_len, lead, n_sym, 3, 1, out) == HIP_ERROR_INVALID_VALUE
    assert lib.boxprobe_run(x.ptr, idx.ptr, llr.ptr, None, 0, slot_len, lead, n_sym, n_bpsc, 1, out) == HIP_ERROR_INVALID_VALUE
    for b in (x, idx, llr, car):
        if b is not None:
            b.free()
    rx.close()


def test_stream_leg_is_clamped_to_the_shorter_buffer():
    """The geometry of the round-4 fault in small: BPSK rows (192 B of LLRs per symbol) against 512 B of samples per symbol -- the bytes
    the kernel reads are 2.7 x what the LLR buffer holds.  out[1] (bytes the stream moved) must not exceed what the two buffers
    allow, and the fence behind `llr` must stand."""
    from wifirx import capi
    lib = _lib()
    rx = capi.WifiRx(max_sym=8)
    n_slots, lead, n_sym, n_bpsc = 4099, 160, 50, 1
    slot_len = 4608
    x = Fenced(rx, n_slots * slot_len * 8)
    idx = Fenced(rx, n_slots * n_sym * 48)
    llr = Fenced(rx, n_slots * n_sym * 48 * n_bpsc * 4)
    out = (C.c_double * 12)()
    rx.sync()
    rc = lib.boxprobe_run(x.ptr, idx.ptr, llr.ptr, None, n_slots, slot_len, lead, n_sym, n_bpsc, 1, out)
    assert rc == 0
    assert llr.fences_intact() and idx.fences_intact() and x.fences_intact()
    rd_wanted = n_slots * ((n_sym + 3) * 512 + (lead + 64) // 16 * 128 + 6 * 512)
    assert rd_wanted > 2 * llr.n                                    # the unclamped stream would have run far past `llr`
    pieces = out[1] / 16.0
    assert pieces <= 2 * (llr.n // 16) + 1                          # read + written pieces: at most one each per piece of `llr`
    for b in (x, idx, llr):
        b.free()
    rx.close()
