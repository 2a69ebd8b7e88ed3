"""bench.py's own arithmetic (CPU): the channel-BER count against the transmitted bits, the byte formula of SURVEY.md 8(d),
and the same-run yardstick library -- measurement tooling, not product: it must load, export its two symbols, carry the
hash of the source it was built from (bench.py refuses a stale binary), and nothing of the product may depend on it."""
import ctypes as C
import hashlib
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_formula(bench):
    # config 2 = 8*4608 + 2400*(1+8) + 32 = 58 496; config 3 (slot 1472) = 25 008
    assert bench.algorithmic_bytes_per_frame(4608, 50, 2) == 58496
    assert bench.algorithmic_bytes_per_frame(1472, 11, 6) == 25008


def test_ber_vs_tx_counts_bit_errors_of_complete_frames_only(bench):
    import torch
    from wifirx import capi, txgen
    n_t, reps, enc = 8, 5, 4                                  # 16-QAM: 4 bits per carrier
    tx = txgen.encode_psdus(txgen.make_psdus(n_t, 60, seed=3), enc)
    n = n_t * reps + 3                                        # a tail that does not fill a round of templates is left out
    idx = np.tile(tx.data_idx.reshape(n_t, -1), (reps + 1, 1))[:n].copy()
    frames = np.zeros(n, capi.FRAME_DTYPE)
    frames["flags"] = capi.F_COMPLETE
    frames["encoding"] = enc
    frames["n_sym"] = tx.n_sym
    idx[1, 5] ^= 0b0101                                       # two bit errors in frame 1
    idx[10, 0] ^= 0b1000                                      # one in frame 10
    idx[12, :] ^= 0b1111                                      # frame 12 is wrong everywhere, but not complete: not compared
    frames["flags"][12] = 0
    r = bench.ber_vs_tx(torch, torch.from_numpy(idx), frames, tx, 4, capi)
    assert r["bit_errors"] == 3
    assert r["frames_compared"] == n_t * reps - 1 and r["frames_not_demodulated"] == 1
    assert r["bits_compared"] == (n_t * reps - 1) * tx.n_sym * 48 * 4
    assert r["coded_ber"] == pytest.approx(3 / r["bits_compared"])


def test_box_probe_library_is_fresh_and_not_part_of_the_product(bench):
    path = os.path.join(ROOT, "tools", "libboxprobe.so")
    if not os.path.exists(path):
        pytest.skip("tools/libboxprobe.so not built (python __graft_entry__.py)")
    lib = C.CDLL(path)
    assert hasattr(lib, "boxprobe_run") and hasattr(lib, "boxprobe_src_sha")
    lib.boxprobe_src_sha.restype = C.c_char_p
    with open(os.path.join(ROOT, "tools", "box_probe.hip"), "rb") as f:
        want = hashlib.sha256(f.read()).hexdigest()[:16]
    assert lib.boxprobe_src_sha().decode() == want, "stale tools/libboxprobe.so: run python __graft_entry__.py"
    l, sha, fresh = bench.boxprobe_lib()
    assert l is not None and fresh and sha == want
    pkg = os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd")
    for d, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".inc")) or fn == "Makefile":
                assert "boxprobe" not in open(os.path.join(d, fn), errors="ignore").read(), fn
