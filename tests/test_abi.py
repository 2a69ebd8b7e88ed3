"""The C ABI on a box without a GPU: the library loads, exports every symbol the header declares,
the record layouts match, and the product refuses to run without the device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "wifirx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wifirx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from wifirx import capi
    syms = header_symbols()
    assert len(syms) >= 18 and set(syms) == set(capi.EXPORTS)
    for s in syms:
        assert hasattr(capi.lib(), s), s
    assert capi.lib().wifirx_abi_version() == capi.ABI_VERSION == 4


def test_record_layouts():
    from wifirx import capi
    assert capi.FRAME_DTYPE.itemsize == 32
    assert [capi.FRAME_DTYPE.fields[n][1] for n in capi.FRAME_DTYPE.names] == [0, 4, 8, 12, 16, 20, 24, 26, 27, 28, 30]
    # the C compiler's view of include/wifirx.h (plain C: the header must compile as C99)
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "wifirx.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n", sizeof(wifirx_frame), sizeof(wifirx_config), sizeof(wifirx_out), sizeof(wifirx_stats), sizeof(wifirx_poll_out));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        sizes = list(map(int, subprocess.check_output([os.path.join(d, "t")]).split()))
    assert sizes == [32, ctypes.sizeof(capi.Config), ctypes.sizeof(capi.Out), ctypes.sizeof(capi.Stats),
                     ctypes.sizeof(capi.PollOut)] == [32, 56, 72, 48, 56]
    from oracle import oracle
    assert oracle.FRAME_DTYPE == capi.FRAME_DTYPE


def test_no_cpu_fallback(gpu_available):
    from wifirx import capi
    if gpu_available:
        pytest.skip("GPU present")
    with pytest.raises(capi.WifiRxError) as e:
        capi.WifiRx()
    assert e.value.code == -2          # WIFIRX_ENODEV
    from wifirx import block
    with pytest.raises(capi.WifiRxError):
        block.wifi_phy_rx()


def test_create_argument_checks():
    from wifirx import capi
    h = ctypes.c_void_p()
    bad = capi.Config(99, 0, 20e6, 5.89e9, 0.56, 2, 0, 64, 0, 0, 0, 0)
    assert capi.lib().wifirx_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    for kw in (dict(max_sym=0), dict(max_sym=512), dict(llr_bits=3), dict(chan_est=4), dict(chan_est=-1), dict(bandwidth=0.0)):
        vals = dict(abi_version=2, device=0, bandwidth=20e6, frequency=5.89e9, sensitivity=0.56, min_plateau=2,
                    chan_est=0, max_sym=64, llr_bits=0, want_carrier=0, max_batch=0, max_slot_len=0)
        vals.update(kw)
        cfg = capi.Config(*[vals[f] for f, _ in capi.Config._fields_])
        assert capi.lib().wifirx_create(ctypes.byref(cfg), ctypes.byref(h)) == -1, kw
        assert b"" != capi.lib().wifirx_last_error(None)


def test_tools_and_examples_do_not_touch_the_oracle():
    """only tests/ (its campaign scripts included), __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/"""
    for sub in ("tools", "examples"):
        for dp, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".sh", ".hip")):
                    txt = open(os.path.join(dp, f)).read()
                    assert "from oracle" not in txt and "import oracle" not in txt and "liboracle" not in txt, (dp, f)


def test_product_does_not_touch_the_oracle():
    """nothing under the package or bench's timed path imports oracle/"""
    pkg = os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("The oracle", "").replace("CPU oracle", "") \
                    or "import" not in txt.split("oracle")[0][-40:], (dp, f)
                assert "from oracle" not in txt and "import oracle" not in txt and "liboracle" not in txt, (dp, f)
