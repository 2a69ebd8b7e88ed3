"""INTEGRATION.md section 2 shows the ctypes binding a maintainer pastes into a `gr.sync_block`
(`/root/reference/gnu_radio/IRS_AP_epy_block_2.py:11-22` is the shape it mimics).  That text is executed here AS WRITTEN:
the code block is cut out of the Markdown file and run with `gnuradio` / `pmt` mapped onto wifirx.grshim (GNU Radio is
absent in this image).  On the CPU: it parses, binds every symbol it names and declares the ports / setters of
`wifi_phy_hier.grc:587-690`.  On the MI355X: BASELINE config 1's pieces go through THAT class and arrive exactly as
they do through wifirx.block.wifi_phy_rx."""
import os
import re
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## 2."):md.index("## 3.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, re.S)
    assert len(blocks) == 1
    return blocks[0]


def _load_stub():
    from wifirx import grshim
    src = _stub_source()
    # the only edit: the library by its in-tree path instead of the loader's search path
    assert src.count('C.CDLL("libwifirx.so")') == 1
    src = src.replace('C.CDLL("libwifirx.so")', 'C.CDLL(%r)' % os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd", "wifirx", "libwifirx.so"))
    gnuradio = types.ModuleType("gnuradio")
    gnuradio.gr = grshim.gr_shim
    saved = {k: sys.modules.get(k) for k in ("gnuradio", "pmt")}
    sys.modules["gnuradio"], sys.modules["pmt"] = gnuradio, grshim.pmt_shim
    try:
        ns = {"__name__": "integration_stub"}
        exec(compile(src, "INTEGRATION.md#2", "exec"), ns)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ns


def test_stub_parses_and_binds_every_symbol_it_names():
    from wifirx import capi
    ns = _load_stub()
    cls = ns["wifi_phy_rx"]
    for m in ("work", "stop", "publish", "set_bandwidth", "set_frequency", "set_sensitivity", "set_chan_est", "set_algorithm"):
        assert callable(getattr(cls, m))
    # every wifirx_* name in the text is exported by the library and declared in the header
    names = set(re.findall(r"\bwifirx_[a-z_0-9]+\b", _stub_source())) - {"wifirx_config", "wifirx_poll_out", "wifirx_frame"}
    hdr = open(os.path.join(ROOT, "include", "wifirx.h")).read()
    for n in sorted(names):
        assert hasattr(ns["lib"], n) and n in hdr and n in capi.EXPORTS, n
    # the structures of the text have the layout of capi's (which tests/test_abi.py checks against the header)
    import ctypes as C
    assert C.sizeof(ns["Config"]) == C.sizeof(capi.Config) and C.sizeof(ns["PollOut"]) == C.sizeof(capi.PollOut)
    assert [f[0] for f in ns["Config"]._fields_] == [f[0] for f in capi.Config._fields_]
    assert ns["FRAME"] == capi.FRAME_DTYPE
    assert (ns["P_BANDWIDTH"], ns["P_FREQUENCY"], ns["P_SENSITIVITY"], ns["P_CHAN_EST"], ns["P_STREAM_BATCH"], ns["P_STREAM_IDX"]) == \
           (capi.P_BANDWIDTH, capi.P_FREQUENCY, capi.P_SENSITIVITY, capi.P_CHAN_EST, capi.P_STREAM_BATCH, capi.P_STREAM_IDX)


@pytest.mark.gpu
def test_config1_pieces_through_the_documented_stub():
    """kodim01's first 400 pieces (the reference's wire format) as BPSK-1/2 frames with packet_pad2 gaps through the class
    of INTEGRATION.md: every PDU on `mac_out` carries the piece, `carrier` gets one PDU per data symbol, and both equal
    what wifirx.block.wifi_phy_rx publishes for the same stream."""
    from wifirx import app, block, grshim, txgen
    img = np.load(os.path.join(ROOT, "tests", "golden", "kodim_300.npz"))["kodim01"]
    n = 400
    pieces = app.detach_image_sorted(img)[:n]
    grams = [app.pack_piece(p) for p in pieces]                  # "=L" length + pickle, upload_image_udp.py:29-32
    frames = []
    for k, d in enumerate(grams):
        psdu = np.frombuffer(txgen.mac_frame(d, seq=k), dtype=np.uint8)[None, :]
        tx = txgen.encode_psdus(psdu, 0, seeds=[(k % 127) + 1])
        frames.append(txgen.packet_pad(tx.samples * np.float32(6.0), 100, 1000))
    x = np.concatenate(frames)
    rng = np.random.default_rng(4)
    x = (x + ((rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size)) * np.sqrt(0.5))).astype(np.complex64)

    def run(blk):
        mac, car = [], []
        grshim.msg_connect(blk, "mac_out", grshim.sink_block(mac.append), "in")
        grshim.msg_connect(blk, "carrier", grshim.sink_block(car.append), "in")
        grshim.run_stream(blk, x, chunk=8192)
        return mac, car

    ns = _load_stub()
    mac_s, car_s = run(ns["wifi_phy_rx"](bandwidth=20e6, chan_est=0, frequency=5.89e9, sensitivity=0.56))
    ref = block.wifi_phy_rx(bandwidth=20e6, frequency=5.89e9, publish_carrier=True)
    mac_b, car_b = run(ref)
    ref.close()
    assert len(mac_s) == n and len(mac_b) == n
    for k, ((meta, blob), (meta_b, blob_b)) in enumerate(zip(mac_s, mac_b)):
        body = bytes(np.asarray(blob, dtype=np.uint8))
        assert body == bytes(np.asarray(blob_b, dtype=np.uint8))
        assert body[24:] == grams[k]                              # Extract Pics strips [24:], then [4:] (IRS_AP_epy_block_2.py:34-36)
        for key in ("frame_bytes", "encoding", "snr", "freq", "freq_offset", "dlt"):
            assert meta[key] == pytest.approx(meta_b[key], rel=1e-12), key
    assert len(car_s) == len(car_b) and len(car_s) >= n * 90
    for (d, v), (d_b, v_b) in zip(car_s[::97], car_b[::97]):
        assert np.array_equal(np.asarray(v), np.asarray(v_b))
    # and through the reference's consumer: the pieces redraw the image region they came from
    got, out = [], np.zeros_like(img)
    pics = app.extract_pics(sink=got.append)
    for m in mac_s:
        pics.handle_msg(m)
    for gmsg in got:
        app.redraw_image(app.load_piece(gmsg), out)
    for (y, xx, c), piece in pieces:
        assert np.array_equal(out[y:y + 10, xx:xx + 10, c:c + 1], piece)
