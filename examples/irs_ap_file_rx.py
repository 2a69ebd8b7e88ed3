#!/usr/bin/env python3
"""The receive side of the reference's IRS_AP flowgraph (gnu_radio/IRS_AP.py) with the 15 RX blocks replaced by
`wifi_phy_rx`, a file source instead of the USRP and no Qt: I/Q file -> wifi_phy_rx (MI355X) -> Extract Pics
-> rebuilt image.  With GNU Radio installed the same block sits in a gr.top_block; here it is
driven by the scheduler stand-in of wifirx.grshim.

    python examples/irs_ap_file_rx.py in.c64 out.png [--bandwidth 20e6] [--chan-est 0]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
from wifirx import app, block, grshim  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("iq")
    ap.add_argument("out")
    ap.add_argument("--bandwidth", type=float, default=20e6)
    ap.add_argument("--frequency", type=float, default=5.89e9)
    ap.add_argument("--chan-est", type=int, default=block.LS)
    ap.add_argument("--chunk", type=int, default=1 << 20)
    a = ap.parse_args()
    x = np.fromfile(a.iq, dtype=np.complex64)
    img = np.zeros((300, 300, 3), np.uint8)
    n = [0]

    def on_piece(data):
        app.redraw_image(app.load_piece(data), img)
        n[0] += 1

    rx = block.wifi_phy_rx(bandwidth=a.bandwidth, frequency=a.frequency, chan_est=a.chan_est, publish_carrier=False)
    pics = app.extract_pics(sink=on_piece)
    grshim.msg_connect(rx, "mac_out", pics, "MAC")
    grshim.run_stream(rx, x, chunk=a.chunk)
    st = rx.stats()
    print("samples %d, frames detected %d, PDUs %d, pieces drawn %d" % (st["samples_in"], st["frames_detected"], st["frames_crc_ok"], n[0]))
    from PIL import Image
    Image.fromarray(img).save(a.out)


if __name__ == "__main__":
    main()
