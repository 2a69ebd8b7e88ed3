#!/usr/bin/env python3
"""Make the fixtures under tests/golden/ from the reference tree (runs ONLY in the build container,
where /root/reference exists; the fixtures are committed, the reference never travels).

  grc_constants.json   the PHY constants the reference itself carries in
                       gnu_radio/wifi_phy_hier.grc:336-424,459-479,698-734 and the block arguments of
                       gnu_radio/IRS_AP.py:268-285 (evaluated from the YAML, no GNU Radio needed)
  sv_taps.npy          1024 x 8 complex64 channel taps built from draws of
                       utils/SV_channel.py (Saleh_Valenzuela_Channel.RicianRefresh), SURVEY.md 8(d) config 3
  kodim_300.npz        the six Kodak images of images/ resized to 300x300 as upload_image_udp.py:21 does
  pieces_kodim01.npz   the datagrams upload_image_udp.py:29-32 would send for the first 64 pieces
                       (sorted by (y,x,c)) of kodim01, made with image_detach_rebuild.detach_image
"""
import json
import os
import pickle
import struct
import sys

import numpy as np
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")


def grc_constants():
    with open(os.path.join(REF, "gnu_radio", "wifi_phy_hier.grc")) as f:
        doc = yaml.safe_load(f)
    blocks = {b["name"]: b for b in doc["blocks"]}
    alloc = blocks["digital_ofdm_carrier_allocator_cvc_0_0_0"]["parameters"]
    env = {"list": list, "range": range}
    occ = eval(alloc["occupied_carriers"], {"__builtins__": {}}, env)
    pil = eval(alloc["pilot_carriers"], {"__builtins__": {}}, env)
    psy = eval(alloc["pilot_symbols"], {"__builtins__": {}}, env)
    syn = eval(alloc["sync_words"], {"__builtins__": {}}, env)
    out = {
        "source": "gnu_radio/wifi_phy_hier.grc",
        "fft_len": int(alloc["fft_len"]),
        "output_is_shifted": alloc["output_is_shifted"],
        "occupied_carriers": [list(map(int, o)) for o in occ],
        "pilot_carriers": [list(map(int, p)) for p in pil],
        "pilot_symbols": [list(map(int, p)) for p in psy],
        "sync_words": [[[float(complex(v).real), float(complex(v).imag)] for v in w] for w in syn],
        "cp_len": int(blocks["digital_ofdm_cyclic_prefixer_0_0"]["parameters"]["cp_len"]),
        "rolloff": int(blocks["digital_ofdm_cyclic_prefixer_0_0"]["parameters"]["rolloff"]),
        "ifft_window": blocks["fft_vxx_0_0"]["parameters"]["window"],
        "ifft_shift": blocks["fft_vxx_0_0"]["parameters"]["shift"],
        "rx_fft_shift": blocks["fft_vxx_0_1"]["parameters"]["shift"],
        "rx_fft_forward": blocks["fft_vxx_0_1"]["parameters"]["forward"],
        "sync_length": int(blocks["sync_length"]["parameters"]["value"]),
        "window_size": int(blocks["window_size"]["parameters"]["value"]),
        "max_symbols_expr": blocks["max_symbols"]["parameters"]["value"],
        "sync_short": {k: blocks["sync_short"]["parameters"][k] for k in ("min_plateau", "threshold")},
        "sync_long": {"sync_length": blocks["sync_long"]["parameters"]["sync_length"]},
        "moving_average_cc_length": blocks["blocks_moving_average_xx_0"]["parameters"]["length"],
        "moving_average_ff_length": blocks["blocks_moving_average_xx_1"]["parameters"]["length"],
        "delay_autocorr": blocks["blocks_delay_0_0"]["parameters"]["delay"],
        "delay_sync_long": blocks["blocks_delay_0"]["parameters"]["delay"],
        "pads": sorted(b["name"] for b in doc["blocks"] if b["id"].startswith("pad_")),
        "pad_labels": sorted(b["parameters"]["label"] for b in doc["blocks"] if b["id"].startswith("pad_")),
        "parameters": {b["name"]: b["parameters"]["value"] for b in doc["blocks"] if b["id"] == "parameter"},
    }
    return out


def sv_taps(n=1024, L=8):
    sys.path.insert(0, REF)
    from utils.SV_channel import Saleh_Valenzuela_Channel     # module seeds numpy with 0 at import
    np.random.seed(0)
    scale, interval = 16, 0.03
    irs_pos = np.array([interval / 2, interval / 2, 0])
    ap_pos = np.array([scale * interval / 2, scale * interval / 2, 4.5])
    usr = np.array([[scale * interval / 2 + 0.3, scale * interval / 2 - 0.2, 1.0]])
    ch = Saleh_Valenzuela_Channel(scale, irs_pos, ap_pos, 1, 1)
    ch.genLoS(usr)
    psi = np.exp(1j * np.pi * np.ones((scale ** 2,)))
    pdp = np.exp(-np.arange(L) / 2.0)
    pdp /= pdp.sum()
    taps = np.zeros((n, L), dtype=np.complex128)
    for i in range(n):
        for l in range(L):
            b2r, r2u, b2u = ch.RicianRefresh(K=10)
            g = (np.linalg.multi_dot([b2r, np.diag(psi), r2u]) + b2u)[0, 0]
            taps[i, l] = np.sqrt(pdp[l]) * g
    # normalise the average channel energy to 1 (per-draw fading kept)
    taps /= np.sqrt(np.mean(np.sum(np.abs(taps) ** 2, axis=1)))
    return taps.astype(np.complex64)


def images():
    from PIL import Image
    out = {}
    for name in sorted(os.listdir(os.path.join(REF, "images"))):
        if name.endswith(".png"):
            out[name[:-4]] = np.array(Image.open(os.path.join(REF, "images", name)).resize((300, 300)), dtype=np.uint8)
    return out


def pieces(img, n=64):
    sys.path.insert(0, REF)
    import image_detach_rebuild as idr
    pcs = sorted(idr.detach_image(img), key=lambda p: p[0])      # the reference shuffles unseeded: sort
    assert np.array_equal(idr.rebuild_image(pcs, img.shape), img)
    grams = []
    for p in pcs[:n]:
        data = pickle.dumps(p)
        grams.append(struct.pack("=L", len(data)) + data)
    return len(pcs), grams


def main():
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "grc_constants.json"), "w") as f:
        json.dump(grc_constants(), f, indent=1)
    np.save(os.path.join(OUT, "sv_taps.npy"), sv_taps())
    imgs = images()
    np.savez_compressed(os.path.join(OUT, "kodim_300.npz"), **imgs)
    n_total, grams = pieces(imgs["kodim01"])
    lens = np.array([len(g) for g in grams], dtype=np.int32)
    blob = np.zeros((len(grams), lens.max()), dtype=np.uint8)
    for i, g in enumerate(grams):
        blob[i, :len(g)] = np.frombuffer(g, dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "pieces_kodim01.npz"), datagrams=blob, lengths=lens,
                        n_pieces_total=np.int32(n_total))
    print("fixtures written to", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
