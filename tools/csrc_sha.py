#!/usr/bin/env python3
"""One hash over the kernel sources (csrc/*.hip, *.h, *.inc and the generated tables they include): profile files that
quote a figure of the kernels carry it (`csrc_sha`), and bench.py refuses to quote a figure whose hash is not the
tree's -- a PMC traffic count or a memory floor measured on other kernels goes stale silently otherwise.

    python tools/csrc_sha.py                 # prints the hash
    python tools/csrc_sha.py --stamp f.json  # adds / refreshes "csrc_sha" in a JSON file made from THIS tree's kernels
"""
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha() -> str:
    h = hashlib.sha256()
    base = os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd", "csrc")
    files = sorted(f for pat in ("*.hip", "*.h", "*.inc") for f in glob.glob(os.path.join(base, pat)))
    files.append(os.path.join(ROOT, "include", "wifirx_tables.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--stamp":
        for f in sys.argv[2:]:
            d = json.load(open(f))
            d["csrc_sha"] = csrc_sha()
            json.dump(d, open(f, "w"), indent=1)
    print(csrc_sha())
