#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of tools/pmc.sh into the two files bench.py and DESIGN.md quote.

usage: tools/pmc_to_profiles.py gpurun_out/pmc_<tag> profiles/r01_final [frames_per_launch]

Writes <prefix>_pmc_summary.csv (mean per dispatch of every counter, per kernel) and <prefix>_traffic.json
(HBM bytes per launch of the demod kernel: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for gfx950 -- the factor was re-measured with tools/calib_fetch.hip).
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_sha import csrc_sha  # noqa: E402


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(src + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            # template arguments stay in the name: demod_batch_kernel<EQ, planes> and decode_kernel<mixed> are different kernels
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Counter_Name"]].append(
                float(r["Counter_Value"]))
    with open(prefix + "_pmc_summary.csv", "w") as out:
        out.write("kernel,counter,mean_per_dispatch,dispatches\n")
        for k in sorted(acc):
            if not ("demod" in k or "decode" in k):
                continue
            for c in sorted(acc[k]):
                v = acc[k][c]
                out.write("\"%s\",%s,%.6g,%d\n" % (k, c, sum(v) / len(v), len(v)))
    dom = "wr::demod_batch_kernel<0, false, false>"            # the timed kernel of bench.py: LS equaliser, no plane output
    d = {c: sum(v) / len(v) for c, v in acc[dom].items()}
    corr = 2.0
    rd, wr = d["FETCH_SIZE"] * 1024 * corr, d["WRITE_SIZE"] * 1024
    traffic = {
        "csrc_sha": csrc_sha(),       # the kernels these counters were taken from (bench.py: traffic_stale when the tree differs)
        "kernel": dom,
        "frames_per_launch": frames,
        "FETCH_SIZE_KB": d["FETCH_SIZE"],
        "WRITE_SIZE_KB": d["WRITE_SIZE"],
        "fetch_correction": corr,
        "fetch_correction_source": "tools/calib_fetch.hip under rocprofv3 --pmc FETCH_SIZE: 5.4272e9 B loaded by lanes in "
                                   "the kernel's own access pattern (rows of 16 lanes x 8 B), FETCH_SIZE = 2650197 KB -> "
                                   "factor 2.000; MI355X_MICROARCH.md HBM section",
        "hbm_read_bytes_per_launch": rd,
        "hbm_write_bytes_per_launch": wr,
        "hbm_bytes_per_launch": rd + wr,
        "hbm_bytes_per_frame": (rd + wr) / frames,
        "algorithmic_bytes_per_frame": 58496,
        "valu_insts_per_frame": d["SQ_INSTS_VALU"] / frames,
        "salu_insts_per_frame": d["SQ_INSTS_SALU"] / frames,
        "lds_insts_per_frame": d["SQ_INSTS_LDS"] / frames,
        "lds_bank_conflict_cycles": d.get("SQ_LDS_BANK_CONFLICT", 0.0),
    }
    json.dump(traffic, open(prefix + "_traffic.json", "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
