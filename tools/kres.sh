#!/bin/bash
# usage: tools/kres.sh <file.hip> [extra hipcc flags]  -- one line per kernel: registers, spills, scratch, LDS, occupancy
# (hipcc -Rpass-analysis=kernel-resource-usage; same flags as csrc/Makefile)
cd "$(dirname "$0")/../gnuradio-wifi-imagetransfer_amd/csrc" || exit 1
F=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC \
  -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -I../../include -I. -c "$F" -o /tmp/kres.o \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys
cur = None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: print(cur)
        name = t.split(":", 1)[1].strip()
        cur = name[:60].ljust(62)
    else:
        k, v = t.split(":", 1)
        k = k.strip()
        if k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"):
            cur += " %s=%s" % (k.split(" ")[0], v.strip())
if cur: print(cur)
'
