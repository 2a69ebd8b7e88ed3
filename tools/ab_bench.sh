#!/bin/bash
# Same-box A/B of two builds of libwifirx.so (the boxes of the pool differ by +-3 %, so only alternating runs on one
# box separate changes of a per cent):
#   cp gnuradio-wifi-imagetransfer_amd/wifirx/libwifirx.so /somewhere/a.so     # build A, then build B likewise
#   gpurun -- 'tools/ab_bench.sh path/a.so path/b.so [rounds] [extra bench.py args]'
# Prints ms_per_step and the HIP-event kernel time of bench.py (demod only: --no-cpu --pdu-steps 0) for each run;
# add "--pdu-steps 5" to compare decode_mac instead (pdu_leg.decode_mac_ms is printed when present).
A=$1; B=$2; R=${3:-3}; shift; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq $R); do
  for v in "$A" "$B"; do
    WIFIRX_LIB=$(realpath "$v") python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu --pdu-steps 0 "$@" 2>/dev/null | tail -1 |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('pdu_leg') or {}; print('$v', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), p.get('decode_mac_ms'))"
  done
done
