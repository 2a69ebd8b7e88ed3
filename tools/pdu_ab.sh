#!/bin/bash
# alternating bench runs: prints kernel_ms, demod_with_planes_ms, demod_planes_only_ms, decode_mac_ms
for i in 1 2 3; do for v in "$@"; do
  WIFIRX_LIB=$(realpath $v) python bench.py --steps 6 --warmup 2 --no-cpu --pdu-steps 5 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['pdu_leg']; print('$v', round(d['roofline']['kernel_ms'],3), round(p['demod_with_planes_ms'],3), round(p['demod_planes_only_ms'],3), round(p['decode_mac_ms'],3))"
done; done
