#!/bin/bash
# Same-box A/B of library builds on ONE geometry of tools/other_configs.py (0 = config-3 geometry, 1 = config 1, 3 = config 2):
#   gpurun -- 'tools/geo_ab.sh <case> <rounds> a.so b.so [c.so ...]'
CASE=$1; R=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq $R); do
  for v in "$@"; do
    WIFIRX_LIB=$(realpath "$v") WIFIRX_ONLY_CASE=$CASE python "$ROOT/tools/other_configs.py" 2>/dev/null | tail -1 |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['cases'][0]; print('$v', round(c['demod_ms'],3), round(c['demod_with_planes_ms'],3))"
  done
done
