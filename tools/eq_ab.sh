#!/bin/bash
# Same-box A/B of two library builds on the equaliser instances (LS / LMS / COMB / STA, config 2 and config-3 geometry):
#   gpurun -- 'tools/eq_ab.sh ab/a.so ab/b.so [rounds]'
A=$1; B=$2; R=${3:-2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq $R); do
  for v in "$A" "$B"; do
    WIFIRX_LIB=$(realpath "$v") WIFIRX_ONLY_EQ=1 python "$ROOT/tools/other_configs.py" 2>/dev/null | tail -1 |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', ' '.join('%s %.2f' % (e['chan_est'], e['demod_ms']) for e in d['equalisers']))"
  done
done
