#!/bin/bash
# usage: tools/ktrace.sh <tag> [bench args]  -- rocprofv3 --kernel-trace of bench.py, prints mean duration per kernel
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ktrace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if "wr::" in k: print("%-50s n=%3d mean %.3f ms min %.3f" % (k, len(v), sum(v) / len(v), min(v)))
PY
