#!/bin/bash
# usage: tools/pmc_mix.sh <tag> [bench args]  -- the dynamic vector-instruction mix of the kernels (two --pmc passes)
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcmix_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64" \
         "SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "demod" in k or "decode" in k:
        print(k)
        for c, v in sorted(d.items()):
            print("   %-28s mean %.6g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
