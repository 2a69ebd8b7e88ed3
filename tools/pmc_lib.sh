#!/bin/bash
# usage (GPU box): tools/pmc_lib.sh <tag> <geometry> <lib.so> [more libs]   -- PMC passes of the demod kernel of ablated /
# variant builds (tools/build_variant.sh), driven by tools/ab_inproc.py (one library per process); prints counters per frame.
TAG=$1; GEO=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmclib_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
  B=$(basename $LIB .so)
  i=0
  for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
           "SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${B}_p$i -- python3 $ROOT/tools/ab_inproc.py $ROOT/$LIB --rounds 1 --geometry $GEO $PMC_AB_ARGS > $OUT/${B}_p$i.log 2>&1 || { tail -5 $OUT/${B}_p$i.log; exit 1; }
  done
  python3 - <<PY > $OUT/$B.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/${B}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "demod" in k:
        print("$B", k)
        for c, v in sorted(d.items()):
            print("   %-28s mean %.6g  per frame %.2f (n=%d)" % (c, sum(v)/len(v), sum(v)/len(v)/1e6, len(v)))
PY
  cat $OUT/$B.txt
done
