#!/bin/bash
# usage (on the GPU box, through gpurun): tools/collect_round.sh <part> <tag>     part = 1 | 2
# ONE collection per round: runs the measurements profiles/ is made from and leaves them under
# gpurun_out/collect_<tag>/.  Before the call, here: tools/build_probes.sh (ab/preamble_only.so, tools/libboxprobe.so).
# Do not edit csrc/ while a call is queued: gpurun snapshots the tree when the box is there, and the files are stamped with
# the hash of what it finds.
# Part 1: bench.py (the JSON line: headline, roofline.box yardstick, variants, BER), the same under rocprofv3 --kernel-trace
#         --stats, the PMC passes (separate runs), the preamble-only build.
# Part 2: the bulk parity campaigns behind the -m gpu suite's reduced ones, the config-3 BER sweep, the host path.
PART=$1; TAG=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/collect_$TAG
mkdir -p $OUT
cd $ROOT
if [ "$PART" = 1 ]; then
  python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  echo "bench done"; python3 -c "import json; j=json.load(open('$OUT/bench.json')); r=j['roofline']; print(j['value']/1e9, r['frac'], r['kernel_ms'], r.get('kernel_vs_box_floor'), j['pdu_leg']['decode_mac_ms'])"
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/stats.err ) || { tail -5 $OUT/stats.err; exit 1; }
  echo "rocprof stats done"
  tools/pmc.sh $TAG --pdu-steps 1 --steps 4 --warmup 1 --pipeline-batches 0 > $OUT/pmc.txt 2>&1 || { tail -5 $OUT/pmc.txt; exit 1; }
  echo "pmc done"
  python3 tools/pmc_to_profiles.py gpurun_out/pmc_$TAG $OUT/final > /dev/null && ls $OUT/final_*
  if [ -f ab/preamble_only.so ]; then
    WIFIRX_LIB=$ROOT/ab/preamble_only.so python bench.py --steps 5 --warmup 2 --no-cpu --pdu-steps 0 > $OUT/preamble_only.json 2>/dev/null
    python3 -c "import json; j=json.load(open('$OUT/preamble_only.json')); print('preamble only ms', j['roofline']['kernel_ms'])"
  fi
fi
if [ "$PART" = 2 ]; then
  python tests/campaigns/parity_campaign.py 60000 31 > $OUT/parity_campaign_seed31.json 2> $OUT/c3.err; tail -c 200 $OUT/parity_campaign_seed31.json; echo
  python tests/campaigns/parity_campaign.py 100000 93 plain > $OUT/parity_campaign_seed93_plain_outputs.json 2> $OUT/c4.err; tail -c 200 $OUT/parity_campaign_seed93_plain_outputs.json; echo
  python tests/campaigns/stream_campaign.py 400 40 2 > $OUT/stream_campaign.json 2> $OUT/c5.err; tail -c 300 $OUT/stream_campaign.json; echo
  python tests/campaigns/ber_sweep.py 100000 0 > $OUT/config3_ber_sweep.json 2> $OUT/c6.err; tail -c 300 $OUT/config3_ber_sweep.json; echo
  python tools/host_path_bench.py 12000 > $OUT/host_path.txt 2>&1; tail -3 $OUT/host_path.txt
fi
