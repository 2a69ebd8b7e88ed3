#!/bin/bash
# usage (on the GPU box, through gpurun): tools/collect_round.sh <part> <tag>     part = 1 | 2 | 3
# Runs the measurements the round's profiles/ are made from and leaves them under gpurun_out/collect_<tag>/.
# Before the first call in a fresh container: tools/build_probes.sh (the probes and the preamble-only library are built here, not on the GPU box).
# Do not edit csrc/ while a call is queued: gpurun snapshots the tree when the box is there, and the files are stamped with the hash of what it finds.
# Part 1: bench, rocprofv3 kernel stats, PMC passes, memory floor.   Part 2: other geometries, host path, probes,
# preamble-only build.   Part 3: parity / stream campaigns and the config-3 sweep.
PART=$1; TAG=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/collect_$TAG
mkdir -p $OUT
cd $ROOT
if [ "$PART" = 1 ]; then
  python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  echo "bench done"; python3 -c "import json; j=json.load(open('$OUT/bench.json')); print(j['value']/1e9, j['roofline']['frac'], j['roofline']['kernel_ms'], j['pdu_leg']['decode_mac_ms'], j.get('host_path'))"
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/stats.err ) || { tail -5 $OUT/stats.err; exit 1; }
  echo "rocprof stats done"
  tools/pmc.sh $TAG --pdu-steps 1 --steps 4 --warmup 1 > $OUT/pmc.txt 2>&1 || { tail -5 $OUT/pmc.txt; exit 1; }
  echo "pmc done"
  ./tools/mem_floor.bin > $OUT/mem_floor.txt 2>&1; tail -1 $OUT/mem_floor.txt
  # the two files bench.py quotes, stamped with the hash of the kernel sources they were measured on (tools/csrc_sha.py)
  grep '^{' $OUT/mem_floor.txt | tail -1 > $OUT/mem_floor.json && python3 tools/csrc_sha.py --stamp $OUT/mem_floor.json > /dev/null
  python3 tools/pmc_to_profiles.py gpurun_out/pmc_$TAG $OUT/final > /dev/null && ls $OUT/final_*
fi
if [ "$PART" = 2 ]; then
  python tools/other_configs.py > $OUT/other_geometries.json 2> $OUT/other.err || tail -3 $OUT/other.err
  echo "geometries done"
  python tools/host_path_bench.py 12000 > $OUT/host_path.txt 2>&1; tail -3 $OUT/host_path.txt
  ./tools/mfma_f32_probe.bin > $OUT/mfma_f32_probe.txt 2>&1
  ./tools/mfma_valu_coexec.bin > $OUT/mfma_valu_coexec.txt 2>&1
  ./tools/valu_rate.bin > $OUT/valu_rate.txt 2>&1
  [ -x tools/mem_floor64.bin ] && ./tools/mem_floor64.bin > $OUT/mem_floor64.txt 2>&1
  if [ -f ab/preamble_only.so ]; then
    WIFIRX_LIB=$ROOT/ab/preamble_only.so python bench.py --steps 5 --warmup 2 --no-cpu --pdu-steps 0 > $OUT/preamble_only.json 2>/dev/null
    python3 -c "import json; j=json.load(open('$OUT/preamble_only.json')); print('preamble only ms', j['roofline']['kernel_ms'])"
  fi
  python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -2 $OUT/pytest_gpu.txt
  cp gpurun_out/r02_gpu_configs.json gpurun_out/r03_campaign_*.json $OUT/ 2>/dev/null
fi
if [ "$PART" = 3 ]; then
  # the bulk parity evidence is part of the -m gpu suite since round 3 (tests/test_gpu_campaign.py writes gpurun_out/r03_campaign_*.json);
  # here: the larger runs behind it
  python tests/campaigns/lts_rule6.py 1400 5 0 0.56 > $OUT/lts_rule6_thr56.json 2> $OUT/c1.err; python3 -c "import json; print(json.load(open('$OUT/lts_rule6_thr56.json'))['totals'])"
  python tests/campaigns/lts_rule6.py 1400 6 0 0.35 > $OUT/lts_rule6_thr35.json 2> $OUT/c2.err; python3 -c "import json; print(json.load(open('$OUT/lts_rule6_thr35.json'))['totals'])"
  python tests/campaigns/parity_campaign.py 60000 31 > $OUT/parity_campaign_seed31.json 2> $OUT/c3.err; tail -c 200 $OUT/parity_campaign_seed31.json; echo
  python tests/campaigns/parity_campaign.py 100000 93 plain > $OUT/parity_campaign_seed93_plain_outputs.json 2> $OUT/c4.err; tail -c 200 $OUT/parity_campaign_seed93_plain_outputs.json; echo
  python tests/campaigns/stream_campaign.py 400 40 2 > $OUT/stream_campaign.json 2> $OUT/c5.err; tail -c 300 $OUT/stream_campaign.json; echo
  python tests/campaigns/ber_sweep.py 100000 0 > $OUT/config3_ber_sweep.json 2> $OUT/c6.err; tail -c 300 $OUT/config3_ber_sweep.json; echo
fi
