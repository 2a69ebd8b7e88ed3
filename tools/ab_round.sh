#!/bin/bash
# usage (GPU box): tools/ab_round.sh <old.so> <new.so> <outdir>  -- the round's net effect, one process per line of the table:
# tools/ab_inproc.py on config 2 (LS, with planes, planes only, carrier on, LMS, COMB, STA) and the other BASELINE geometries
OLD=$1; NEW=$2; OUT=$3
mkdir -p $OUT
run() { python tools/ab_inproc.py $OLD $NEW --rounds 5 "$@" > $OUT/ab_$TAG.txt 2>&1; echo "$TAG: $(tail -2 $OUT/ab_$TAG.txt | awk '{print $1, $5}' | tr '\n' ' ')"; }
TAG=ls run
TAG=planes run --planes
TAG=planes_only run --planes-only
TAG=carrier run --carrier
TAG=lms run --eq 1
TAG=comb run --eq 2
TAG=sta run --eq 3
TAG=g1 run --geometry 1
TAG=g3_awgn run --geometry 3
TAG=g3_sv run --geometry 3 --sv --snr 20
TAG=g3_sv_comb run --geometry 3 --sv --snr 20 --eq 2
TAG=g3_sv_sta run --geometry 3 --sv --snr 20 --eq 3
