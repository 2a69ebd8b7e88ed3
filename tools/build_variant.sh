#!/bin/bash
# usage: tools/build_variant.sh <out.so> [extra -D flags]   -- builds the working tree's csrc in a scratch copy
# (for same-box A/B runs with tools/ab_bench.sh; e.g. -DWR_ABLATE=1 = preamble phase only)
set -e
OUT=$(realpath -m "$1"); shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/gnuradio-wifi-imagetransfer_amd/wifirx
cp -r $ROOT/gnuradio-wifi-imagetransfer_amd/csrc $T/gnuradio-wifi-imagetransfer_amd/
cp -r $ROOT/include $T/
cd $T/gnuradio-wifi-imagetransfer_amd/csrc
make clean -s
make -j4 -s FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -I../../include -I. -Wall -Wno-unused-function -Wno-inline-asm $*"
cp ../wifirx/libwifirx.so "$OUT"
rm -rf $T
