#!/bin/bash
# usage: tools/isa.sh <file.hip> <out.s> [extra hipcc flags]  -- the gfx950 assembly of one translation unit (same flags as csrc/Makefile)
cd "$(dirname "$0")/../gnuradio-wifi-imagetransfer_amd/csrc" || exit 1
F=$1; O=$2; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC \
  -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -I../../include -I. --cuda-device-only -S "$F" -o "$O" "$@"
