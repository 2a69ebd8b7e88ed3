#!/bin/bash
# Builds what tools/collect_round.sh expects to find next to the library (run HERE, in the build container: hipcc cross-compiles
# gfx950; the binaries travel to the GPU box with the snapshot, they are git-ignored):
#   tools/*.bin            the measurement probes (memory floors, copy / burst probes, MFMA and issue-rate probes)
#   ab/preamble_only.so    the library with -DWR_ABLATE=1 (preamble phase only) for part 2 of the collection
# usage: tools/build_probes.sh
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/tools" || exit 1
for f in mem_floor mem_floor64 mem_burst copy_probe mfma_f32_probe mfma_valu_coexec valu_rate valu_cost calib_fetch; do
  [ -f $f.hip ] || continue
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $f.hip -o $f.bin 2>&1 | grep -i "error" ) &
done
wait
mkdir -p "$ROOT/ab"
"$ROOT/tools/build_variant.sh" "$ROOT/ab/preamble_only.so" -DWR_ABLATE=1 > /dev/null 2>&1 || echo "preamble_only.so failed"
ls "$ROOT"/tools/*.bin "$ROOT"/ab/preamble_only.so
