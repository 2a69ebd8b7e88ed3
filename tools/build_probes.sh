#!/bin/bash
# Builds what tools/collect_round.sh expects to find next to the library (run HERE, in the build container: hipcc cross-compiles
# gfx950; the binaries travel to the GPU box with the snapshot, they are git-ignored):
#   tools/*.bin            the measurement probes (memory floors, copy / burst probes, MFMA and issue-rate probes)
#   tools/libboxprobe.so   bench.py's same-run yardstick (also built by __graft_entry__.build())
#   ab/preamble_only.so    the library with -DWR_ABLATE=1 (preamble phase only) for part 2 of the collection
# Exits non-zero when any compile fails or an expected binary is missing (errors used to be swallowed).
# usage: tools/build_probes.sh
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/tools" || exit 1
fail=0
pids=()
names=()
for f in mem_floor mem_floor64 mem_burst copy_probe mfma_f32_probe mfma_valu_coexec valu_rate valu_cost calib_fetch; do
  [ -f $f.hip ] || continue
  rm -f $f.bin
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $f.hip -o $f.bin > /tmp/build_probe_$f.log 2>&1 &
  pids+=($!)
  names+=($f)
done
for i in "${!pids[@]}"; do
  if ! wait "${pids[$i]}" || [ ! -x "${names[$i]}.bin" ]; then
    echo "build_probes: ${names[$i]} FAILED"; tail -5 /tmp/build_probe_${names[$i]}.log; fail=1
  fi
done
( cd "$ROOT" && python3 -c "import __graft_entry__ as g; g.build_box_probe()" ) || { echo "build_probes: libboxprobe.so FAILED"; fail=1; }
mkdir -p "$ROOT/ab"
"$ROOT/tools/build_variant.sh" "$ROOT/ab/preamble_only.so" -DWR_ABLATE=1 > /tmp/build_probe_preamble_only.log 2>&1 || { echo "build_probes: preamble_only.so FAILED"; tail -5 /tmp/build_probe_preamble_only.log; fail=1; }
ls "$ROOT"/tools/*.bin "$ROOT"/tools/libboxprobe.so "$ROOT"/ab/preamble_only.so
exit $fail
