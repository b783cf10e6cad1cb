#!/bin/bash
# Per-kernel time per step (rocprofv3 kernel stats of bench.py's timed loop) for several builds of the library, alternating.
# usage: tools/step_ab.sh <workload> <rounds> lib1.so lib2.so ...      (STEP_AB_ARGS: extra bench arguments)
w=$1; n=$2; shift 2
for r in $(seq $n); do
  for lib in "$@"; do
    echo "== $lib"
    NFC_AMD_LIB=$lib KSTATS_ARGS="${STEP_AB_ARGS:-}" bash tools/kstats.sh $w ab 2>&1 | grep -v "rocclr\|set_state\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | awk '{printf "%-44s %s %s\n", substr($0,1,44), $(NF-1), $NF}'
  done
done
