#!/bin/bash
# Per-kernel time per step of one workload for several builds of the library: usage: tools/tail.sh <workload> lib1.so lib2.so ...
w=$1; shift
for lib in "$@"; do
  tag=$(basename $lib .so)
  echo "== $lib"
  NFC_AMD_LIB=$lib tools/kstats.sh $w $tag 2>&1 | grep -v "rocclr\|set_state"
done
