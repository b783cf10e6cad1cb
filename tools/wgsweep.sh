#!/bin/bash
# threshold-kernel time of one workload under a list of environment settings (no counters).  usage: tools/wgsweep.sh <workload> "ENV..." ...
w=$1; shift
for v in "$@"; do
  AB_ARGS=--sync-steps bash tools/ab.sh $w "$v" 2>&1 | grep -v "^$" | tr '\n' ' '; echo
done
