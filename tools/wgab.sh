#!/bin/bash
# quick A/B of the threshold kernel under several environment settings, with SQ counters for each.  usage: tools/wgab.sh <workload> "ENV..." ...
w=$1; shift
i=0
for v in "$@"; do
  i=$((i+1))
  AB_ARGS=--sync-steps bash tools/ab.sh $w "$v" 2>&1 | grep -v "^$"
  bash tools/sqc.sh ab$i "$v" $w
done
