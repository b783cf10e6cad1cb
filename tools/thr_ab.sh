#!/bin/bash
# The threshold kernel's launch time (HIP events of bench.py's timed loop) for several builds of the library, alternating, N rounds.
# usage: tools/thr_ab.sh <workload> <rounds> lib1.so lib2.so ...
w=$1; n=$2; shift 2
for r in $(seq $n); do
  for lib in "$@"; do
    NFC_AMD_LIB=$lib python3 bench.py --workload $w --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-parity ${AB_ARGS:-} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('%-32s launch %.4f ms  step %.4f ms' % ('$lib'.split('/')[-1], d['roofline']['avg_launch_ms'], d['ms_per_step']))
"
  done
done
