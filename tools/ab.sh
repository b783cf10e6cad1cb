#!/bin/bash
# A/B of environment settings on one bench workload: usage: tools/ab.sh <workload> "VAR=val VAR2=val" "..." ...
w=$1; shift
mkdir -p gpurun_out/ab
for v in "$@"; do
  echo "== $v"
  env $v python3 bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras ${AB_ARGS:---primary ahead} 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('ms_per_step %.4f  k_thr avg %.4f ms  frac %.3f  launches/step %s  chunks %d' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['launches_per_step'], d['config']['time_chunks']))
    else:
        print(l)
"
done
