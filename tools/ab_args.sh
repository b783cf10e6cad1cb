#!/bin/bash
# A/B of bench arguments: usage: tools/ab_args.sh <workload> "args" "args" ...
w=$1; shift
for v in "$@"; do
  echo "== $v"
  python3 bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --no-parity $v 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('ms_per_step %.4f  k_thr avg %.4f ms  frac %.3f  launches/step %s  chunks %d x %d' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['launches_per_step'], d['config']['time_chunks'], d['config']['time_chunk_samples']))
    else:
        print(l)
"
done
