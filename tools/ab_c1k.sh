#!/bin/bash
# A/B of environment settings on the 10 Msps configuration: usage: tools/ab_c1k.sh "VAR=val ..." ...
for v in "$@"; do
  echo "== $v"
  env $v python3 bench.py --workload classic1k --steps 6 --warmup 2 --no-cpu-baseline --no-extras ${C1K_ARGS:-} 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('ms/step %.3f  k_thr %.3f ms  frac %.3f  launches/step %s  chunks %d x %d  parity %s' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['launches_per_step'], d['config']['time_chunks'], d['config']['time_chunk_samples'], d.get('parity', {}).get('edges_equal')))
    else:
        print(l.rstrip()[:200])
"
done
