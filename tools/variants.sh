#!/bin/bash
# A/B of builds of the library (usrp_nfc_amd/build.py --variant): per build a parity check against the oracle on a short capture, then the
# per-kernel time per step of one workload under rocprofv3.  usage: tools/variants.sh <workload> lib1.so lib2.so ...
w=$1; shift
for lib in "$@"; do
  tag=$(basename $lib .so)
  echo "== $lib"
  NFC_AMD_LIB=$lib python3 bench.py --workload $w --samples 2e7 --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['parity']
        print('parity:', p['edges_equal'], p['symbols_equal'], p['packets_equal'], p['n_edges'])
    elif 'rror' in l: print(l.strip())
"
  NFC_AMD_LIB=$lib tools/kstats.sh $w $tag 2>&1 | grep -v "rocclr\|set_state\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
done
