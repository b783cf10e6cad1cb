#!/bin/bash
# usage: tools/prof.sh <outdir> bench args...   -> per-kernel per-step table
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out -o p -- python3 bench.py "$@" > gpurun_out/$out.log 2>&1
python3 - <<PY
import csv,sys
rows=list(csv.DictReader(open("gpurun_out/$out/p_kernel_stats.csv")))
thr=[r for r in rows if "k_threshold" in r["Name"]]
nstep=int(thr[0]["Calls"]) if thr else 1
tot=0
for r in rows:
    short=r["Name"].replace("nfc::","").replace("void ","")[:74]
    per=float(r["TotalDurationNs"])/nstep/1e3; tot+=per
    if per>3: print("%-76s calls/step %5.1f avg %8.1f us per-step %8.1f us"%(short,int(r["Calls"])/nstep,float(r["AverageNs"])/1e3,per))
print("total per step us %.1f (steps incl warmup %d)"%(tot,nstep))
PY
