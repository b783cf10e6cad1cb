#!/bin/bash
# One rocprofv3 counter pass over a bench run; prints the per-dispatch averages of the kernels whose name matches $PMC_MATCH
# (default: threshold).  usage: tools/pmc.sh <outdir> <counters...> -- bench args
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d gpurun_out/$out -o p -- python3 bench.py "$@" > gpurun_out/$out.log 2>&1
python3 - <<PY
import csv,collections,os
match=os.environ.get("PMC_MATCH","threshold").split(",")
rows=list(csv.DictReader(open("gpurun_out/$out/p_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in rows:
    k=r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
seen=set()
for r in rows:
    key=(r["Kernel_Name"][:60], r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); n[r["Kernel_Name"][:60]]+=1
for k,v in agg.items():
    if any(m in k for m in match):
        print(k, "dispatches", n[k], {a:round(b/n[k]) for a,b in v.items()})
PY
