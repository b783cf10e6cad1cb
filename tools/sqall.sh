#!/bin/bash
# SQ counters per wave of EVERY kernel of a step (one --pmc pass).  usage: tools/sqall.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out=gpurun_out/sqall_$tag
timeout 600 rocprofv3 --pmc ${SQALL_PMC:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU} --kernel-trace --output-format csv -d $out -o p -- python3 bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-extras --sync-steps > /dev/null 2> $out.log
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$out/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
names = []
for r in rows:
    k = r["Kernel_Name"].replace("void nfc::", "").replace("(anonymous namespace)::", "").split("(")[0][:34]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
    if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
print("per wave:", " ".join(n.replace("SQ_", "") for n in names if n != "SQ_WAVES"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = len(nd[k]); wv = max(v["SQ_WAVES"] / n, 1)
    print("%-34s disp %3d waves %7d |" % (k, n, wv), " ".join("%9.0f" % (v[c] / n / wv) for c in names if c != "SQ_WAVES"))
PY
