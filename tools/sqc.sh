#!/bin/bash
# SQ instruction counters per wave of the threshold kernel for one environment setting.  usage: tools/sqc.sh <tag> "VAR=val ..." [workload]
tag=$1; envs=$2; w=${3:-miller}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out=gpurun_out/sqc_$tag
export $envs
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out -o p -- python3 bench.py --workload $w ${SQC_ARGS:-} --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-extras --sync-steps > /dev/null 2> $out.log
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$out/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].replace("void nfc::", "").split("(")[0][:50]
    if "threshold" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = len(nd[k]); wv = max(v["SQ_WAVES"] / n, 1)
    print("$tag %-40s waves %6d valu %6.0f salu %6.0f lds %5.0f | per wave, quad-cycles: life %7.0f wait_any %7.0f wait_inst %7.0f active %7.0f" % (
        k, wv, v["SQ_INSTS_VALU"]/n/wv, v["SQ_INSTS_SALU"]/n/wv, v["SQ_INSTS_LDS"]/n/wv, v["SQ_WAVE_CYCLES"]/n/wv, v["SQ_WAIT_ANY"]/n/wv, v["SQ_WAIT_INST_ANY"]/n/wv, v["SQ_ACTIVE_INST_ANY"]/n/wv))
PY
