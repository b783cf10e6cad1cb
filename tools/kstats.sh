#!/bin/bash
# Per-kernel time per step of one bench workload under rocprofv3 (kernel trace + stats).  usage: tools/kstats.sh <workload> [tag]
w=${1:-miller}; tag=${2:-k}
out=gpurun_out/kstats_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras ${KSTATS_ARGS:-} > $out/bench.json 2> $out/log.txt
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/${tag}_kernel_stats.csv")))
thr = [r for r in rows if "k_threshold" in r["Name"]]
nstep = int(thr[0]["Calls"]) if thr else 1
tot = 0
for r in rows:
    per = float(r["TotalDurationNs"]) / nstep / 1e3; tot += per
    print("%-70s calls/step %5.2f avg %8.1f us per-step %8.1f us" % (r["Name"].replace("nfc::", "").replace("void ", "")[:68], int(r["Calls"]) / nstep, float(r["AverageNs"]) / 1e3, per))
print("total per step %.1f us" % tot)
PY
