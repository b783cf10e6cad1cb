#!/bin/bash
# Regenerates the files under profiles/ on the GPU box (outputs land in gpurun_out/prof, copy what is judged).
# usage: tools/profiles.sh <round-tag>
tag=${1:-r01}
out=gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o $tag -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-parity > $out/${tag}_bench_under_rocprof.json 2> $out/stats.log
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o p -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2> $out/fetch.log
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o p -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2> $out/write.log
timeout 400 python3 bench.py > $out/${tag}_bench.json 2> $out/bench.log
python3 - <<PY
import csv, json, collections
out = "$out"; tag = "$tag"
rows = list(csv.DictReader(open(f"{out}/stats/{tag}_kernel_stats.csv")))
thr = [r for r in rows if "k_threshold" in r["Name"]]
nstep = int(thr[0]["Calls"]) if thr else 1
lines = []; tot = 0
for r in rows:
    per = float(r["TotalDurationNs"]) / nstep / 1e3; tot += per
    lines.append("%-90s calls/step %5.2f  avg %8.1f us  per-step %8.1f us" % (r["Name"].replace("nfc::", "").replace("void ", "")[:88], int(r["Calls"]) / nstep, float(r["AverageNs"]) / 1e3, per))
lines.append("total per step %.1f us over %d steps (incl. warm-up and the untimed extra step)" % (tot, nstep))
open(f"{out}/{tag}_kernel_stats_per_step.txt", "w").write("\n".join(lines) + "\n")
def pmc(d, name):
    rs = [r for r in csv.DictReader(open(f"{out}/{d}/p_counter_collection.csv")) if "k_threshold" in r["Kernel_Name"] and r["Counter_Name"] == name]
    byd = collections.defaultdict(float)
    for r in rs: byd[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = sorted(byd.values()); return v[len(v) // 2]
f, w = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
rec = {"workload": "miller", "samples": 100000000, "kernel": "k_threshold<0,4>", "FETCH_SIZE_kb": round(f), "WRITE_SIZE_kb": round(w),
       "correction": "FETCH_SIZE x2 (gfx950: TCC_EA0_RDREQ counted at 64 B per 128-B request, MI355X_MICROARCH.md section HBM); WRITE_SIZE as reported",
       "bytes_per_launch": int((2 * f + w) * 1024),
       "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-parity   (and the same with --pmc WRITE_SIZE); median over the launches"}
json.dump(rec, open(f"{out}/hbm_traffic.json", "w"), indent=1)
print(rec)
print(open(f"{out}/{tag}_kernel_stats_per_step.txt").read())
PY
tail -1 $out/${tag}_bench.json | cut -c1-1500
