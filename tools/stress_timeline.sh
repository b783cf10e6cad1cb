#!/bin/bash
# Kernel timeline of ONE adapted step of a stress capture (bench.py: stress_config's captures), with the gaps the host's turns leave.
# usage: tools/stress_timeline.sh stress_hover|stress_dropouts_steps|stress_dropouts
nm=$1
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_$nm -o p -- python3 tools/stress_step.py $nm > gpurun_out/kt_$nm.log 2>&1
grep " ms " gpurun_out/kt_$nm.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/kt_$nm/p_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fin = [i for i, r in enumerate(rows) if "k_pkt_finish" in r["Kernel_Name"]]
seg = rows[fin[-2] + 1:fin[-1] + 1]   # the last step: everything between the last two k_pkt_finish
t0 = int(seg[0]["Start_Timestamp"])
print("# the last step of the run ($nm): start (us), duration (us), kernel")
for r in seg:
    print("%9.1f +%8.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].replace("void nfc::", "").replace("nfc::", "")[:70]))
PY
