#!/bin/bash
# Regenerates the files under profiles/ on the GPU box (outputs land in gpurun_out/prof, copy what is judged).
# usage: tools/profiles.sh <round-tag>
tag=${1:-r06}
out=gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o $tag -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-parity --no-extras --sync-steps > $out/${tag}_bench_under_rocprof.json 2> $out/stats.log
# the same with batches submitted ahead (--primary ahead): the threshold stage of batch k + 1 beside the later stages of batch k
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_ahead -o ${tag}a -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-parity --no-extras --primary ahead > $out/${tag}_bench_ahead_under_rocprof.json 2> $out/stats_ahead.log
python3 tools/timeline2.py $out/stats_ahead/${tag}a_kernel_trace.csv 60 22 > $out/${tag}_timeline_ahead.txt 2>&1
# counter passes, per workload (configs[1], [2], [3]): FETCH_SIZE, WRITE_SIZE and the SQ instruction counters each in a pass of its own
for wl in miller manchester classic1k env; do
  extra=""; [ $wl = classic1k ] && extra="--samples 1e9"
  wlarg="--workload $wl"; [ $wl = env ] && wlarg="--input-kind env"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch_$wl -o p -- python3 bench.py $wlarg $extra --steps 2 --warmup 0 --no-cpu-baseline --no-parity --no-extras --sync-steps > /dev/null 2> $out/fetch_$wl.log
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write_$wl -o p -- python3 bench.py $wlarg $extra --steps 2 --warmup 0 --no-cpu-baseline --no-parity --no-extras --sync-steps > /dev/null 2> $out/write_$wl.log
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $out/sq_$wl -o p -- python3 bench.py $wlarg $extra --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-extras --sync-steps > /dev/null 2> $out/sq_$wl.log
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -o ${tag}_$wl -- python3 bench.py $wlarg $extra --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras --sync-steps > $out/${tag}_bench_${wl}_under_rocprof.json 2> $out/stats_$wl.log
done
# the other input kinds of the boundary on configs[1]: the float32 envelope (what transition_sink.work receives) and 16-bit PCM
for kd in i16; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$kd -o ${tag}_$kd -- python3 bench.py --input-kind $kd --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras --sync-steps > $out/${tag}_bench_${kd}_under_rocprof.json 2> $out/stats_$kd.log
done
# what a read-only kernel with the threshold kernel's access pattern reaches on this box (tools/ubench/stream_chunks.hip)
[ -x tools/ubench/stream_chunks ] && tools/ubench/stream_chunks > $out/${tag}_stream_ceiling.txt 2>&1
# where the workgroups of the later stages spend their time (a build with -DNFC_TAIL_PROF: scratch/r5/tailprof.so), and the waves of
# the re-run kernel on the stress captures (-DNFC_GEN_PROF: scratch/r5/genprof.so)
# (the switches -- NFC_DEC_SPEC, NFC_TAIL -- exist in the test build only: the profiling builds carry -DNFC_TEST_HOOKS too)
if [ -f scratch/r6/tailprof.so ]; then
  { echo "# NFC_AMD_LIB=<-DNFC_TEST_HOOKS -DNFC_TAIL_PROF build> python tools/tailprof.py miller 1e8   (s_memtime ticks, thread 0 of every workgroup)"; NFC_AMD_LIB=scratch/r6/tailprof.so python3 tools/tailprof.py miller 1e8; echo "# ... NFC_DEC_SPEC=0: the three-launch decode"; NFC_DEC_SPEC=0 NFC_AMD_LIB=scratch/r6/tailprof.so python3 tools/tailprof.py miller 1e8;
    echo "# ... NFC_TAIL=1: the fused tail (tail.hip.h: built, measured, not adopted), ticks summed over a workgroup's tiles"; NFC_TAIL=1 NFC_AMD_LIB=scratch/r6/tailprof.so python3 tools/tailprof6.py miller 1e8; } > $out/${tag}_tail_phases.txt 2>&1
fi
if [ -f scratch/r6/genprof.so ]; then
  { echo "# NFC_AMD_LIB=<-DNFC_GEN_PROF build> python tools/genprof.py hover / dropsteps   (the atomics of the iteration counter stretch the ticks: read the counts)"; NFC_AMD_LIB=scratch/r6/genprof.so python3 tools/genprof.py hover; NFC_AMD_LIB=scratch/r6/genprof.so python3 tools/genprof.py dropsteps; } > $out/${tag}_rerun_phases.txt 2>&1
fi
# the fused tail against the five launches it would replace: same call, alternating (tools/ab.py env: the test build)
{ echo "# python tools/ab.py env miller --rounds 3 NFC_TAIL=0 NFC_TAIL=1"; python3 tools/ab.py env miller --rounds 3 "NFC_TAIL=0" "NFC_TAIL=1";
  echo "# per kernel under rocprofv3, NFC_TAIL=1 (test build)"; NFC_TAIL=1 NFC_AMD_LIB=usrp_nfc_amd/libnfc_amd_hooks.so bash tools/kstats.sh miller tail1 | grep -v "^$";
  echo "# chunks cut by dispatch row against the equal cut, same call, alternating"; python3 tools/ab.py env miller --rounds 4 "NFC_WG_ROWBAL=0" "NFC_WG_ROWBAL=1"; } > $out/${tag}_tail_fused_ab.txt 2>&1
# per-wave counters instead of the first barrier of a round (k_threshold_wg<KIND, NR, false, true>, test build): same call, alternating
{ echo "# python tools/ab.py env miller --rounds 4 NFC_WG_FLAGS=0 NFC_WG_FLAGS=1"; python3 tools/ab.py env miller --rounds 4 "NFC_WG_FLAGS=0" "NFC_WG_FLAGS=1";
  echo "# python tools/ab.py env manchester --rounds 2 NFC_WG_FLAGS=0 NFC_WG_FLAGS=1"; python3 tools/ab.py env manchester --rounds 2 "NFC_WG_FLAGS=0" "NFC_WG_FLAGS=1";
  echo "# python tools/ab.py env classic1k --rounds 2 --bench '--samples 1e9 --steps 10 --warmup 2' NFC_WG_FLAGS=0 NFC_WG_FLAGS=1   (eight rows per step, the staging ring)"; python3 tools/ab.py env classic1k --rounds 2 --bench "--samples 1e9 --steps 10 --warmup 2" "NFC_WG_FLAGS=0" "NFC_WG_FLAGS=1";
  if [ -f scratch/r6/wgprof.so ]; then echo "# where a wave waits (a -DNFC_WG_PROF -DNFC_TEST_HOOKS build, NFC_DEBUG_CLK=1: s_memtime ticks per wave of a chunk): the barrier form, then NFC_WG_FLAGS=1";
    NFC_DEBUG_CLK=1 NFC_AMD_LIB=scratch/r6/wgprof.so python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extras --sync-steps 2>&1 | grep "wg kernel" | tail -1;
    NFC_WG_FLAGS=1 NFC_DEBUG_CLK=1 NFC_AMD_LIB=scratch/r6/wgprof.so python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extras --sync-steps 2>&1 | grep "wg kernel" | tail -1; fi; } > $out/${tag}_wg_flags_ab.txt 2>&1
# the re-runs that evaluate failed rounds in place (k_threshold_wg<KIND, 4, true>) against k_threshold re-running everything: same call, alternating
{ echo "# python tools/ab.py stress stress_dropouts_steps NFC_WG_EX=0 NFC_WG_EX=1024 NFC_WG_EX=0 NFC_WG_EX=1024   (ms per batch, threshold launches, chunks re-run)"; python3 tools/ab.py stress stress_dropouts_steps "NFC_WG_EX=0" "NFC_WG_EX=1024" "NFC_WG_EX=0" "NFC_WG_EX=1024";
  echo "# ... the capture where EVERY chunk fails, with the limit of a machine-full lifted: python tools/ab.py stress stress_hover NFC_WG_EX=1024 NFC_WG_EX=100000"; python3 tools/ab.py stress stress_hover "NFC_WG_EX=1024" "NFC_WG_EX=100000";
  if [ -f scratch/r6/exprintf.so ]; then echo "# NFC_AMD_LIB=<-DNFC_TEST_HOOKS -DNFC_EX_PRINTF build> python tools/stress_step.py stress_dropouts_steps | python tools/exrounds_summary.py"; NFC_AMD_LIB=scratch/r6/exprintf.so python3 tools/stress_step.py stress_dropouts_steps 2>&1 | python3 tools/exrounds_summary.py; fi; } > $out/${tag}_rerun_in_place_ab.txt 2>&1
for nm in stress_dropouts_steps stress_hover; do tools/stress_timeline.sh $nm > $out/${tag}_timeline_$nm.txt 2>&1; done
timeout 600 python3 bench.py > $out/${tag}_bench.json 2> $out/bench.log
python3 - <<PY
import csv, json, collections
out = "$out"; tag = "$tag"
rows = list(csv.DictReader(open(f"{out}/stats/{tag}_kernel_stats.csv")))
thr = [r for r in rows if "k_threshold" in r["Name"]]
nstep = sum(int(r["Calls"]) for r in thr) if thr else 1
lines = []; tot = 0
for r in rows:
    per = float(r["TotalDurationNs"]) / nstep / 1e3; tot += per
    lines.append("%-90s calls/step %5.2f  avg %8.1f us  per-step %8.1f us" % (r["Name"].replace("nfc::", "").replace("void ", "")[:88], int(r["Calls"]) / nstep, float(r["AverageNs"]) / 1e3, per))
lines.append("total per step %.1f us over %d steps (incl. warm-up and the untimed extra step); --sync-steps: one batch at a time" % (tot, nstep))
open(f"{out}/{tag}_kernel_stats_per_step.txt", "w").write("\n".join(lines) + "\n")
# the same table for the run with batches submitted ahead (kernels of consecutive batches overlap: durations are longer, the step shorter)
rows = list(csv.DictReader(open(f"{out}/stats_ahead/{tag}a_kernel_stats.csv")))
thr = [r for r in rows if "k_threshold" in r["Name"]]
nstep = sum(int(r["Calls"]) for r in thr) if thr else 1
lines = []; tot = 0
for r in rows:
    per = float(r["TotalDurationNs"]) / nstep / 1e3; tot += per
    lines.append("%-90s calls/step %5.2f  avg %8.1f us  per-step %8.1f us" % (r["Name"].replace("nfc::", "").replace("void ", "")[:88], int(r["Calls"]) / nstep, float(r["AverageNs"]) / 1e3, per))
lines.append("sum of kernel durations per step %.1f us over %d steps -- they overlap (two queues): see the timeline" % (tot, nstep))
open(f"{out}/{tag}_kernel_stats_ahead_per_step.txt", "w").write("\n".join(lines) + "\n")
def pmc(d, name):
    rs = [r for r in csv.DictReader(open(f"{out}/{d}/p_counter_collection.csv")) if "k_threshold" in r["Kernel_Name"] and r["Counter_Name"] == name]
    byd = collections.defaultdict(float)
    for r in rs: byd[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = sorted(byd.values()); return v[len(v) // 2]
recs = []
for wl, ns in (("miller", 100000000), ("manchester", 100000000), ("classic1k", 1000000000), ("env", 100000000)):
    try:
        f, w = pmc("fetch_" + wl, "FETCH_SIZE"), pmc("write_" + wl, "WRITE_SIZE")
        kname = sorted({r["Kernel_Name"] for r in csv.DictReader(open(f"{out}/fetch_{wl}/p_counter_collection.csv")) if "k_threshold" in r["Kernel_Name"]})
        recs.append({"workload": "miller" if wl == "env" else wl, "input_kind": "env" if wl == "env" else "iq", "samples": ns, "kernel": ", ".join(k.replace("void nfc::", "").split("(")[0] for k in kname), "FETCH_SIZE_kb": round(f), "WRITE_SIZE_kb": round(w),
                     "correction": "FETCH_SIZE x2 (gfx950: TCC_EA0_RDREQ counted at 64 B per 128-B request, MI355X_MICROARCH.md section HBM); WRITE_SIZE as reported",
                     "bytes_per_launch": int((2 * f + w) * 1024),
                     "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --workload %s%s --steps 2 --warmup 0 --no-cpu-baseline --no-parity --no-extras --sync-steps   (and the same with --pmc WRITE_SIZE); median over the launches" % (wl, " --samples 1e9" if wl == "classic1k" else "")})
    except Exception as e:
        print("no traffic record for", wl, e)
rec = recs
json.dump(recs, open(f"{out}/hbm_traffic.json", "w"), indent=1)
# instruction counts per wave of every kernel of a step (one --pmc pass of SQ counters), per workload
with open(f"{out}/{tag}_sq_counters.txt", "w") as fh:
    for wl in ("miller", "manchester", "classic1k", "env"):
        try:
            rows = list(csv.DictReader(open(f"{out}/sq_{wl}/p_counter_collection.csv")))
        except Exception as e:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
        for r in rows:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void nfc::", "").replace("nfc::", "").replace("void ", "").split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
        fh.write("== %s: per dispatch: waves, and VALU / SALU / LDS / VMEM-read / VMEM-write instructions per wave\n" % wl)
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
            n = len(nd[k]); wv = max(v["SQ_WAVES"] / n, 1)
            fh.write("%-62s dispatches %3d waves %8d  valu %7.0f salu %7.0f lds %6.0f vmem_rd %6.1f vmem_wr %6.1f\n" % (
                k, n, wv, v["SQ_INSTS_VALU"] / n / wv, v["SQ_INSTS_SALU"] / n / wv, v["SQ_INSTS_LDS"] / n / wv, v["SQ_INSTS_VMEM_RD"] / n / wv, v["SQ_INSTS_VMEM_WR"] / n / wv))
# per-workload kernel tables (one batch at a time)
for wl in ("manchester", "classic1k", "env", "i16"):
    try:
        rows = list(csv.DictReader(open(f"{out}/stats_{wl}/{tag}_{wl}_kernel_stats.csv")))
    except Exception as e:
        continue
    thr = [r for r in rows if "k_threshold" in r["Name"]]
    nstep = sum(int(r["Calls"]) for r in thr) if thr else 1
    lines = []; tot = 0
    for r in rows:
        per = float(r["TotalDurationNs"]) / nstep / 1e3; tot += per
        lines.append("%-90s calls/step %5.2f  avg %8.1f us  per-step %8.1f us" % (r["Name"].replace("nfc::", "").replace("void ", "")[:88], int(r["Calls"]) / nstep, float(r["AverageNs"]) / 1e3, per))
    lines.append("total per step %.1f us over %d steps; --sync-steps: one batch at a time" % (tot, nstep))
    open(f"{out}/{tag}_kernel_stats_per_step_{wl}.txt", "w").write("\n".join(lines) + "\n")
print(open(f"{out}/{tag}_sq_counters.txt").read())
print(rec)
print(open(f"{out}/{tag}_kernel_stats_per_step.txt").read())
PY
tail -1 $out/${tag}_bench.json | cut -c1-1500
