#!/bin/bash
# Kernel timeline of ONE rebuild of the plain integrator at 10^7 atoms (rocprofv3 --kernel-trace): every kernel between the last fused
# step before a rebuild and the first one after it, with the empty queue in front of each.  Usage (GPU box): bash profiles/rebuild_timeline.sh [out]
R=$PWD; O=${1:-gpurun_out/rbtl}; mkdir -p $R/$O; cd /tmp && export TMPDIR=/tmp
EMDEE_RUN_AHEAD=1 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 10 > $R/$O/run.json 2> $R/$O/run.err
cd $R
python3 - $O <<'PY'
import csv, glob, sys, statistics
f = glob.glob(sys.argv[1] + "/trace/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
def short(n): return n.split("(")[0].replace("void emdee::", "").replace("emdee::", "")[:64]
step = [i for i, n in enumerate(names) if "k_brick<" in n and ", 3, 1, true>" in n]
builds = [i for i, n in enumerate(names) if "k_brick_build" in n]
tot, kern = [], []
for b in builds[2:]:
    lo = max([i for i in step if i < b], default=None); hi = min([i for i in step if i > b], default=None)
    if lo is None or hi is None: continue
    t0 = int(rows[lo]["End_Timestamp"]); t1 = int(rows[hi]["Start_Timestamp"])
    k = sum(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(lo + 1, hi))
    tot.append((t1 - t0) / 1e3); kern.append(k / 1e3)
which = builds[len(builds) // 2]
lo = max(i for i in step if i < which); hi = min(i for i in step if i > which)
prev = int(rows[lo]["End_Timestamp"]); out = open(sys.argv[1] + "/timeline.txt", "w")
for i in range(lo, hi + 1):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    line = "%8.1f us gap  %8.1f us  %s" % ((s - prev) / 1e3 if i > lo else 0.0, (e - s) / 1e3, short(names[i]))
    print(line); out.write(line + "\n"); prev = e
line = "rebuilds: %d; last step end -> next step start: median %.1f us, of which kernels %.1f us, empty queue %.1f us" % (len(tot), statistics.median(tot), statistics.median(kern), statistics.median(tot) - statistics.median(kern))
print(line); out.write(line + "\n")
PY
