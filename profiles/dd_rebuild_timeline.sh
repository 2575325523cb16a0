#!/bin/bash
# Kernel timeline of ONE rebuild of a decomposed run: every kernel between the last fused step before a rebuild and the
# first one after it, with the empty queue in front of each (rocprofv3 --kernel-trace).  One domain of a rank's size
# (68^3 x 4 atoms) pushed through the whole ownership path (EMDEE_DD_NO_SHORTCUT=1), in-order halo form.
# Usage (GPU box, repository root): bash profiles/dd_rebuild_timeline.sh [cells=68] [out=gpurun_out/ddtl]
R=$PWD; C=${1:-68}; O=${2:-gpurun_out/ddtl}; mkdir -p $R/$O; cd /tmp && export TMPDIR=/tmp
EMDEE_DD_NO_SHORTCUT=1 EMDEE_DD_OVERLAP=0 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python3 $R/profiles/dd_one_domain_overhead.py $C dd > $R/$O/run.log 2>&1
cd $R; grep atoms $O/run.log
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
def short(n): return n.split("(")[0].replace("void emdee::", "").replace("emdee::", "")[:60]
# rebuilds: runs of kernels between two fused step launches that contain a build kernel
step = [i for i, n in enumerate(names) if "k_brick<" in n or "k_brick(" in n]
step_set = set(step)
builds = [i for i, n in enumerate(names) if "k_brick_build" in n]
out = open(sys.argv[1] + "/timeline.txt", "w")
import statistics
tot, gaps, kern = [], [], []
for b in builds[3:]:
    lo = max(i for i in step if i < b)
    his = [i for i in step if i > b]
    if not his: continue
    hi = his[0]
    # (the first k_brick after the build is the rebuild's own force pass; the next one is the next step)
    his2 = [i for i in his if i > hi]
    hi2 = his2[0] if his2 else hi
    t0 = int(rows[lo]["End_Timestamp"]); t1 = int(rows[hi2]["Start_Timestamp"])
    k = sum(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(lo + 1, hi2))
    tot.append((t1 - t0) / 1e3); kern.append(k / 1e3)
which = builds[len(builds) // 2]
lo = max(i for i in step if i < which); hi = [i for i in step if i > which][1]
prev = int(rows[lo]["End_Timestamp"])
for i in range(lo, hi + 1):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    line = "%8.1f us gap  %8.1f us  %s" % ((s - prev) / 1e3 if i > lo else 0.0, (e - s) / 1e3, short(names[i]))
    print(line); out.write(line + "\n")
    prev = e
line = "rebuilds: %d; last step end -> next step start: median %.1f us, of which kernels %.1f us, empty queue %.1f us" % (
    len(tot), statistics.median(tot), statistics.median(kern), statistics.median(tot) - statistics.median(kern))
print(line); out.write(line + "\n")
PY
