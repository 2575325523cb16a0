#!/bin/bash
# Kernel traces of emdee_dd_* with ONE domain and of the plain integrator on the same box (profiles/dd_one_domain_overhead.py):
# which kernels the decomposition adds per step and per rebuild.  Usage (GPU box, repository root): bash profiles/dd_overhead_trace.sh [cells=68]
R=$PWD; C=${1:-68}; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for m in dd plain; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ddo_$m -- python3 $R/profiles/dd_one_domain_overhead.py $C $m > $R/gpurun_out/ddo_$m.log 2>&1
done
cd $R
for m in dd plain; do
  echo "== $m"; grep atoms gpurun_out/ddo_$m.log
  python3 - $m <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/ddo_%s/*/*kernel_stats.csv" % sys.argv[1])[0]
tot = 0
for r in list(csv.DictReader(open(f))):
    tot += int(r["TotalDurationNs"])
    if float(r["Percentage"]) > 0.3:
        print("  %-72s %5s calls  avg %9.1f us  total %8.2f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6))
print("  total kernel time %.2f ms" % (tot / 1e6))
PY
done
