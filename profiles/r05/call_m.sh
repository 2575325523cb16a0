#!/bin/bash
# round 5, GPU call M: where the configs[4] rebuild (6.4 ms) goes: kernel trace of the mixture bench, rebuild kernels only
O=$PWD/gpurun_out/r05m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --mixture --rc 3.5 --steps 40 --warmup 10 > $O/bench.json 2> $O/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY' | tee $O/mixture_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    print("%-110s calls %5s  avg %10.1f us  total %9.2f ms  %5.1f %%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
find $O/trace -name "*.csv" -size +1M -delete
