#!/bin/bash
O=$PWD/gpurun_out/r04o; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
export EMDEE_RUN_AHEAD=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1 || echo "stats failed"
cd $R; F=$(find $O/stats -name "*kernel_stats.csv" | head -1); python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:5]:
    print("%-100s calls %5s avg %10.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
