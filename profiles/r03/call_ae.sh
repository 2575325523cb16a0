#!/bin/bash
# round 3 (second session), GPU call AE: kernel stats of the 8-domain one-GPU rehearsal
O=$PWD/gpurun_out/r03ae; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dd8 -- python3 $R/bench.py --no-cpu-baseline --domains 8 --steps 40 --warmup 10 > $O/dd8.json 2> $O/dd8.err
cd $R
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/r03ae/dd8/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
for r in rows[:26]:
    print("  %-70s %6s calls  avg %9.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6))
print("  total kernel time %.2f ms in %d launches" % (tot / 1e6, calls))
d = json.loads(open("gpurun_out/r03ae/dd8.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
PY
