#!/bin/bash
# round 3 (second session), GPU call M: where the 2-domain one-GPU rehearsal spends its time (kernel stats), both halo forms
O=$PWD/gpurun_out/r03m; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dd2 -- python3 $R/bench.py --no-cpu-baseline --domains 2 --steps 40 --warmup 10 > $O/dd2.json 2> $O/dd2.err
cd $R
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/r03m/dd2/*/*kernel_stats.csv")[0]
tot = 0
rows = list(csv.DictReader(open(f)))
for r in rows: tot += int(r["TotalDurationNs"])
for r in rows[:22]:
    print("  %-80s %5s calls  avg %9.1f us  total %8.2f ms" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6))
print("  total kernel time %.2f ms" % (tot / 1e6))
d = json.loads(open("gpurun_out/r03m/dd2.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
PY
for ov in 0 1; do
EMDEE_DD_OVERLAP=$ov timeout -k 10 300 python bench.py --no-cpu-baseline --domains 2 --steps 40 --warmup 10 > $O/dd2_ov$ov.json 2> $O/dd2_ov$ov.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r03m/dd2_ov$ov.json').read().strip().splitlines()[-1]); print('overlap $ov', d['value'], d['ms_per_step'])"
done
