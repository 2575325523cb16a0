#!/bin/bash
# round 4, GPU call Z: kernel timeline of the operator path at a reload (the list outrun by the caller's positions)
O=$PWD/gpurun_out/r04z; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/profiles/operator_path.py > $O/log.txt 2>&1
cd $R; F=$(find $O/tr -name "*kernel_trace.csv" | head -1); python3 - "$F" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# find the LAST k_gather_user (a reload in steady state) and print the kernels from the previous force launch to the next one
idx = [i for i, r in enumerate(rows) if "k_gather_user" in r["Kernel_Name"]]
i = idx[-2]
a = max(j for j in range(i) if "k_brick<" in rows[j]["Kernel_Name"])
b = min(j for j in range(i, len(rows)) if "k_brick<" in rows[j]["Kernel_Name"] and j > i) 
prev = None
for r in rows[a:b + 2]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us gap %9.1f us  %s" % ((s - prev) / 1e3 if prev else 0.0, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev = e
PY
