#!/bin/bash
# round 4, GPU call X: k_species_collect with one look-up per distinct key of a wavefront: its time at 10^7 atoms (single species and mixture)
O=$PWD/gpurun_out/r04x; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
for M in "" "--mixture --rc 3.5"; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline $M > $O/st.log 2>&1
F=$(find $O/st -name "*kernel_stats.csv" | head -1); python3 - "$F" "$M" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_species_collect" in r["Name"] or "k_atoms_differ" in r["Name"]:
        print("%-22s %-40s calls %s avg %.1f us" % (sys.argv[2] or "single species", r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $O/st
done
