#!/bin/bash
# round 4, GPU call E: transposed build with the written-out emission: sets, A/B (6 and 4 waves per SIMD, old build), kernel trace
O=$PWD/gpurun_out/r04e; mkdir -p $O; R=$PWD
timeout -k 10 500 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu --timeout 300 -k "neighbour or rebuilds" > $O/pytest_sets.log 2>&1; echo "sets rc=$?"; tail -2 $O/pytest_sets.log
bash profiles/ab_libs.sh $O "base tbw4" --steps 20 --warmup 5
EMDEE_NO_TBUILD=1 bash profiles/ab_libs.sh $O/old "base" --steps 20 --warmup 5
cd /tmp && export TMPDIR=/tmp
export EMDEE_RUN_AHEAD=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1 || echo "stats failed"
cd $R; F=$(find $O/stats -name "*kernel_stats.csv" | head -1); python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:6]:
    print("%-90s calls %5s avg %10.1f us total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
