#!/bin/bash
# round 5, GPU call AD: the operator alone (forces + energies + virials, positions unchanged), kernel by kernel, round 4's tree against this one
O=$PWD/gpurun_out/r05ad; mkdir -p $O; R=$PWD
export TMPDIR=/tmp
for t in _r04 .; do n=$( [ $t = . ] && echo r05 || echo r04 )
  cd $R/$t
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python3 profiles/operator_trace.py 7 > $O/$n.log 2>&1
done
cd $R
python3 - $O <<'PY'
import csv, glob, sys
for tag in ("r04", "r05"):
    f = glob.glob(sys.argv[1] + "/%s/*/*kernel_stats.csv" % tag)[0]
    print("==", tag)
    for r in list(csv.DictReader(open(f)))[:10]:
        print("%-110s calls %4s avg %9.1f us total %8.2f ms" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $O -name "*.csv" -size +1M -delete
