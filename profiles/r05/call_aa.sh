O=$PWD/gpurun_out/r05aa; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
export EMDEE_RUN_AHEAD=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05 -- python3 $R/bench.py --no-cpu-baseline --precision f32 --steps 40 --warmup 10 > $O/r05.log 2>&1
cd $R/_r04
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04 -- python3 $R/_r04/bench.py --no-cpu-baseline --precision f32 --steps 40 --warmup 10 > $O/r04.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys
for tag in ("r04", "r05"):
    f = glob.glob(sys.argv[1] + "/%s/*/*kernel_stats.csv" % tag)[0]
    print("==", tag)
    for r in list(csv.DictReader(open(f)))[:16]:
        print("%-100s calls %4s avg %9.1f us total %8.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $O -name "*.csv" -size +1M -delete
