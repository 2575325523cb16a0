#!/bin/bash
# round 4, GPU call Q: SQ counters of the 4-lane build (EMDEE_BUILD4=1) next to the 8-lane one
O=$PWD/gpurun_out/r04q; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
export EMDEE_RUN_AHEAD=1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_LDS"
for V in 1 0; do
export EMDEE_BUILD4=$V
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/b$V/pmc_sq1 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/b${V}_sq1.log 2>&1 || echo "sq1 failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/b$V/pmc_sq2 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/b${V}_sq2.log 2>&1 || echo "sq2 failed"
done
cd $R; python3 - <<'PY'
import csv, glob, collections
for v in ("1", "0"):
    print("EMDEE_BUILD4=%s" % v)
    for d in ("pmc_sq1", "pmc_sq2"):
        for f in glob.glob("gpurun_out/r04q/b%s/%s/**/*counter_collection.csv" % (v, d), recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"][:70]
                if "k_brick_build" not in k: continue
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
            for k in acc:
                print(" ", k)
                for c, val in sorted(acc[k].items()): print("   %-24s %14.4g per launch" % (c, val / n[(k, c)]))
PY
