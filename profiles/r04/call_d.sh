#!/bin/bash
# round 4, GPU call D: why is k_brick_build_t 3x its instruction count?  (a) 4 waves/SIMD, no scratch; (b) SQ counters
O=$PWD/gpurun_out/r04d; mkdir -p $O; R=$PWD
bash profiles/ab_libs.sh $O "base tbw4" --steps 20 --warmup 5
cd /tmp && export TMPDIR=/tmp
export EMDEE_RUN_AHEAD=1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/pmc_sq1.log 2>&1 || echo "sq1 failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/pmc_sq2.log 2>&1 || echo "sq2 failed"
cd $R; python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_sq1", "pmc_sq2"):
    for f in glob.glob("gpurun_out/r04d/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "k_brick_build" not in k and "k_brick<" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            print(k)
            for c, v in sorted(acc[k].items()): print("   %-24s %14.4g per launch" % (c, v / n[(k, c)]))
PY
