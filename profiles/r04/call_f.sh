#!/bin/bash
# round 4, GPU call F: timing ablations of k_brick_build_t (1 no emission, 2 no distance tests, 4 no flush, 7 none of them) + SQ counters of the full kernel
O=$PWD/gpurun_out/r04f; mkdir -p $O; R=$PWD
export EMDEE_SKIP_CHECKS=1
bash profiles/ab_libs.sh $O "base tbabl1 tbabl2 tbabl4 tbabl7" --steps 12 --warmup 3 --rebuild-every 2
cd /tmp && export TMPDIR=/tmp
export EMDEE_RUN_AHEAD=1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/pmc_sq1.log 2>&1 || echo "sq1 failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/pmc_sq2.log 2>&1 || echo "sq2 failed"
cd $R; python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_sq1", "pmc_sq2"):
    for f in glob.glob("gpurun_out/r04f/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "k_brick_build" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            print(k)
            for c, v in sorted(acc[k].items()): print("   %-24s %14.4g per launch" % (c, v / n[(k, c)]))
PY
