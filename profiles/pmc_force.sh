#!/bin/bash
# SQ counters of the force kernel (two passes: 8 SQ slots each). Usage: pmc_force.sh <outdir-tag>
R=$PWD; TAG=${1:-pmc}; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH"
P3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_UNALIGNED_STALL"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/${TAG}_p$i -- python3 $R/bench.py --steps 6 --warmup 2 --rebuild-every 1000 --no-cpu-baseline > $R/gpurun_out/${TAG}_p$i.log 2>&1 || echo "pass $i failed"
done
python3 - $R/gpurun_out $TAG <<'PY'
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/%s_p*/*/*_counter_collection.csv" % (root, tag)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_brick<" in k and (", 1, 1>" in k or ", 3, 1>" in k):
            agg["force"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, v in sorted(agg["force"].items()):
    print("%-28s %16.0f  (avg of %d launches)" % (name, sum(v) / len(v), len(v)))
PY
