#!/bin/bash
# SQ counters of the fused force + integrate kernel (three passes of 8 SQ counters each).
# Usage (GPU box, repository root): bash profiles/pmc_force.sh <tag> [extra bench.py arguments, e.g. --precision f32]
# KERNEL_RE=<regex> selects another kernel (default: the fused step kernel), e.g. KERNEL_RE=k_brick_build with
# REBUILD_EVERY=2 so that the short profiled run contains rebuilds.
R=$PWD; TAG=${1:-pmc}; shift; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export EMDEE_RUN_AHEAD=1      # no no-op launches in the per-launch averages (profiles/README.md)
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH"
P3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/${TAG}_p$i -- python3 $R/bench.py --steps 6 --warmup 2 --rebuild-every ${REBUILD_EVERY:-1000} --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_p$i.log 2>&1 || echo "pass $i failed"
done
python3 - $R/gpurun_out $TAG "${KERNEL_RE:-k_brick<.*?, 3, 1[,>]}" <<'PY'
import csv, glob, re, sys, collections
root, tag, KRE = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(list)
dur = []
for f in glob.glob("%s/%s_p*/*/*_counter_collection.csv" % (root, tag)):
    for r in csv.DictReader(open(f)):
        if re.search(KRE, r["Kernel_Name"]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("%s/%s_p1/*/*_kernel_trace.csv" % (root, tag)):
    for r in csv.DictReader(open(f)):
        if re.search(KRE, r["Kernel_Name"]):
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
if dur:
    print("kernel /%s/, %d launches, mean %.3f ms (under the counter pass)" % (KRE, len(dur), sum(dur) / len(dur)))
for name, v in sorted(agg.items()):
    print("%-28s %16.0f  (avg of %d launches)" % (name, sum(v) / len(v), len(v)))
PY
