#!/bin/bash
# VALU issue floor (SQ counters) of the step kernel of one bench configuration -> profiles/valu.json entry + profiles/<tag>/valu_*.txt
# usage (GPU box, repository root): bash profiles/pmc_valu.sh TAG DTYPE RC MIXTURE [bench.py arguments for that configuration]
#   e.g.  bash profiles/pmc_valu.sh r05 f32 2.5 0 --precision f32      bash profiles/pmc_valu.sh r05 f64 3.5 1 --mixture --rc 3.5
TAG=$1; DT=$2; RC=$3; MIX=$4; shift 4
R=$PWD; OUT=$R/gpurun_out/valu_${TAG}_${DT}_${RC}_${MIX}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export EMDEE_RUN_AHEAD=1     # one step per host round trip: no no-op launches diluting the per-launch means
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
P2="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P3="SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_sq$i -- python3 $R/bench.py --steps 8 --warmup 3 --rebuild-every 1000 --no-cpu-baseline "$@" > $OUT/pmc_sq$i.log 2>&1 || echo "SQ pass $i failed (the entry is made from the passes that ran)"
done
cd $R
ATOMS=$(python3 -c "
import json
for l in open('$OUT/pmc_sq1.log'):
    if l.startswith('{') and '\"metric\"' in l:
        print(int(json.loads(l)['config']['atoms_per_gpu']))
")
python3 profiles/valu_entry.py $OUT $TAG --atoms $ATOMS --dtype $DT --rc $RC --mixture $MIX --command "--steps 8 --warmup 3 --rebuild-every 1000 $*"
