#!/bin/bash
# HBM traffic (PMC) of the step kernel of one bench configuration -> profiles/traffic.json entry + profiles/<tag>/traffic_*.txt
# usage (GPU box, repository root): bash profiles/pmc_traffic.sh TAG DTYPE RC MIXTURE [bench.py arguments for that configuration]
#   e.g.  bash profiles/pmc_traffic.sh r04 f32 2.5 0 --precision f32      bash profiles/pmc_traffic.sh r04 f64 3.5 1 --mixture --rc 3.5
TAG=$1; DT=$2; RC=$3; MIX=$4; shift 4
R=$PWD; OUT=$R/gpurun_out/traffic_${TAG}_${DT}_${RC}_${MIX}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export EMDEE_RUN_AHEAD=1     # one step per host round trip: no no-op launches diluting the per-launch means
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1 || { echo "fetch pass failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1 || { echo "write pass failed"; exit 1; }
cd $R
ATOMS=$(python3 -c "
import json
for l in open('$OUT/pmc_fetch.log'):
    if l.startswith('{') and '\"metric\"' in l:
        print(int(json.loads(l)['config']['atoms_per_gpu']))
")
python3 profiles/traffic_entry.py $OUT $TAG --atoms $ATOMS --dtype $DT --rc $RC --mixture $MIX --command "--steps 12 --warmup 3 $*"
