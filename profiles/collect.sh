#!/bin/bash
# Produces the evidence kept under profiles/<round>/ : the default bench line, the rocprofv3 kernel
# statistics of the same command, the HBM traffic of the dominant kernel from PMC counters (FETCH_SIZE and
# WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) and its SQ instruction counters.
# Usage (on the GPU box, from the repository root):  bash profiles/collect.sh r01
TAG=${1:-r01}; R=$PWD; OUT=$R/gpurun_out/collect_$TAG; mkdir -p $OUT
timeout -k 10 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "bench failed"
cd /tmp && export TMPDIR=/tmp
# profiled runs queue one step per host round trip: with run-ahead batches some launches of the step kernel are
# no-ops (a rebuild was requested by the previous step) and would dilute the per-launch averages
export EMDEE_RUN_AHEAD=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $OUT/stats.log 2>&1 || echo "stats failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || echo "fetch failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || echo "write failed"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS"
SQ2="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $OUT/pmc_sq1 -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $OUT/pmc_sq1.log 2>&1 || echo "sq1 failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $OUT/pmc_sq2.log 2>&1 || echo "sq2 failed"
cd $R && python3 profiles/summarize.py $OUT $TAG
# the bench line once more, now quoting the traffic and instruction counts that were just measured for this build
unset EMDEE_RUN_AHEAD
timeout -k 10 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err && cp $OUT/bench_default.json profiles/$TAG/final_bench_default.json
