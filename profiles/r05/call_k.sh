#!/bin/bash
# round 5, GPU call K: cell-relative fp32 records: fp32 tests, energy drift at 10^7 atoms against absolute records and against the
# fp64 run of the same start (same box), the fp32 bench both ways, the fp32 10^8-atom box once
O=gpurun_out/r05k; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -x -q --timeout 600 -k "f32 or fp32 or mixed or precision or float32 or trajectory or state or checkpoint" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python profiles/fp32_drift.py 136 400 > $O/drift_rel.txt 2>&1; grep records $O/drift_rel.txt
EMDEE_F32_ABS=1 timeout -k 10 300 python profiles/fp32_drift.py 136 400 > $O/drift_abs.txt 2>&1; grep records $O/drift_abs.txt
F64=1 timeout -k 10 300 python profiles/fp32_drift.py 136 400 > $O/drift_f64.txt 2>&1; grep records $O/drift_f64.txt
for mode in rel abs; do
  if [ $mode = abs ]; then export EMDEE_F32_ABS=1; else unset EMDEE_F32_ABS; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --precision f32 --steps 60 --warmup 10 > $O/bench_f32_$mode.json 2> $O/bench_f32_$mode.err
  python -c "
import json; d=json.loads(open('$O/bench_f32_$mode.json').read().strip().splitlines()[-1]); k=d['kernels_ms']; rb=k['rebuild(bin+sort+nbr_build)']
print('fp32 bench $mode', round(d['value'],1), 'steps/s, fused', round(d['roofline']['avg_launch_ms'],4), 'ms, rebuild', round(rb[0]/max(rb[1],1),3), 'ms x', rb[1])"
done
unset EMDEE_F32_ABS
timeout -k 10 600 python profiles/fp32_drift.py 293 400 > $O/drift_rel_1e8.txt 2>&1; grep records $O/drift_rel_1e8.txt
EMDEE_F32_ABS=1 timeout -k 10 600 python profiles/fp32_drift.py 293 400 > $O/drift_abs_1e8.txt 2>&1; grep records $O/drift_abs_1e8.txt
