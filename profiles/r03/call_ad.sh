#!/bin/bash
# round 3 (second session), GPU call AD: parity suites after the sub-bin margin became a function of M and the precision; fp32 and fp64 bench lines
O=gpurun_out/r03ad; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_parity.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for a in "--precision f32 --steps 60 --warmup 10" "--steps 40 --warmup 10"; do
EMDEE_DEBUG_PLAN=1 timeout -k 10 300 python bench.py --no-cpu-baseline $a > $O/b.json 2> $O/b.err
python -c "
import json
d=json.loads(open('gpurun_out/r03ad/b.json').read().strip().splitlines()[-1]); rb=d['kernels_ms']['rebuild(bin+sort+nbr_build)']; print('$a', d['value'], rb[0]/rb[1])"
grep "emdee plan" $O/b.err | tail -1
done
