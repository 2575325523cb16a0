#!/bin/bash
# round 3 (second session), GPU call AJ: typed tables skip (species, cell) blocks of ghosts: DD suite (typed decomposed cases included), mixture DD bench
O=gpurun_out/r03aj; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_bench.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --mixture --rc 3.5 --domains 2 --target-cells 0 --steps 40 --warmup 10 > $O/bench_mix35_dd2.json 2> $O/bench_mix35_dd2.err
python -c "
import json
d=json.loads(open('gpurun_out/r03aj/bench_mix35_dd2.json').read().strip().splitlines()[-1]); print('mix35 dd2', d['value'], d['ms_per_step'], d['energy_per_atom'])"
