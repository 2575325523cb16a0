#!/bin/bash
# round 5, GPU call J: exclusions and scaled 1-4 pairs (tests), headline unchanged when no table is set
O=gpurun_out/r05j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity2.py -x -q --timeout 600 -k "exclusions or operator_calls" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100 --warmup 20 > $O/head.json 2> $O/head.err; python -c "
import json; d=json.loads(open('$O/head.json').read().strip().splitlines()[-1]); print('headline', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"
