#!/bin/bash
# round 3 (second session), GPU call AB: count-free rebuild tests after the fix of the padded ghost pack (no row is packed once a capacity is exceeded)
O=gpurun_out/r03ab; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_dd.py -m gpu -q --timeout 200 -x -k "count_free" > $O/pytest_cf.log 2>&1; rc=$?
tail -5 $O/pytest_cf.log
if [ $rc -ne 0 ]; then echo "count_free tests rc=$rc: stopping"; exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_bench.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1
tail -5 $O/pytest.log
