#!/bin/bash
# round 4, GPU call J: the bounds-checked build: injection test + DD / parity suites under it
O=gpurun_out/r04j; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_bounds.py -x -q -m gpu --timeout 900 --durations=5 > $O/pytest.log 2>&1; echo "rc=$?"; tail -30 $O/pytest.log
