#!/bin/bash
# round 3 (second session), GPU call Y: integer completeness check of the lists inside a decomposition
O=gpurun_out/r03y; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dd.py -m gpu -q --timeout 600 -k "complete or count_free" > $O/pytest.log 2>&1
tail -30 $O/pytest.log
