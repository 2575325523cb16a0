#!/bin/bash
# round 3 (second session), GPU call AA: DD suite with the ghost-capacity overflow case
O=gpurun_out/r03aa; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dd.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1
tail -15 $O/pytest.log
