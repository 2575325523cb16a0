#!/bin/bash
# round 3 (second session), GPU call AC: DD suite with the agreement check on the overflow word and Langevin through the lock-step halves
O=gpurun_out/r03ac; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dd.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1
tail -8 $O/pytest.log
