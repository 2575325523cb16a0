#!/bin/bash
# round 4, GPU call G: the new full-size decomposed tests + the transposed build's set tests
O=gpurun_out/r04g; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_dd.py tests/test_gpu_parity2.py -x -q -m gpu --timeout 600 -k "full_size or transposed" --durations=8 > $O/pytest.log 2>&1; echo "rc=$?"; tail -25 $O/pytest.log
