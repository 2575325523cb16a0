#!/bin/bash
# round 3 (second session), GPU call AI: the documented fallback switches still give a green suite
O=gpurun_out/r03ai; mkdir -p $O
EMDEE_READBACK=copy EMDEE_NO_SUBBINS=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -q --timeout 600 -k "not sub_bins and not posted_read" > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
EMDEE_READBACK=copy EMDEE_DD_LOCKSTEP=0 EMDEE_DD_COUNT_FREE=0 timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_domain.py -m gpu -q --timeout 600 -k "not count_free and not complete" > $O/pytest_b.log 2>&1; tail -3 $O/pytest_b.log
