#!/bin/bash
# round 4, GPU call Y: the reference-shaped operator on its own (profiles/operator_path.py): a reload that re-sorts from the refreshed records
# against one that loads afresh from the caller's arrays (EMDEE_OPERATOR_RELOAD=1), after the operator / parity tests
O=gpurun_out/r04y; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_c_client.py -x -q -m gpu --timeout 600 -k "not hundred_million" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
timeout -k 10 400 python profiles/operator_path.py > $O/operator_path.txt 2>&1; tail -2 $O/operator_path.txt
EMDEE_OPERATOR_RELOAD=1 timeout -k 10 400 python profiles/operator_path.py > $O/operator_path_reload.txt 2>&1; tail -2 $O/operator_path_reload.txt
