#!/bin/bash
# round 4, GPU call T: near/far rows with sub-bins and ten paired emission loops: set tests under EMDEE_BUILD_NEARFAR=1, A/B against plain rows
O=$PWD/gpurun_out/r04t; mkdir -p $O
EMDEE_BUILD_NEARFAR=1 timeout -k 10 500 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_parity.py -x -q -m gpu --timeout 300 -k "neighbour or rebuilds or medium_box or random_boxes or density or million_atoms_prop or reproducible" > $O/pytest_sets.log 2>&1; echo "sets rc=$?"; tail -3 $O/pytest_sets.log
bash profiles/ab_libs.sh $O/plain "base" --steps 100 --warmup 20
EMDEE_BUILD_NEARFAR=1 bash profiles/ab_libs.sh $O/nf "base" --steps 100 --warmup 20
bash profiles/ab_libs.sh $O/plaind "base" --steps 20 --warmup 5
EMDEE_BUILD_NEARFAR=1 bash profiles/ab_libs.sh $O/nfd "base" --steps 20 --warmup 5
