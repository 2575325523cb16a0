#!/bin/bash
# round 4, GPU call N: the round-robin build with FOUR lanes per atom (768-thread workgroups): neighbour-set tests, A/B against 8 lanes
O=gpurun_out/r04n; mkdir -p $O
EMDEE_DEBUG_PLAN=1 timeout -k 10 500 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_parity.py -x -q -m gpu --timeout 300 -k "neighbour or rebuilds or medium_box or random_boxes or density or million_atoms_prop" > $O/pytest_sets.log 2>&1; echo "sets rc=$?"; tail -3 $O/pytest_sets.log
for V in b4 b8; do
  if [ $V = b8 ]; then export EMDEE_BUILD4=0; else unset EMDEE_BUILD4; fi
  bash profiles/ab_libs.sh $O/$V "base" --steps 100 --warmup 20
  bash profiles/ab_libs.sh $O/${V}d "base" --steps 20 --warmup 5
done
