#!/bin/bash
# round 5, GPU call AG: the steps that wait for a rebuild request run in order (dd.hpp emdee_dd_step, EMDEE_DD_HOLD_INTERIOR=0 = overlapped
# throughout as before): decomposition tests, then profiles/soak_dd.py both ways on one box -- eight in-process domains (where seven of
# eight domains learn of a rebuild from a neighbour, as real ranks do) and one rank by the replica rehearsal (where none does)
O=gpurun_out/r05ag; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_domain.py tests/test_gpu_bench.py -x -q --timeout 600 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
for mode in hold overlapped hold2 overlapped2; do
  case $mode in overlapped*) export EMDEE_DD_HOLD_INTERIOR=0;; *) unset EMDEE_DD_HOLD_INTERIOR;; esac
  timeout -k 10 400 python profiles/soak_dd.py 136 2 500 8 > $O/soak_$mode.txt 2>&1
  echo "== $mode"; grep -E "step  1050" $O/soak_$mode.txt | cut -c1-60,150-400
done
