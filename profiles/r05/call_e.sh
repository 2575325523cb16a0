#!/bin/bash
# round 5, GPU call E: the stream-fence test with and without the fence (the second must FAIL: the test sees the race);
# bench per-rank line on two RCCL ranks; one rank's cost per step through the production path at three rank sizes
O=gpurun_out/r05e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dd.py -x -q --timeout 600 -k "ordered_on_the_callers or replica" > $O/pytest_fence.log 2>&1; rc=$?; echo "pytest fence rc=$rc"; tail -5 $O/pytest_fence.log
EMDEE_HIP_LIB=$PWD/emdee.jl_amd/variants/libemdee_hip_nofence.so timeout -k 10 600 python -m pytest tests/test_gpu_dd.py -x -q --timeout 600 -k "ordered_on_the_callers" > $O/pytest_nofence.log 2>&1; echo "pytest WITHOUT the fence rc=$? (expected: 1)"; tail -5 $O/pytest_nofence.log
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py -x -q --timeout 600 > $O/pytest_bench.log 2>&1; rc=$?; echo "pytest bench rc=$rc"; tail -5 $O/pytest_bench.log
for cfg in "136 200 8" "136 120 4" "136 80 2" "294 40 8"; do
  set -- $cfg
  timeout -k 10 500 python profiles/dd_rank_mirror.py $1 $2 $3 > $O/rank_mirror_$1_$3.txt 2>&1; echo "== cells $1 world $3"; grep -v amdgpu.ids $O/rank_mirror_$1_$3.txt
done
