#!/bin/bash
# round 5, GPU call Q (an experiment that was NOT kept: profiles/r05/dd_aligned_grid_experiment.txt; the switch it reads is gone with it):
# a domain's local grid aligned with its faces and with the kernels' bricks (dd.hpp DdGeom::init):
# the decomposition tests (bounds-checked library first), then one rank's cost by the replica rehearsal with the local box of
# rounds 1-4 (EMDEE_DD_ALIGNED_GRID=0) and with the aligned one, same box
O=gpurun_out/r05q; mkdir -p $O
EMDEE_HIP_LIB=$PWD/emdee.jl_amd/libemdee_hip_bounds.so timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_domain.py -x -q --timeout 600 -k "not full_size" > $O/pytest_bounds.log 2>&1; rc=$?; echo "pytest(bounds) rc=$rc"; tail -4 $O/pytest_bounds.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_domain.py tests/test_gpu_bench.py -x -q --timeout 600 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
for mode in old new old2 new2; do
  case $mode in old*) export EMDEE_DD_ALIGNED_GRID=0;; *) unset EMDEE_DD_ALIGNED_GRID;; esac
  timeout -k 10 500 python profiles/dd_rank_mirror.py 136 200 8 > $O/rank_mirror_136_8_$mode.txt 2>&1; echo "== $mode: cells 136 world 8"; grep -v amdgpu.ids $O/rank_mirror_136_8_$mode.txt
done
for mode in old new; do
  case $mode in old*) export EMDEE_DD_ALIGNED_GRID=0;; *) unset EMDEE_DD_ALIGNED_GRID;; esac
  timeout -k 10 500 python profiles/dd_rank_mirror.py 294 40 8 > $O/rank_mirror_294_8_$mode.txt 2>&1; echo "== $mode: cells 294 world 8"; grep -v amdgpu.ids $O/rank_mirror_294_8_$mode.txt
  timeout -k 10 500 python profiles/dd_rank_mirror.py 136 120 4 > $O/rank_mirror_136_4_$mode.txt 2>&1; echo "== $mode: cells 136 world 4"; grep -v amdgpu.ids $O/rank_mirror_136_4_$mode.txt
done
