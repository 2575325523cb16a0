#!/bin/bash
# round 5, GPU call D: the replica rehearsal of one rank (mirror transport): its tests, then its cost per step at rank size
O=gpurun_out/r05d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dd.py -x -q --timeout 600 -k "replica or count_free or trajectory" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python profiles/dd_rank_mirror.py 136 200 8 > $O/rank_mirror_136.txt 2>&1; grep -v amdgpu.ids $O/rank_mirror_136.txt
