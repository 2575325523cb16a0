#!/bin/bash
# round 4, GPU call L: build kernel with the early exit of empty wavefront-rounds + whole-record tail read; sets + bench
O=gpurun_out/r04l; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu --timeout 300 -k "neighbour or rebuilds" > $O/pytest_sets.log 2>&1; echo "sets rc=$?"; tail -2 $O/pytest_sets.log
bash profiles/ab_libs.sh $O "base" --steps 100 --warmup 20
bash profiles/ab_libs.sh $O/d "base" --steps 20 --warmup 5
