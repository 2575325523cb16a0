#!/bin/bash
# round 4, GPU call W: the first parity file and the domain / C-client tests once under the bounds-checked build (not part of the suite: time)
O=gpurun_out/r04w; mkdir -p $O
EMDEE_HIP_LIB=$PWD/emdee.jl_amd/libemdee_hip_bounds.so timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_domain.py -q -m gpu --timeout 600 -k "not hundred_million" -p no:cacheprovider > $O/pytest_bounds.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_bounds.log
