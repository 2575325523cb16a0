#!/bin/bash
# round 5, GPU call L: two-species long-row boxes on 2 x 2 x 2 bricks (512 threads, two workgroups per CU) and the two-pass
# typed build: the typed tests, then configs[4] (binary mixture, rc = 3.5) three ways on one box:
# this build (default), this build with EMDEE_TYPED_BRICKS=7 (the 4 x 2 x 2 bricks of round 4) (the round-4 library lacks this round's entry points: its numbers are round 4's, profiles/README.md)
O=gpurun_out/r05l; mkdir -p $O
EMDEE_HIP_LIB=$PWD/emdee.jl_amd/libemdee_hip_bounds.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_gpu_dd.py -x -q --timeout 600 -k "typed or species or mixture or long_rows" > $O/pytest_bounds.log 2>&1; rc=$?; echo "pytest(bounds) rc=$rc"; tail -4 $O/pytest_bounds.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_gpu_dd.py -x -q --timeout 600 -k "typed or species or mixture or long_rows" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
run() {  # name, env...
  name=$1; shift
  env "$@" EMDEE_DEBUG_PLAN=1 timeout -k 10 300 python bench.py --no-cpu-baseline --mixture --rc 3.5 --steps 40 --warmup 10 > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); k=d['kernels_ms']; rb=k['rebuild(bin+sort+nbr_build)']
print('mixture $name', round(d['value'],1), 'steps/s, step kernel', round(d['roofline']['avg_launch_ms'],4), 'ms, rebuild', round(rb[0]/max(rb[1],1),3), 'ms x', rb[1])"
  grep -i -m3 "plan" $O/bench_$name.err
}
run v9 A=1 && run v9_nosub EMDEE_TYPED_SUBBINS=0 && run v7 EMDEE_TYPED_BRICKS=7 && run v9_again A=1
