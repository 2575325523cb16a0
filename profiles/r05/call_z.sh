#!/bin/bash
# round 5, GPU call Z: as call Y (round 4's tree against this round's, one box), after the in-cell ranking went back to key-only compares where
# keys are unique: headline, driver form, fp32, twice each; then the sort/neighbour-set tests
O=gpurun_out/r05z; mkdir -p $O
run() { tree=$1; name=$2; shift 2
  ( cd $tree && timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $OLDPWD/$O/bench_${name}.json 2> $OLDPWD/$O/bench_${name}.err )
  python -c "
import json; d=json.loads(open('$O/bench_${name}.json').read().strip().splitlines()[-1]); k=d['kernels_ms']; rb=k['rebuild(bin+sort+nbr_build)']
print('%-22s' % '$name', round(d['value'],1), 'steps/s, step kernel', round(d['roofline']['avg_launch_ms'],4), 'ms, rebuild', round(rb[0]/max(rb[1],1),3), 'ms x', rb[1])"
}
for rep in 1 2; do
  run _r04 r04_f32_$rep --precision f32 --steps 60 --warmup 10
  run .    r05_f32_$rep --precision f32 --steps 60 --warmup 10
  run _r04 r04_head_$rep --steps 100 --warmup 20
  run .    r05_head_$rep --steps 100 --warmup 20
  run _r04 r04_driver_$rep --steps 20 --warmup 5
  run .    r05_driver_$rep --steps 20 --warmup 5
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_parity.py -x -q --timeout 600 -k "neighbour_set or cells or sort or sub_bins or deterministic or canonical" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
