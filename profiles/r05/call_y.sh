#!/bin/bash
# round 5, GPU call Y: round 4's tree (git worktree of 1045b3c under _r04/, built in the container) against this round's on ONE box, every
# single-GPU configuration, alternating -- is anything slower than it was?
O=gpurun_out/r05y; mkdir -p $O
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
run _r04 r04_1e6 --cells 63 --steps 100 --warmup 20
run .    r05_1e6 --cells 63 --steps 100 --warmup 20
run _r04 r04_864 --cells 6 --steps 200 --warmup 20
run .    r05_864 --cells 6 --steps 200 --warmup 20
run _r04 r04_1e8 --cells 293 --steps 30 --warmup 8
run .    r05_1e8 --cells 293 --steps 30 --warmup 8
