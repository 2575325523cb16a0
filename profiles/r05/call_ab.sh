#!/bin/bash
# round 5, GPU call AB: round 4's tree against this round's on one box, the paths call Y did not cover: the operator driven from the host
# side, the thermostat, the two-species box at rc = 2.5 (general-species kernels), the fp32 mixture, in-process domains
O=gpurun_out/r05ab; mkdir -p $O
run() { tree=$1; name=$2; shift 2
  ( cd $tree && timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $OLDPWD/$O/bench_${name}.json 2> $OLDPWD/$O/bench_${name}.err )
  python -c "
import json; d=json.loads(open('$O/bench_${name}.json').read().strip().splitlines()[-1]); k=d['kernels_ms']; rb=k['rebuild(bin+sort+nbr_build)']
print('%-24s' % '$name', round(d['value'],1), 'steps/s,', round(d['ms_per_step'],4), 'ms/step, step kernel', round(d['roofline']['avg_launch_ms'],4), 'ms, rebuild', round(rb[0]/max(rb[1],1),3), 'ms x', rb[1])"
}
for t in _r04 .; do n=$( [ $t = . ] && echo r05 || echo r04 )
  ( cd $t && timeout -k 10 300 python profiles/operator_path.py 136 2>/dev/null | grep -v amdgpu | sed "s/^/$n operator: /" )
  run $t ${n}_langevin --langevin 1 --steps 60 --warmup 15
  run $t ${n}_mix_rc25 --mixture --steps 60 --warmup 10
  run $t ${n}_mix_rc35_f32 --mixture --rc 3.5 --precision f32 --steps 40 --warmup 10
  run $t ${n}_mix_rc35 --mixture --rc 3.5 --steps 40 --warmup 10
  run $t ${n}_domains2 --domains 2 --steps 40 --warmup 10
  run $t ${n}_domains8 --domains 8 --steps 40 --warmup 10
done
