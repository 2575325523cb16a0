O=gpurun_out/r05ae; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_gpu_dd.py -x -q --timeout 600 -k "f32 or fp32 or float32 or mixed or precision or typed or species" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
run() { tree=$1; name=$2; shift 2
  ( cd $tree && timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $OLDPWD/$O/bench_${name}.json 2> $OLDPWD/$O/bench_${name}.err )
  python -c "
import json; d=json.loads(open('$O/bench_${name}.json').read().strip().splitlines()[-1]); k=d['kernels_ms']; rb=k['rebuild(bin+sort+nbr_build)']
print('%-22s' % '$name', round(d['value'],1), 'steps/s, step kernel', round(d['roofline']['avg_launch_ms'],4), 'ms, rebuild', round(rb[0]/max(rb[1],1),3), 'ms x', rb[1])"
}
for rep in 1 2 3; do
  run _r04 r04_f32_$rep --precision f32 --steps 60 --warmup 10
  run .    r05_f32_$rep --precision f32 --steps 60 --warmup 10
done
run . r05_mix_f32 --mixture --rc 3.5 --precision f32 --steps 40 --warmup 10
