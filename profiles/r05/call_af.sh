O=gpurun_out/r05af; mkdir -p $O
run() { name=$1; shift
  env "$@" EMDEE_DEBUG_PLAN=1 timeout -k 10 300 python bench.py --no-cpu-baseline --mixture --steps 60 --warmup 10 > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); k=d['kernels_ms']; rb=k['rebuild(bin+sort+nbr_build)']
print('mixture rc=2.5 $name', round(d['value'],1), 'steps/s, step kernel', round(d['roofline']['avg_launch_ms'],4), 'ms, rebuild', round(rb[0]/max(rb[1],1),3), 'ms x', rb[1])"
  grep "two species" $O/bench_$name.err | tail -1
}
run general A=1 && run typed_all EMDEE_TYPED_ALL=1 && run general_2 A=1 && run typed_all_2 EMDEE_TYPED_ALL=1
