O=gpurun_out/r05w; mkdir -p $O
run() { name=$1; shift
  env "$@" EMDEE_DEBUG_PLAN=1 timeout -k 10 300 python bench.py --no-cpu-baseline --mixture --rc 3.5 --precision f32 --steps 40 --warmup 10 > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); k=d['kernels_ms']; rb=k['rebuild(bin+sort+nbr_build)']
print('f32 mixture $name', round(d['value'],1), 'steps/s, step kernel', round(d['roofline']['avg_launch_ms'],4), 'ms, rebuild', round(rb[0]/max(rb[1],1),3), 'ms x', rb[1])"
  grep "two species" $O/bench_$name.err | tail -2
}
run default A=1 && run abs EMDEE_F32_ABS=1 && run bricks7 EMDEE_TYPED_BRICKS=7 && run abs_bricks7 EMDEE_F32_ABS=1 EMDEE_TYPED_BRICKS=7 && run notyped EMDEE_NO_TYPED=1
# (second run of this script: after the fix -- rel_cell_const no longer indexes the kernel arguments with a run-time dimension)
