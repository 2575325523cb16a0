#!/bin/bash
# round 5, GPU call X: where the fp32 configuration (configs[3]) stands against round 4 (806-815 steps/s there, 775-783 here):
# with and without the coordinate prefetch (-DEMDEE_PREFETCH_XJ=0, fp32 translation unit only), with absolute records, alternating on one box
O=gpurun_out/r05x; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --precision f32 --steps 60 --warmup 10 > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); k=d['kernels_ms']; rb=k['rebuild(bin+sort+nbr_build)']
print('fp32 $name', round(d['value'],1), 'steps/s, fused launch', round(d['roofline']['avg_launch_ms'],4), 'ms, rebuild', round(rb[0]/max(rb[1],1),3), 'ms x', rb[1])"
}
NA=EMDEE_HIP_LIB=$PWD/emdee.jl_amd/variants/libemdee_hip_noahead32.so
run default_1 A=1 && run noahead_1 $NA && run abs_1 EMDEE_F32_ABS=1 && run noahead_abs_1 $NA EMDEE_F32_ABS=1 && run default_2 A=1 && run noahead_2 $NA && run abs_2 EMDEE_F32_ABS=1 && run noahead_abs_2 $NA EMDEE_F32_ABS=1
