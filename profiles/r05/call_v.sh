#!/bin/bash
# round 5, GPU call V: 4 eps_ij folded into the switch constants of the typed force-only launches (lj_pair.hpp LJSeg): typed tests, then
# configs[4] fp64 and fp32 with the fold and without it (profiles/build_variant.sh nofold -DEMDEE_TYPED_NO_FOLD=1 both), alternating
O=gpurun_out/r05v; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_gpu_dd.py -x -q --timeout 600 -k "typed or species or mixture or long_rows" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
run() {  # name, precision, env...
  name=$1; prec=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --mixture --rc 3.5 --precision $prec --steps 40 --warmup 10 > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); k=d['kernels_ms']; rb=k['rebuild(bin+sort+nbr_build)']
print('mixture $name', round(d['value'],1), 'steps/s, step kernel', round(d['roofline']['avg_launch_ms'],4), 'ms, rebuild', round(rb[0]/max(rb[1],1),3), 'ms x', rb[1], ' E/N', d['energy_per_atom']['potential'])"
}
NF=EMDEE_HIP_LIB=$PWD/emdee.jl_amd/variants/libemdee_hip_nofold.so
run fold_1 f64 A=1 && run nofold_1 f64 $NF && run fold_2 f64 A=1 && run nofold_2 f64 $NF && run f32_fold f32 A=1 && run f32_nofold f32 $NF
