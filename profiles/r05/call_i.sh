#!/bin/bash
# round 5, GPU call I: next neighbour's coordinates requested before the arithmetic of the current one: the mixture's typed kernel
# (product build against the round-4 library) and, as a build variant, the single-species fp64 kernel of the headline
O=gpurun_out/r05i; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels_ms"]; rb = k["rebuild(bin+sort+nbr_build)"]
print("%-28s %7.1f steps/s  %7.4f ms/step  fused launch %6.4f ms  frac %.3f  rebuild %6.3f ms x %2d  E/N %.10f" % (
    sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0] / max(rb[1], 1), rb[1], d["energy_per_atom"]["potential"]))
PY
}
for rep in 1 2; do
for lib in new r04; do
  if [ $lib = r04 ]; then export EMDEE_HIP_LIB=$PWD/emdee.jl_amd/variants/libemdee_hip_r04.so; else unset EMDEE_HIP_LIB; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --mixture --rc 3.5 --steps 40 --warmup 10 > $O/mix_${lib}_$rep.json 2> $O/mix_${lib}_$rep.err || exit 1
  line $O/mix_${lib}_$rep.json "mixture rc3.5 $lib #$rep"
done
done
for rep in 1 2; do
for lib in new ahead; do
  if [ $lib = ahead ]; then export EMDEE_HIP_LIB=$PWD/emdee.jl_amd/variants/libemdee_hip_ahead.so; else unset EMDEE_HIP_LIB; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100 --warmup 20 > $O/head_${lib}_$rep.json 2> $O/head_${lib}_$rep.err || exit 1
  line $O/head_${lib}_$rep.json "headline fp64 $lib #$rep"
done
done
unset EMDEE_HIP_LIB
