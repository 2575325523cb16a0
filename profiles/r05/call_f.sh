#!/bin/bash
# round 5, GPU call F: near/far rows with the far class skipped outright while nobody has moved delta / 2:
# fused launch, rebuild, steps/s in bench.py's form and in the driver's, against plain rows on the same box
O=gpurun_out/r05f; mkdir -p $O
run() {  # name, env...
  name=$1; shift
  for form in "default:--steps 100 --warmup 20" "driver:--steps 20 --warmup 5"; do
    f=${form%%:*}; args=${form#*:}
    env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline $args > $O/${name}_$f.json 2> $O/${name}_$f.err || { echo "$name $f failed"; tail -3 $O/${name}_$f.err; return 1; }
    python - $O/${name}_$f.json "$name" $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels_ms"]; rb = k["rebuild(bin+sort+nbr_build)"]
print("%-22s %-8s %7.1f steps/s  %7.4f ms/step  fused launch %6.4f ms  frac %.3f  rebuild %6.3f ms x %2d  E/N %.10f" % (
    sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0] / max(rb[1], 1), rb[1], d["energy_per_atom"]["potential"]))
PY
  done
}
run plain A=1 || exit 1
run nearfar_0.04 EMDEE_BUILD_NEARFAR=1 EMDEE_NEAR_DELTA=0.04 || exit 1
for dl in 0.08 0.12 0.15; do
  run farskip_$dl EMDEE_BUILD_NEARFAR=1 EMDEE_NEAR_DELTA=$dl EMDEE_FAR_SKIP=1 || exit 1
done
run plain_again A=1
EMDEE_HIP_LIB=$PWD/emdee.jl_amd/variants/libemdee_hip_r04.so run r04_library A=1
