#!/bin/bash
# round 3 (second session), GPU call O: near/far rows with ONE emission loop per tile row (EMDEE_BUILD_NEARFAR=1) against two loops and against plain rows
O=gpurun_out/r03o; mkdir -p $O
run() { # name env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline $ARGS > $O/$name.json 2> $O/$name.err; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed $name"; exit $rc; fi
  python - $O/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["kernels_ms"]; rb = k["rebuild(bin+sort+nbr_build)"]
    print("%-22s %7.1f steps/s  %.4f ms/step  force %.3f ms  frac %.3f  rebuild %.3f ms x %d  E/N %.6f" % (sys.argv[2], d["value"], d["ms_per_step"],
          d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0] / max(rb[1], 1), rb[1], d["energy_per_atom"]["potential"]))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
ARGS=""
run plain A=1
run nearfar EMDEE_BUILD_NEARFAR=1
run nearfar_two_loops EMDEE_BUILD_NEARFAR=1 EMDEE_HIP_LIB=$PWD/emdee.jl_amd/variants/libemdee_hip_nf2loops.so
run nearfar_d02 EMDEE_BUILD_NEARFAR=1 EMDEE_NEAR_DELTA=0.02
run nearfar_d06 EMDEE_BUILD_NEARFAR=1 EMDEE_NEAR_DELTA=0.06
run nearfar_d08 EMDEE_BUILD_NEARFAR=1 EMDEE_NEAR_DELTA=0.08
ARGS="--steps 20 --warmup 5"
run drv_plain A=1
run drv_nearfar EMDEE_BUILD_NEARFAR=1
run drv_plain2 A=1
run drv_nearfar2 EMDEE_BUILD_NEARFAR=1
