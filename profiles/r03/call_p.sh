#!/bin/bash
# round 3 (second session), GPU call P: x sub-bins in the sort and the build: parity suite, A/B against EMDEE_NO_SUBBINS=1
O=gpurun_out/r03p; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" $O/pytest.log | tail -8
run() { name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline $ARGS > $O/$name.json 2> $O/$name.err; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed $name"; exit $rc; fi
  python - $O/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["kernels_ms"]; rb = k["rebuild(bin+sort+nbr_build)"]
    print("%-22s %7.1f steps/s  %.4f ms/step  force %.3f ms  frac %.3f  rebuild %.3f ms x %d  E/N %.6f  listed %d" % (sys.argv[2], d["value"], d["ms_per_step"],
          d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0] / max(rb[1], 1), rb[1], d["energy_per_atom"]["potential"], d["neighbor_list"]["listed"]))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace(".json", ".err")).read()[-300:])
PY
}
ARGS=""
run sub EMDEE_DEBUG_PLAN=1
run nosub EMDEE_NO_SUBBINS=1
run sub2 A=1
run nosub2 EMDEE_NO_SUBBINS=1
ARGS="--steps 20 --warmup 5"
run drv_sub A=1
run drv_nosub EMDEE_NO_SUBBINS=1
ARGS="--precision f32 --steps 60 --warmup 10"
run f32_sub A=1
run f32_nosub EMDEE_NO_SUBBINS=1
grep "emdee plan" $O/sub.err | tail -2
