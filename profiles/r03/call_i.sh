#!/bin/bash
# round 3, GPU call I: two-species ("typed") kernels: parity against the oracle and the general-species kernels, bench A/B
O=gpurun_out/r03i; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 600 python -m pytest tests/test_gpu_parity2.py -m gpu -q --timeout 300 -x -k "typed" > $O/pytest_typed.log 2>&1
grep -E "passed|failed|^FAILED|Error|assert" $O/pytest_typed.log | tail -12
if grep -q "failed" $O/pytest_typed.log; then tail -60 $O/pytest_typed.log; exit 0; fi
step timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED" $O/pytest.log | tail -8
B="timeout -k 10 300 python bench.py --no-cpu-baseline --mixture"
EMDEE_DEBUG_PLAN=1 step $B --rc 3.5 > $O/bench_mix35_typed.json 2> $O/bench_mix35_typed.err
grep "emdee plan" $O/bench_mix35_typed.err | sort | uniq -c | head -4
EMDEE_NO_TYPED=1 step $B --rc 3.5 > $O/bench_mix35_untyped.json 2> $O/bench_mix35_untyped.err
step $B --rc 3.5 --precision f32 > $O/bench_mix35_f32_typed.json 2> $O/bench_mix35_f32_typed.err
EMDEE_NO_TYPED=1 step $B --rc 3.5 --precision f32 > $O/bench_mix35_f32_untyped.json 2> $O/bench_mix35_f32_untyped.err
step $B --rc 2.5 > $O/bench_mix25_typed.json 2> $O/bench_mix25_typed.err
EMDEE_NO_TYPED=1 step $B --rc 2.5 > $O/bench_mix25_untyped.json 2> $O/bench_mix25_untyped.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03i/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["kernels_ms"]; rb=k["rebuild(bin+sort+nbr_build)"]
        print("%-34s %.1f steps/s  %.3f ms/step  force %.3f ms  frac %.3f  rebuild %.3f ms x %d  cap %d  E/N %.6f %.6f" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0]/max(rb[1],1), rb[1], d["neighbor_list"]["capacity"], d["energy_per_atom"]["potential"], d["energy_per_atom"]["kinetic"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
