#!/bin/bash
# round 3, GPU call E: mixture rc = 3.5: deeper index prefetch with 4 lanes per atom (variant 7), round-robin build on fp32 records with 8 lanes
O=gpurun_out/r03e; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
EMDEE_BRICK_VARIANT=7 step timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "mixture or long_rows or fcc864" > $O/pytest_v7.log 2>&1
grep -E "passed|failed|^FAILED" $O/pytest_v7.log | tail -5
EMDEE_BUILD_STRIDED=1 step timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -q --timeout 600 -k "mixture or long_rows or fp32 or f32 or float32" > $O/pytest_strided.log 2>&1
grep -E "passed|failed|^FAILED" $O/pytest_strided.log | tail -5
B="timeout -k 10 200 python bench.py --no-cpu-baseline --mixture --rc 3.5"
step $B > $O/bench_mix35_v8.json 2> $O/bench_mix35_v8.err
EMDEE_BRICK_VARIANT=7 step $B > $O/bench_mix35_v7.json 2> $O/bench_mix35_v7.err
step $B --precision f32 > $O/bench_mix35_f32_v8.json 2> $O/bench_mix35_f32_v8.err
EMDEE_BUILD_STRIDED=1 step $B --precision f32 > $O/bench_mix35_f32_v8_strided.json 2> $O/bench_mix35_f32_v8_strided.err
EMDEE_BRICK_VARIANT=7 step $B --precision f32 > $O/bench_mix35_f32_v7.json 2> $O/bench_mix35_f32_v7.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03e/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["kernels_ms"]; rb=k["rebuild(bin+sort+nbr_build)"]
        print("%-36s %.1f steps/s  %.3f ms/step  force %.3f ms  frac %.3f  rebuild %.3f ms x %d  cap %d" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0]/max(rb[1],1), rb[1], d["neighbor_list"]["capacity"]))
    except Exception as e:
        print(f, "ERR", e)
PY
