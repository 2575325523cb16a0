#!/bin/bash
# round 3, GPU call D: one-instruction hit bit in the build (parity + timing), mixture: 1024 threads with 4 lanes per atom
O=gpurun_out/r03d; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_parity.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED" $O/pytest.log | tail -15
B="timeout -k 10 200 python bench.py --no-cpu-baseline"
step $B > $O/bench_default.json 2> $O/bench_default.err
step $B --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
step $B --precision f32 > $O/bench_f32.json 2> $O/bench_f32.err
step $B --mixture --rc 3.5 > $O/bench_mix35_v8.json 2> $O/bench_mix35_v8.err
EMDEE_BRICK_VARIANT=7 step $B --mixture --rc 3.5 > $O/bench_mix35_v7.json 2> $O/bench_mix35_v7.err
EMDEE_BRICK_VARIANT=7 step $B --mixture --rc 3.5 --precision f32 > $O/bench_mix35_f32_v7.json 2> $O/bench_mix35_f32_v7.err
step $B --mixture --rc 3.5 --precision f32 > $O/bench_mix35_f32_v8.json 2> $O/bench_mix35_f32_v8.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03d/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["kernels_ms"]; rb=k["rebuild(bin+sort+nbr_build)"]
        print("%-28s %.1f steps/s  %.3f ms/step  force %.3f ms  frac %.3f  rebuild %.3f ms x %d  cap %d" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0]/max(rb[1],1), rb[1], d["neighbor_list"]["capacity"]))
    except Exception as e:
        print(f, "ERR", e)
PY
