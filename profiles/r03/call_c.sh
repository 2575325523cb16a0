#!/bin/bash
# round 3, GPU call C: near/far build (ALG 23): list parity tests, A/B against ALG 13, delta sweep
O=gpurun_out/r03c; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_parity.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED" $O/pytest.log | tail -15
B="timeout -k 10 200 python bench.py --no-cpu-baseline"
EMDEE_BUILD_NEARFAR=0 step $B > $O/bench_nf0.json 2> $O/bench_nf0.err
for d in 0.04 0.08 0.12 0.16; do EMDEE_NEAR_DELTA=$d step $B > $O/bench_nf_d$d.json 2> $O/bench_nf_d$d.err; done
EMDEE_BUILD_NEARFAR=0 step $B --steps 20 --warmup 5 > $O/bench_driver_nf0.json 2> $O/bench_driver_nf0.err
step $B --steps 20 --warmup 5 > $O/bench_driver_nf1.json 2> $O/bench_driver_nf1.err
EMDEE_BUILD_NEARFAR=0 step $B --precision f32 > $O/bench_f32_nf0.json 2> $O/bench_f32_nf0.err
step $B --precision f32 > $O/bench_f32_nf1.json 2> $O/bench_f32_nf1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03c/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["kernels_ms"]; rb=k["rebuild(bin+sort+nbr_build)"]
        print("%-28s %.1f steps/s  %.3f ms/step  force %.3f ms  frac %.3f  rebuild %.3f ms x %d" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0]/max(rb[1],1), rb[1]))
    except Exception as e:
        print(f, "ERR", e)
PY
