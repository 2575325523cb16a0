#!/bin/bash
# round 3, GPU call A: full -m gpu suite, fp32 error probe, bench lines (default, driver form, stride 96, fp32, mixture)
O=gpurun_out/r3a; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1
tail -5 $O/pytest.log
step timeout -k 10 120 python tests/probe_fp32_errors.py > $O/fp32_probe.txt 2>&1
cat $O/fp32_probe.txt
step timeout -k 10 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
step timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
EMDEE_STRIDE=96 step timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_stride96.json 2> $O/bench_stride96.err
step timeout -k 10 200 python bench.py --precision f32 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err
step timeout -k 10 300 python bench.py --mixture --rc 3.5 --no-cpu-baseline > $O/bench_mix35.json 2> $O/bench_mix35.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3a/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.1f steps/s" % d["value"], "force %.3f ms" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], d["kernels_ms"], d["neighbor_list"])
    except Exception as e:
        print(f, "ERR", e)
PY
