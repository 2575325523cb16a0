#!/bin/bash
# round 4, GPU call U: final check of the last build: whole -m gpu suite, smoke(), the default and the driver-form bench lines
O=gpurun_out/r04u; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver rc=$?"
python - <<'PY'
import json
for f in ("bench_default", "bench_driver"):
    d = json.loads(open("gpurun_out/r04u/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), round(d["ms_per_step"], 4), round(d["roofline"]["avg_launch_ms"], 4), round(d["roofline"]["frac"], 4), d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
