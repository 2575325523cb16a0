#!/bin/bash
# round 4, GPU call A: baseline of the round-3 build on today's box (default bench, driver form)
O=gpurun_out/r04a; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err; echo rc=$?
python - <<'PY'
import json
for f in ("bench_default", "bench_driver"):
    d = json.loads(open("gpurun_out/r04a/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["kernels_ms"])
PY
