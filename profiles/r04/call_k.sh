#!/bin/bash
# round 4, GPU call K: bench.py's per-rank phase breakdown (in-process domains and two RCCL ranks on the one GPU) + DD suite after the timer changes
O=gpurun_out/r04k; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_bench.py tests/test_gpu_dd.py -x -q -m gpu --timeout 600 --durations=6 > $O/pytest.log 2>&1; echo "rc=$?"; tail -22 $O/pytest.log
timeout -k 10 300 python bench.py --domains 8 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_8dom.json 2> $O/bench_8dom.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04k/bench_8dom.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for r in d["per_rank"]["ranks"][:3]: print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
PY
