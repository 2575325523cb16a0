#!/bin/bash
# round 3 (second session), GPU call AF: in-process transport as one pull kernel per receiving domain on one communication stream: DD suite, rehearsals
O=gpurun_out/r03af; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_bench.py tests/test_gpu_domain.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
$B --domains 8 --steps 40 --warmup 10 > $O/bench_dd8.json 2> $O/bench_dd8.err
$B --domains 2 --steps 40 --warmup 10 > $O/bench_dd2.json 2> $O/bench_dd2.err
$B --domains 8 --steps 40 --warmup 10 > $O/bench_dd8b.json 2> $O/bench_dd8b.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03af/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %.1f steps/s  %.4f ms/step  E/N %.9f %.9f" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["energy_per_atom"]["potential"], d["energy_per_atom"]["kinetic"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
