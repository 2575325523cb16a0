#!/bin/bash
# round 3 (second session), GPU call V: own cells that hold only ghosts leave the own-atom loops (decomposed runs): DD tests, rank proxy, rehearsals
O=gpurun_out/r03v; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_bench.py tests/test_gpu_domain.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" $O/pytest.log | tail -8
step timeout -k 10 200 python profiles/dd_rank_proxy.py > $O/dd_rank_proxy.txt 2>&1
grep -v amdgpu $O/dd_rank_proxy.txt
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
step $B --domains 8 --steps 40 --warmup 10 > $O/bench_dd8.json 2> $O/bench_dd8.err
step $B --domains 2 --steps 40 --warmup 10 > $O/bench_dd2.json 2> $O/bench_dd2.err
step $B --steps 40 --warmup 10 > $O/bench_1.json 2> $O/bench_1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03v/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %.1f steps/s  %.4f ms/step  E/N %.9f %.9f" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["energy_per_atom"]["potential"], d["energy_per_atom"]["kinetic"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
