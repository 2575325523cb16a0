#!/bin/bash
# round 3 (second session), GPU call L: count-free rebuild messages: DD tests, overhead probes, rebuild timeline
O=gpurun_out/r03l; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_bench.py tests/test_gpu_domain.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" $O/pytest.log | tail -8
EMDEE_DD_NO_SHORTCUT=1 EMDEE_DD_OVERLAP=0 step timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/dd_one_domain_inorder_full.txt 2>&1
EMDEE_DD_COUNT_FREE=0 EMDEE_DD_NO_SHORTCUT=1 EMDEE_DD_OVERLAP=0 step timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/dd_one_domain_inorder_full_counts.txt 2>&1
grep -H atoms $O/dd_one_domain*.txt
step timeout -k 10 300 bash profiles/dd_rebuild_timeline.sh 68 $O/tl > $O/timeline_stdout.txt 2>&1; tail -60 $O/timeline_stdout.txt
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
step $B --domains 8 --steps 40 --warmup 10 > $O/bench_dd8.json 2> $O/bench_dd8.err
step $B --domains 2 --steps 40 --warmup 10 > $O/bench_dd2.json 2> $O/bench_dd2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03l/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %.1f steps/s  %.4f ms/step" % (f.split("/")[-1], d["value"], d["ms_per_step"]), d["config"].get("halo_exchange"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
