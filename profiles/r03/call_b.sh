#!/bin/bash
# round 3, GPU call B: full -m gpu suite (no -x), fp32 probe, plan debug, dd overhead probes, 8-domain rehearsal
O=gpurun_out/r03b; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 600 -s > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED|fp32 10" $O/pytest.log | tail -15
step timeout -k 10 120 python tests/probe_fp32_errors.py > $O/fp32_probe.txt 2>&1
cat $O/fp32_probe.txt
EMDEE_DEBUG_PLAN=1 step timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
grep "emdee plan" $O/bench_default.err | sort | uniq -c | head -5
EMDEE_PLAN_SYNC=1 step timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_plansync.json 2> $O/bench_plansync.err
step timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
step timeout -k 10 200 python bench.py --cells 63 --no-cpu-baseline > $O/bench_1m.json 2> $O/bench_1m.err
EMDEE_PLAN_SYNC=1 step timeout -k 10 200 python bench.py --cells 63 --no-cpu-baseline > $O/bench_1m_plansync.json 2> $O/bench_1m_plansync.err
step timeout -k 10 200 python profiles/dd_one_domain_overhead.py > $O/dd_one_domain.txt 2>&1
EMDEE_DD_OVERLAP=0 step timeout -k 10 200 python profiles/dd_one_domain_overhead.py 86 dd > $O/dd_one_domain_inorder.txt 2>&1
cat $O/dd_one_domain.txt $O/dd_one_domain_inorder.txt
step timeout -k 10 200 python profiles/dd_rank_proxy.py > $O/dd_rank_proxy.txt 2>&1
cat $O/dd_rank_proxy.txt
step timeout -k 10 300 python bench.py --domains 8 --no-cpu-baseline --target-cells 0 > $O/bench_dd8.json 2> $O/bench_dd8.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03b/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.1f steps/s" % d["value"], "%.3f ms/step" % d["ms_per_step"], "force %.3f ms" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], d["kernels_ms"], d["neighbor_list"])
    except Exception as e:
        print(f, "ERR", e)
PY
