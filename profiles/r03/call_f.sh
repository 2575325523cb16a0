#!/bin/bash
# round 3, GPU call F: maxima from k_brick_tables under a kept plan; DD suite; rank-size probes
O=gpurun_out/r03f; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 1000 python -m pytest tests/test_gpu_bench.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED" $O/pytest.log | tail -8
B="timeout -k 10 200 python bench.py --no-cpu-baseline"
EMDEE_DEBUG_PLAN=1 step $B > $O/bench_default.json 2> $O/bench_default.err
EMDEE_PLAN_MAXIMA=tables step $B > $O/bench_maxima_tables.json 2> $O/bench_maxima_tables.err
grep -c "emdee plan" $O/bench_default.err
step $B --cells 63 > $O/bench_1m.json 2> $O/bench_1m.err
step timeout -k 10 200 python profiles/dd_one_domain_overhead.py > $O/dd_one_domain.txt 2>&1
EMDEE_DD_OVERLAP=0 step timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/dd_one_domain_inorder.txt 2>&1
cat $O/dd_one_domain.txt $O/dd_one_domain_inorder.txt | grep -v amdgpu
step $B --domains 8 --target-cells 0 > $O/bench_dd8.json 2> $O/bench_dd8.err
EMDEE_DD_OVERLAP=0 step $B --domains 8 --target-cells 0 > $O/bench_dd8_inorder.json 2> $O/bench_dd8_inorder.err
step $B --domains 2 --target-cells 0 > $O/bench_dd2.json 2> $O/bench_dd2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03f/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["kernels_ms"]; rb=k["rebuild(bin+sort+nbr_build)"]
        print("%-28s %.1f steps/s  %.3f ms/step  force %.3f ms  frac %.3f  rebuild %.3f ms x %d" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0]/max(rb[1],1), rb[1]))
    except Exception as e:
        print(f, "ERR", e)
PY
