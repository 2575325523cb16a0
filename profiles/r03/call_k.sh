#!/bin/bash
# round 3 (second session), GPU call K: posted read-backs + opposite row pairing: GPU tests, A/B of the read-back form, DD rebuild timeline
O=gpurun_out/r03k; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" $O/pytest.log | tail -8
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
step $B > $O/bench_default.json 2> $O/bench_default.err
step $B --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
step $B --cells 63 > $O/bench_1m.json 2> $O/bench_1m.err
EMDEE_READBACK=copy step $B --cells 63 > $O/bench_1m_copy.json 2> $O/bench_1m_copy.err
step $B --cells 6 --steps 200 > $O/bench_864.json 2> $O/bench_864.err
EMDEE_READBACK=copy step $B --cells 6 --steps 200 > $O/bench_864_copy.json 2> $O/bench_864_copy.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03k/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["kernels_ms"]; rb=k["rebuild(bin+sort+nbr_build)"]
        print("%-28s %.1f steps/s  %.4f ms/step  force %.3f ms  frac %.3f  rebuild %.3f ms x %d" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0]/max(rb[1],1), rb[1]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
step timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 both > $O/dd_one_domain.txt 2>&1
EMDEE_DD_OVERLAP=0 step timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/dd_one_domain_inorder.txt 2>&1
EMDEE_DD_NO_SHORTCUT=1 EMDEE_DD_OVERLAP=0 step timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/dd_one_domain_inorder_full.txt 2>&1
EMDEE_READBACK=copy EMDEE_DD_NO_SHORTCUT=1 EMDEE_DD_OVERLAP=0 step timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/dd_one_domain_inorder_full_copy.txt 2>&1
grep -H atoms $O/dd_one_domain*.txt
step timeout -k 10 300 bash profiles/dd_rebuild_timeline.sh 68 $O/tl > $O/timeline_stdout.txt 2>&1; tail -3 $O/timeline_stdout.txt
