#!/bin/bash
# round 3, GPU call J: whole -m gpu suite with the typed kernels in (single process and decomposed), headline regression check, mixture lines
O=gpurun_out/r03j; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 800 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED" $O/pytest.log | tail -8
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
step $B > $O/bench_default.json 2> $O/bench_default.err
step $B --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
step $B --mixture --rc 3.5 > $O/bench_mix35.json 2> $O/bench_mix35.err
step $B --mixture --rc 3.5 --precision f32 > $O/bench_mix35_f32.json 2> $O/bench_mix35_f32.err
step $B --mixture --rc 3.5 --domains 2 --target-cells 0 > $O/bench_mix35_dd2.json 2> $O/bench_mix35_dd2.err
EMDEE_NO_TYPED=1 step $B --mixture --rc 3.5 --domains 2 --target-cells 0 > $O/bench_mix35_dd2_untyped.json 2> $O/bench_mix35_dd2_untyped.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03j/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["kernels_ms"]; rb=k["rebuild(bin+sort+nbr_build)"]
        print("%-34s %.1f steps/s  %.3f ms/step  force %.3f ms  frac %.3f  rebuild %.3f ms x %d  cap %d  E/N %.6f %.6f" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0]/max(rb[1],1), rb[1], d["neighbor_list"]["capacity"], d["energy_per_atom"]["potential"], d["energy_per_atom"]["kinetic"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
