#!/bin/bash
# round 3 (second session), GPU call Q: whole -m gpu suite with x sub-bins, count-free rebuilds, lock-step halves; mixture lines (typed build with opposite rows paired)
O=gpurun_out/r03q; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" $O/pytest.log | tail -12
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
step $B --mixture --rc 3.5 --steps 40 --warmup 10 > $O/bench_mix35.json 2> $O/bench_mix35.err
step $B --mixture --rc 3.5 --precision f32 --steps 40 --warmup 10 > $O/bench_mix35_f32.json 2> $O/bench_mix35_f32.err
step $B --mixture --steps 40 --warmup 10 > $O/bench_mix25.json 2> $O/bench_mix25.err
EMDEE_NO_SUBBINS=1 step $B --mixture --steps 40 --warmup 10 > $O/bench_mix25_nosub.json 2> $O/bench_mix25_nosub.err
step $B --cells 63 > $O/bench_1m.json 2> $O/bench_1m.err
step $B --cells 293 --steps 30 --warmup 8 > $O/bench_100m.json 2> $O/bench_100m.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03q/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["kernels_ms"]; rb=k["rebuild(bin+sort+nbr_build)"]
        print("%-28s %.1f steps/s  %.4f ms/step  force %.3f ms  frac %.3f  rebuild %.3f ms x %d" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0]/max(rb[1],1), rb[1]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
