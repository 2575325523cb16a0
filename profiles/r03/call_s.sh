#!/bin/bash
# round 3 (second session), GPU call S: one bench line per BASELINE configuration on the final build (profiles/collect_configs.sh r03),
# mixture lines, and separate-process RCCL ranks on the one GPU (grids 2x1x1, 2x2x1, 3x2x1; fp32 mixture rc = 3.5 on three)
O=gpurun_out/r03s; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 1000 bash profiles/collect_configs.sh r03 > $O/collect_configs.log 2>&1
tail -24 $O/collect_configs.log
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
step $B --mixture --steps 40 --warmup 10 > $O/bench_mix25.json 2> $O/bench_mix25.err
step $B --mixture --rc 3.5 --precision f32 --steps 40 --warmup 10 > $O/bench_mix35_f32.json 2> $O/bench_mix35_f32.err
for w in 2 4 6; do
  step timeout -k 10 400 python profiles/rccl_ranks_one_gpu.py --world $w --cells 24 --steps 24 > $O/rccl_w$w.txt 2>&1
  tail -4 $O/rccl_w$w.txt
done
step timeout -k 10 400 python profiles/rccl_ranks_one_gpu.py --world 3 --cells 24 --steps 24 --precision f32 --mixture --rc 3.5 --switch-overlap > $O/rccl_w3_mix.txt 2>&1
tail -3 $O/rccl_w3_mix.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03s/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d["kernels_ms"]; rb=k["rebuild(bin+sort+nbr_build)"]
        print("%-28s %.1f steps/s  %.4f ms/step  force %.3f ms  frac %.3f  rebuild %.3f ms x %d" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0]/max(rb[1],1), rb[1]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
cp $PWD/gpurun_out/collect_r03/configs.jsonl $O/configs.jsonl 2>/dev/null; cp profiles/target_box_1gpu.json $O/ 2>/dev/null
