#!/bin/bash
# The other BASELINE.json configurations on one GPU, one bench line each -> profiles/<round>/final_bench_configs.jsonl
# Usage (on the GPU box, from the repository root):  bash profiles/collect_configs.sh r01
TAG=${1:-r01}; R=$PWD; OUT=$R/gpurun_out/collect_$TAG; mkdir -p $OUT; : > $OUT/configs.jsonl
run() { echo "# bench.py $*" >> $OUT/configs.jsonl; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline >> $OUT/configs.jsonl 2>> $OUT/configs.err || echo "failed: $*"; }
# configs[0] with its own CPU baseline (single-threaded all-pairs, as the reference's CPU path)
echo "# bench.py --cells 6 --steps 200 --warmup 20 (with cpu_baseline)" >> $OUT/configs.jsonl
timeout -k 10 400 python bench.py --cells 6 --steps 200 --warmup 20 >> $OUT/configs.jsonl 2>> $OUT/configs.err || echo "failed: cells 6"
run --cells 63 --steps 100 --warmup 20
run --precision f32 --steps 60 --warmup 10
run --mixture --rc 3.5 --steps 40 --warmup 10
run --cells 293 --steps 30 --warmup 8
mkdir -p profiles/$TAG && cp $OUT/configs.jsonl profiles/$TAG/final_bench_configs.jsonl
python3 - $OUT/configs.jsonl <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("#"):
        print(line.strip()); continue
    d = json.loads(line)
    print("   %.1f steps/s  %.3f ms/step  %.3g pair-interactions/s  frac %.3f" % (d["box_steps_per_sec"], d["ms_per_step"], d["pair_interactions_per_sec"], d["roofline"]["frac"]))
PY
