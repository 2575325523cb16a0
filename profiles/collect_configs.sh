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
# the driver's form of the headline run (20 timed steps after 5: the melting lattice rebuilds more often early on)
run --steps 20 --warmup 5
# the native decomposition with every domain on this one GPU (device copies instead of RCCL): total work of all ranks, serialised
run --domains 2 --steps 40 --warmup 10
run --domains 8 --steps 40 --warmup 10
mkdir -p profiles/$TAG && cp $OUT/configs.jsonl profiles/$TAG/final_bench_configs.jsonl
# 1-GPU reference of the north-star target box, quoted by bench.py's `target_box` on N > 1 runs
python3 - $OUT/configs.jsonl $TAG <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("#"):
        continue
    d = json.loads(line)
    if d["config"]["atoms"] == 100615028 and d["n_gpus"] == 1 and d["config"].get("decomposition") is None:
        json.dump({"atoms": 100615028, "value": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
                   "source": "profiles/%s/final_bench_configs.jsonl (bench.py --cells 293 --steps 30 --warmup 8, one MI355X)" % sys.argv[2]},
                  open("profiles/target_box_1gpu.json", "w"), indent=1)
PY
python3 - $OUT/configs.jsonl <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("#"):
        print(line.strip()); continue
    d = json.loads(line)
    print("   %.1f steps/s  %.3f ms/step  %.3g pair-interactions/s  frac %.3f" % (d.get("box_steps_per_sec", d["value"]), d["ms_per_step"], d["pair_interactions_per_sec"], d["roofline"]["frac"]))
PY
