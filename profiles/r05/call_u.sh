#!/bin/bash
# round 5, GPU call U: six separate processes, one RCCL rank each (grid 3 x 2 x 1, TCP loopback transport), step the decomposed box with
# the rebuild in the engines' own order -- against the same grid inside one process; then the full 10^7-atom box on two RCCL ranks
O=gpurun_out/r05u; mkdir -p $O
timeout -k 10 500 python profiles/rccl_ranks_one_gpu.py --world 6 --cells 30 --steps 40 > $O/rccl_six_ranks.txt 2>&1; echo "six ranks rc=$?"; tail -12 $O/rccl_six_ranks.txt
timeout -k 10 500 python bench.py --gpus 2 --share-gpu --rccl-loopback --steps 20 --warmup 5 --target-cells 0 > $O/rccl_two_ranks_full_box.json 2> $O/rccl_two_ranks_full_box.err; echo "two ranks, full box rc=$?"
python3 - $O/rccl_two_ranks_full_box.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f steps/s, %.3f ms/step, exit_status %s, degraded %s" % (d["value"], d["ms_per_step"], d.get("exit_status"), d.get("degraded")))
for r in d["per_rank"]["ranks"]:
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
PY
