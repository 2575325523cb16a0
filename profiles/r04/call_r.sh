#!/bin/bash
# round 4, GPU call R: 30,000-step soak of the 10^7-atom box on the final build; the full box on 2 and 3 RCCL ranks sharing the GPU (per-rank phases in the line)
O=gpurun_out/r04r; mkdir -p $O
timeout -k 10 500 python profiles/soak.py > $O/soak_30000_steps.txt 2>&1; echo "soak rc=$?"; tail -5 $O/soak_30000_steps.txt
: > $O/rccl_ranks_full_box.jsonl
for N in 2 3; do
  echo "# python bench.py --gpus $N --share-gpu --rccl-loopback --steps 20 --warmup 5 --target-cells 0" >> $O/rccl_ranks_full_box.jsonl
  timeout -k 10 400 python bench.py --gpus $N --share-gpu --rccl-loopback --steps 20 --warmup 5 --target-cells 0 >> $O/rccl_ranks_full_box.jsonl 2> $O/rccl_$N.err; echo "rccl $N rc=$?"
done
python - <<'PY'
import json
for l in open("gpurun_out/r04r/rccl_ranks_full_box.jsonl"):
    if l.startswith("#"): print(l.strip()); continue
    d = json.loads(l)
    print("  %.1f steps/s %.3f ms/step E/N %.14f %.14f halo %s exit_status %s" % (d["value"], d["ms_per_step"], d["energy_per_atom"]["potential"], d["energy_per_atom"]["kinetic"], d["config"]["halo_exchange"]["chosen"], d.get("exit_status")))
    for r in d["per_rank"]["ranks"]: print("    ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()})
PY
