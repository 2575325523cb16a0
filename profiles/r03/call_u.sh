#!/bin/bash
# round 3 (second session), GPU call U: what the Langevin thermostat costs at 10^7 atoms (same box, with and without)
O=gpurun_out/r03u; mkdir -p $O
for name in nve langevin nve2 langevin2; do
  extra=""; case $name in langevin*) extra="--langevin 1.0";; esac
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 60 --warmup 15 $extra > $O/$name.json 2> $O/$name.err; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed $name"; exit $rc; fi
  python - $O/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["kernels_ms"]; rb = k["rebuild(bin+sort+nbr_build)"]
print("%-12s %7.1f steps/s  %.4f ms/step  force %.3f ms  rebuild %.3f ms x %d  %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], rb[0] / max(rb[1], 1), rb[1], d["config"]["thermostat"]))
PY
done
